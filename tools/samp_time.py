import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import os
os.environ["VV_NO_TORCH_PRELOAD"] = "1"
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos
ds = SyntheticVideos(seed=1701, n_videos=2048)
s = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=1024, context_size=5, num_negative_samples=50, max_buffer_size=5000, negative_swap_percentage=50)
for _ in range(5): s.next()
best = 1e9
for rep in range(12):
    t0 = time.perf_counter()
    for _ in range(40): s.next()
    best = min(best, (time.perf_counter() - t0) / 40)
print("%.3f ms per 1024-item batch (best of 12 x 40)" % (best * 1e3))
