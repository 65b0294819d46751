"""Lab: block size of the generated stream and software prefetch distance against the 4-thread pipeline's rate (8192-item batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos
ds = SyntheticVideos(seed=1701, n_videos=2048)
def run(B, env):
    for k in ("VV_SAMPLER_BLOCK", "VV_SAMPLER_PREFETCH", "VV_SAMPLER_RING_ITEMS"): os.environ.pop(k, None)
    os.environ.update(env)
    ms = []
    for r in range(5):
        s = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=5, num_negative_samples=50, max_buffer_size=5000, negative_swap_percentage=50)
        s.prefetch_start(depth=8, threads=4)
        n = max(8, 400 * 1024 // B)
        for _ in range(4): s.next()
        t0 = time.perf_counter()
        for _ in range(n): s.next()
        ms.append((time.perf_counter() - t0) / n * 1e3)
        s.close()
    a = np.sort(ms)
    print("B %5d %-60s min %.3f median %.3f max %.3f" % (B, env, a[0], np.median(a), a[-1]), flush=True)
for B in (8192, 1024):
    run(B, {})
    for ri in (256, 512, 1024, 2048, 4096, 16384):
        run(B, {"VV_SAMPLER_RING_ITEMS": str(ri)})
    run(B, {})
