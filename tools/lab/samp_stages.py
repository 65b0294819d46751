"""Where the sampler's prefetch pipeline spends its time: per stage the share of the wall clock it waited (vv_sampler_stat
3..6), with a consumer that pops as fast as it can (producer-bound: what matters when the GPUs outrun the sampler)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos
ds = SyntheticVideos(seed=1701, n_videos=2048)
for B in (1024, 8192):
    for threads in (2, 3, 4):
        for rep in range(3):
            s = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=5, num_negative_samples=50,
                           max_buffer_size=5000, negative_swap_percentage=50)
            s.prefetch_start(depth=8, threads=threads)
            n = max(8, 400 * 1024 // B)
            for _ in range(4): s.next()
            t0 = time.perf_counter(); c0 = [s.stat(i) for i in (3, 4, 5, 6, 8)]
            for _ in range(n): s.next()
            el = time.perf_counter() - t0; c1 = [s.stat(i) for i in (3, 4, 5, 6, 8)]
            d = [b - a for a, b in zip(c0, c1)]
            print("B %5d threads %d: %.3f ms/batch; waiting: walk %4.1f%% (+ %4.1f%% for the stream thread)  negs %4.1f%%  frames %4.1f%%" % (
                B, threads, el / n * 1e3, 100 * d[1] / d[0], 100 * d[4] / d[0], 100 * d[2] / d[0], 100 * d[3] / d[0]), flush=True)
            s.close()
