"""Lab: tools/samp_rates.py's 8192-item case with the CONSUMER (this thread) pinned to the CPU it runs on / left to the scheduler, 20 samplers each:
are the one-in-ten slow runs the consumer landing on a stage thread's core?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos
import ctypes
_libc = ctypes.CDLL(None)
def _getcpu(): return _libc.sched_getcpu()
ds = SyntheticVideos(seed=1701, n_videos=2048)
full = os.sched_getaffinity(0)
for mode in ("free", "pinned", "free", "pinned"):
    ms = []; cpus = []
    for run in range(20):
        s = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=8192, context_size=5, num_negative_samples=50, max_buffer_size=5000, negative_swap_percentage=50)
        s.prefetch_start(depth=8, threads=4)
        if mode == "pinned":                       # (after the stage threads exist: they inherit the creator's mask)
            os.sched_setaffinity(0, {_getcpu()})
        for _ in range(4): s.next()
        t0 = time.perf_counter()
        for _ in range(25): s.next()
        ms.append((time.perf_counter() - t0) / 25 * 1e3)
        s.close()
        os.sched_setaffinity(0, full)
    a = np.sort(ms)
    print("consumer %-6s: min %.3f  median %.3f  p90 %.3f  max %.3f ms per 8192 items; all: %s" % (mode, a[0], np.median(a), a[17], a[-1], " ".join("%.2f" % x for x in ms)), flush=True)
