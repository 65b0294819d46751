"""Host sampler rate with spread: ms per batch of the prefetch pipeline (2 / 3 stage threads) and of the calling-thread
path at the per-GPU batch (1024) and the 8-GPU global batch (8192); `runs` independent samplers each, min / median / max.
Usage: python tools/samp_rates.py [runs] [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ds = SyntheticVideos(seed=1701, n_videos=2048)
res = {"host_cpus": os.cpu_count(), "runs": runs, "cases": []}
for B in (1024, 8192):
    for threads in (0, 2, 3, 4):
        ms = []
        for run in range(runs):
            s = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=5, num_negative_samples=50,
                           max_buffer_size=5000, negative_swap_percentage=50)
            if threads:
                s.prefetch_start(depth=8, threads=threads)
            n = max(8, 200 * 1024 // B)
            for _ in range(max(4, n // 8)):
                s.next()
            t0 = time.perf_counter()
            for _ in range(n):
                s.next()
            ms.append((time.perf_counter() - t0) / n * 1e3)
            s.close()
        a = np.sort(ms)
        res["cases"].append({"batch": B, "threads": threads, "ms_per_batch": {"min": round(float(a[0]), 4), "median": round(float(np.median(a)), 4), "max": round(float(a[-1]), 4)}})
        print("batch %5d  threads %d: min %.3f  median %.3f  max %.3f ms per batch (%d runs of %d batches)" % (B, threads, a[0], np.median(a), a[-1], runs, n), flush=True)
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
