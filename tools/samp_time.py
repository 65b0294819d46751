"""Host sampler rate: ms per batch for the calling-thread path (threads 0) and the prefetch pipeline (1-3 threads),
at the per-GPU batch (1024) and the 8-GPU global batch (8192).  Usage: python tools/samp_time.py [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VV_NO_TORCH_PRELOAD"] = "1"
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos

ds = SyntheticVideos(seed=1701, n_videos=2048)
res = {"host_cpus": os.cpu_count(), "cases": []}
for B in (1024, 8192):
    for threads in (0, 1, 2, 3):
        s = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=5, num_negative_samples=50,
                       max_buffer_size=5000, negative_swap_percentage=50)
        if threads:
            s.prefetch_start(depth=4, threads=threads)
        n = max(4, 40 * 1024 // B)
        for _ in range(5):
            s.next()
        best = 1e9
        for rep in range(8):
            t0 = time.perf_counter()
            for _ in range(n):
                s.next()
            best = min(best, (time.perf_counter() - t0) / n)
        s.close()
        res["cases"].append({"batch": B, "threads": threads, "ms_per_batch": round(best * 1e3, 4), "ns_per_item": round(best * 1e9 / B, 1)})
        print("batch %5d  threads %d: %.3f ms per batch, %.1f ns per item" % (B, threads, best * 1e3, best * 1e9 / B), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
