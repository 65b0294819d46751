"""Where the CPU baseline's time goes on the GPU box's host, by thread count and binding.
One child process per (binding, threads): OMP_PROC_BIND / OMP_PLACES are read when the oracle's OpenMP runtime loads.
Usage: python tools/lab/cpu_baseline_scaling.py            (parent: runs the grid)
       python tools/lab/cpu_baseline_scaling.py child T     (one configuration; ORC_TIMING=1 prints the phases of the last iteration)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def child(threads):
    import numpy as np
    from oracle import oracle as orc
    from videovector_amd.synth import SyntheticVideos, init_weights
    B, C, NN, F, D = 1024, 5, 50, 4096, 512
    ds = SyntheticVideos(seed=1701, n_videos=2048)
    smp = orc.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=NN, max_buffer_size=5000, negative_swap_percentage=50)
    idx = smp.next()[0]
    uniq, inv = np.unique(idx.reshape(-1), return_inverse=True)
    table = ds.table(F, uniq); il = inv.reshape(idx.shape).astype(np.int32)
    W, b = init_weights(1701, D, F)
    orc.set_threads(threads)
    hW, hb = np.zeros_like(W), np.zeros_like(b)
    def one():
        t0 = time.perf_counter()
        r = orc.forward_backward(table, il, W, b, C_=C, Nn=NN, want=("dW", "db"))
        orc.sgd_update(W, r["dW"], hW, 1e-3, 1.0, 0.9, 5e-4, 1.0)
        orc.sgd_update(b, r["db"], hb, 1e-3, 2.0, 0.9, 5e-4, 0.0)
        return time.perf_counter() - t0
    one(); one()
    ts = [one() for _ in range(3)]
    os.environ["ORC_TIMING"] = "1"
    sys.stderr.flush()
    one()
    print("RESULT bind=%s places=%s threads=%d: %.3f s per iteration (min %.3f) = %.0f k triplets/s" % (
        os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), threads, sum(ts) / len(ts), min(ts), B * NN / (sum(ts) / len(ts)) / 1e3), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
    else:
        ncpu = os.cpu_count()
        print("host logical CPUs:", ncpu, flush=True)
        for bind in (None, ("spread", "cores"), ("close", "cores")):
            for t in (32, 64, 128, 256):
                if t > ncpu: continue
                env = dict(os.environ)
                if bind: env.update(OMP_PROC_BIND=bind[0], OMP_PLACES=bind[1])
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(t)], env=env, capture_output=True, text=True, timeout=600)
                print(r.stdout.strip()[-300:], flush=True)
                ph = [l for l in r.stderr.splitlines() if l.startswith("[orc]")]
                if ph: print("   " + " | ".join(l[6:].strip().replace("  ", " ") for l in ph[-14:]), flush=True)
