"""(lab) the pipelined score kernel against the product kernel on the benchmark's batch: which outputs differ, and both kernels' durations.
Usage: VV_LIB=videovector_amd/lib/libvideovec_lab.so python tools/lab/score_pipe_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, subprocess

def run(pipe):
    import videovector_amd as vv
    from videovector_amd.synth import SyntheticVideos, init_weights
    B, C, Nn, F, D = 1024, 5, 50, 4096, 512
    ds = SyntheticVideos(seed=1701, n_videos=2048)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=5000, negative_swap_percentage=50)
    idx = smp.next()
    W, b = init_weights(1701, D, F)
    e = vv.Engine(0, "f16"); e.table_synth(ds.seed, ds.n_rows, F); e.params_set(W, b)
    cfg = vv.StepConfig(B, C, Nn, lr=0.01)
    e.forward_backward(cfg, idx); e.forward_backward(cfg, idx)
    bl = e.blobs(cfg, ip1_diff=True)
    dW, db = e.grads()
    e.profile_enable(1)
    for _ in range(30): e.forward_backward(cfg, idx)
    e.synchronize()
    t = e.profile_get("score_loss")
    np.savez("/tmp/sp_%d.npz" % pipe, loss=np.array(e.loss()), ts=bl["target_score"], ns=bl["negative_scores"], dy=bl["ip1_diff"], dW=dW, db=db, t=np.array(t[0]))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(int(sys.argv[1]))
    else:
        for pipe in (0, 1):
            subprocess.run([sys.executable, os.path.abspath(__file__), str(pipe)], env=dict(os.environ, VV_LAB_SCORE_PIPE=str(pipe)), check=True)
        a, b = np.load("/tmp/sp_0.npz"), np.load("/tmp/sp_1.npz")
        print("score kernel ms: product %.4f  pipelined %.4f" % (float(a["t"]), float(b["t"])))
        for k in ("loss", "ts", "ns", "dy", "dW", "db"):
            x, y = a[k], b[k]
            nd = int((x != y).sum())
            print("%-5s differing elements %d of %d, max |d| %.3e" % (k, nd, x.size, float(np.abs(x.astype(np.float64) - y).max()) if nd else 0.0))
        if (a["ns"] != b["ns"]).any():
            bad = np.argwhere(a["ns"] != b["ns"])
            print("first differing (item, negative):", bad[:10].tolist(), "items affected:", len(set(bad[:, 0].tolist())))
