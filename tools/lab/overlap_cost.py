"""Cost of the data-parallel schedules on ONE rank over real RCCL (the collective itself moves nothing): BASELINE configs[1]
steps with resident indices -- no communicator / sync (whole-buffer all-reduce in the compute stream) / overlap (update
F-chunk by F-chunk on the communication stream, next forward GEMM gated per chunk) / overlap with the communication
stream held D us in front of every chunk (VV_COMM_TEST_DELAY_US: what the gates cost when the exchange is really slow).
Usage: python tools/lab/overlap_cost.py [steps]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def run(mode, steps):
    import numpy as np, torch
    import videovector_amd as vv
    from videovector_amd.synth import SyntheticVideos, init_weights
    B, C, Nn, F, D = 1024, 5, 50, 4096, 512
    ds = SyntheticVideos(seed=1701, n_videos=2048)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=5000, negative_swap_percentage=50)
    n = 150 + max(steps, 200)
    idx = torch.from_numpy(np.stack([smp.next() for _ in range(n)])).to("cuda:0")
    torch.cuda.synchronize()
    W, b = init_weights(1701, D, F)
    eng = vv.Engine(0, "f16"); eng.table_synth(ds.seed, ds.n_rows, F); eng.params_set(W, b)
    if mode != "none":
        eng.comm_init(1, 0, "/tmp/vv_overlap_cost_%d" % os.getpid(), "rccl")
        if mode.startswith("sharded"): eng.comm_schedule("sharded")
        else: eng.comm_overlap(mode.startswith("overlap"))
    cfg = vv.StepConfig(B, C, Nn)
    stride = B * (C + Nn) * 4
    def go(a, bnd):
        for i in range(a, bnd):
            eng.forward_backward(cfg, idx_dev_ptr=idx.data_ptr() + i * stride, idx_ready=True)
            eng.apply_update(cfg)
    go(0, 150); eng.synchronize()
    t0 = time.perf_counter(); go(150, n); eng.synchronize(); el = time.perf_counter() - t0
    eng.profile_enable(8)
    go(150, 150 + 200); eng.synchronize()
    prof = {k: eng.profile_get(k)[0] for k in ("fwd_gemm", "score_loss", "segsum", "guard", "wgrad_gemm", "reduce", "sgd", "reduce_sgd")}
    eng.profile_enable(False)
    print("   kernels (us): " + "  ".join("%s %.1f" % (k, v * 1e3) for k, v in prof.items()), flush=True)
    print("%-22s %.4f ms/step  loss %.6f" % (mode + (" delay " + os.environ["VV_COMM_TEST_DELAY_US"] if os.environ.get("VV_COMM_TEST_DELAY_US") else ""), el / steps * 1e3, eng.loss()[0]), flush=True)
    eng.close()

if __name__ == "__main__":
    if len(sys.argv) > 2:
        run(sys.argv[1], int(sys.argv[2]))
    else:
        steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
        for rep in range(2):
            for mode, env in (("none", {}), ("sync", {}), ("overlap", {}), ("overlap_chunks2", {"VV_COMM_CHUNKS": "2"}), ("overlap_chunks4", {"VV_COMM_CHUNKS": "4"}),
                              ("sharded", {}), ("sharded_comm_stream", {"VV_COMM_INLINE": "0"}), ("sharded", {"VV_COMM_TEST_DELAY_US": "60"}), ("sharded_comm_stream", {"VV_COMM_INLINE": "0", "VV_COMM_TEST_DELAY_US": "60"}), ("overlap", {"VV_COMM_TEST_DELAY_US": "20"}), ("overlap_chunks2", {"VV_COMM_CHUNKS": "2", "VV_COMM_TEST_DELAY_US": "30"}), ("overlap_nogate", {"VV_COMM_GATE": "0"})):
                subprocess.run([sys.executable, os.path.abspath(__file__), mode, str(steps)], env=dict(os.environ, **env))
