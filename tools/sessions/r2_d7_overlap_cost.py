"""Cost of the data-parallel schedules on ONE rank over real RCCL (world 1: the collective itself is a device copy): what the
chunked weight gradient of --allreduce overlap costs against the whole-buffer all-reduce of --allreduce sync."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos, init_weights

B, C, NN, F, D = 1024, 5, 50, 4096, 512
ds = SyntheticVideos(seed=1701, n_videos=2048)
smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=NN, max_buffer_size=5000, negative_swap_percentage=50)
batches = np.stack([smp.next() for _ in range(64)])
dev = torch.device("cuda", 0)
idx = torch.from_numpy(batches).to(dev)
W0, b0 = init_weights(1701, D, F)
for mode in ("none", "sync", "overlap", "none", "sync", "overlap"):
    eng = vv.Engine(0, "f16")
    eng.table_synth(ds.seed, ds.n_rows, F)
    eng.params_set(W0, b0)
    cfg = vv.StepConfig(B, C, NN, global_count=B * NN)
    if mode != "none":
        p = "/tmp/vv_ovl_%d_%s" % (os.getpid(), mode)
        if os.path.exists(p): os.unlink(p)
        eng.comm_init(1, 0, p, "rccl")
        eng.comm_overlap(mode == "overlap")
    stride = B * (C + NN) * 4
    def run(n):
        for i in range(n):
            eng.forward_backward(cfg, idx_dev_ptr=idx.data_ptr() + (i % 64) * stride)
            eng.apply_update(cfg)
    run(400); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(2000); torch.cuda.synchronize()
    print("%-8s %.4f ms per step" % (mode, (time.perf_counter() - t0) / 2000 * 1e3), flush=True)
    eng.close()
