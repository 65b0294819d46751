"""Where the CPU baseline (oracle step at the bench size) spends its time on the GPU box's host: ORC_TIMING phases."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["ORC_TIMING"] = "1"
from oracle import oracle as orc
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos, init_weights
orc.build()
B, C, NN, F, D = 1024, 5, 50, 4096, 512
ds = SyntheticVideos(seed=1701, n_videos=2048)
smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=NN)
idx = smp.next()
uniq, inv = np.unique(idx.reshape(-1), return_inverse=True)
table = ds.table(F, uniq)
il = inv.reshape(idx.shape).astype(np.int32)
W, b = init_weights(1701, D, F)
for th in ([int(x) for x in sys.argv[1:]] or [0]):
    orc.set_threads(th)
    for it in range(2):
        t0 = time.perf_counter()
        r = orc.forward_backward(table, il, W, b, C_=C, Nn=NN, want=("dW", "db"))
        print("threads %d iteration %d: %.3f s" % (th, it, time.perf_counter() - t0), file=sys.stderr, flush=True)
    print(orc.gemm_gflops(B * (C + NN), D, F), file=sys.stderr)
