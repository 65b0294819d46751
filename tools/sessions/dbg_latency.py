"""How much of the GEMM time is HBM-gather latency?  Times fwd/wgrad (dense path) on the cfg-2 shape with
(a) the sampled batch, (b) every slot naming one row (A operand always L2-hot), (c) 256 distinct rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos, init_weights
B, C, Nn, F, D = 1024, 5, 50, 4096, 512
ds = SyntheticVideos(seed=1701, n_videos=2048)
W, b = init_weights(1, D, F)
e = vv.Engine(0, "f16"); e.set_dedup(0); e.table_synth(ds.seed, ds.n_rows, F); e.params_set(W, b)
smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=5000, negative_swap_percentage=50)
cfg = vv.StepConfig(B, C, Nn)
rng = np.random.default_rng(0)
cases = {"sampled": smp.next()[0] if isinstance(smp.next(), tuple) else smp.next(),
         "one row": np.full((B, C + Nn), 1234, np.int32),
         "256 rows": rng.integers(0, 256, (B, C + Nn)).astype(np.int32),
         "4096 rows": rng.integers(0, 4096, (B, C + Nn)).astype(np.int32),
         "all random": rng.integers(0, ds.n_rows, (B, C + Nn)).astype(np.int32)}
for name, idx in cases.items():
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    for _ in range(3): e.forward_backward(cfg, idx)
    e.profile_enable(True)
    for _ in range(10): e.forward_backward(cfg, idx)
    print("%-12s fwd %.4f ms  wgrad %.4f ms  score %.4f" % (name, e.profile_get("fwd_gemm")[0], e.profile_get("wgrad_gemm")[0], e.profile_get("score_loss")[0]))
    e.profile_enable(False)
