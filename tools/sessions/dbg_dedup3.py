import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import videovector_amd as vv
from tests.test_gpu_parity import make_case, rel_fro
B, C, Nn, F, D = 32, 5, 4, 256, 128
ds, table, idx, W, b = make_case(11, 40, B, C, Nn, F, D, wstd=0.01)
for eps in (0.0, 1e-6, 1e-5, 1e-4):
    engs = []
    for k in range(2):
        e = vv.Engine(0, "f16"); e.set_dedup(0); e.table_set(table)
        Wk = W if k == 0 else (W * (1 + eps * np.random.default_rng(5).standard_normal(W.shape))).astype(np.float32)
        e.params_set(Wk, b); engs.append(e)
    rng = np.random.default_rng(0)
    line = []
    for it in range(4):
        idx = rng.integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
        cfg = vv.StepConfig(B, C, Nn, lr=0.05, momentum=0.9, weight_decay=5e-4)
        g = []
        for e in engs:
            e.forward_backward(cfg, idx); g.append(e.grads()[0].copy()); e.apply_update(cfg)
        line.append("dW %.2e W %.2e" % (rel_fro(g[1], g[0]), rel_fro(engs[1].params_get()[0], engs[0].params_get()[0])))
    print("dense vs dense perturbed eps=%g: " % eps + " | ".join(line))
