import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import videovector_amd as vv
from tests.test_gpu_parity import make_case, rel_fro
B, C, Nn, F, D = 32, 5, 4, 256, 128
ds, table, idx, W, b = make_case(11, 40, B, C, Nn, F, D, wstd=0.01)
engs = []
for mode in (0, 1):
    e = vv.Engine(0, "f16"); e.set_dedup(mode); e.table_set(table); e.params_set(W, b); engs.append(e)
rng = np.random.default_rng(0)
for it in range(4):
    idx = rng.integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
    cfg = vv.StepConfig(B, C, Nn, lr=0.05, momentum=0.9, weight_decay=5e-4)
    # teacher forcing: the dedup engine restarts every step from the dense engine's state
    engs[1].params_set(*engs[0].params_get())
    outs = []
    for e in engs:
        e.forward_backward(cfg, idx)
        bl = e.blobs(cfg, ip1_diff=True)
        dW, db = e.grads()
        outs.append((bl, dW.copy(), db.copy(), e.loss(), e.dedup_stats()))
        e.apply_update(cfg)
    a, d = outs
    print(it, "stats", d[4], "ip2 eq", np.array_equal(a[0]["ip2"], d[0]["ip2"]), "dy eq", np.array_equal(a[0]["ip1_diff"], d[0]["ip1_diff"]),
          "dW rel %.3e" % rel_fro(d[1], a[1]), "db rel %.3e" % rel_fro(d[2], a[2]))
    print("   W rel after update %.3e" % rel_fro(engs[1].params_get()[0], engs[0].params_get()[0]))
