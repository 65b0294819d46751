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
    outs = []
    for e in engs:
        e.forward_backward(cfg, idx)
        bl = e.blobs(cfg, ip1_diff=True)
        dW, db = e.grads()
        outs.append((bl, dW.copy(), db.copy(), e.loss(), e.dedup_stats()))
        e.apply_update(cfg)
    a, d = outs
    print(it, "stats", d[4], "ip2 eq", np.array_equal(a[0]["ip2"], d[0]["ip2"]), "dy eq", np.array_equal(a[0]["ip1_diff"], d[0]["ip1_diff"]),
          "dW rel %.3e" % rel_fro(d[1], a[1]), "db rel %.3e" % rel_fro(d[2], a[2]), "loss", a[3], d[3])
    Wa, Wd = engs[0].params_get()[0], engs[1].params_get()[0]
    print("   W rel %.3e" % rel_fro(Wd, Wa))

print("---- exactness at step 0 against fp64 dY^T X from the GPU's own dY")
from tests.test_gpu_parity import round_table
for (B, C, Nn, F, D, nv) in [(32, 5, 4, 256, 128, 40), (96, 5, 20, 384, 320, 6), (256, 5, 50, 1024, 512, 60)]:
    ds, table, idx, W, b = make_case(11, nv, B, C, Nn, F, D, wstd=0.01)
    res = {}
    for mode in (0, 1):
        e = vv.Engine(0, "f16"); e.set_dedup(mode); e.table_set(table); e.params_set(W, b)
        cfg = vv.StepConfig(B, C, Nn)
        e.forward_backward(cfg, idx)
        dY = e.blobs(cfg, ip1_diff=True)["ip1_diff"].astype(np.float64)        # [CN*B][D] channel-major rows
        rows = idx.T.reshape(-1)                                              # channel-major
        X = round_table(table, "f16")[rows].astype(np.float64)
        exact = dY.T @ X
        dW = e.grads()[0]
        res[mode] = rel_fro(dW, exact)
    print((B, C, Nn, F, D), "dense vs exact %.3e   dedup vs exact %.3e" % (res[0], res[1]))

print("---- against the oracle (fp32 dY) on rounded operands")
from oracle import oracle
from tests.test_gpu_parity import round_operand
for (B, C, Nn, F, D, nv) in [(32, 5, 4, 256, 128, 40), (96, 5, 20, 384, 320, 6), (256, 5, 50, 1024, 512, 60)]:
    ds, table, idx, W, b = make_case(11, nv, B, C, Nn, F, D, wstd=0.01)
    q = oracle.forward_backward(round_table(table, "f16"), idx, round_operand(W, "f16"), b, C_=C, Nn=Nn, want=("dW", "db", "dY"))
    res = {}
    for mode in (0, 1):
        e = vv.Engine(0, "f16"); e.set_dedup(mode); e.table_set(table); e.params_set(W, b)
        cfg = vv.StepConfig(B, C, Nn)
        e.forward_backward(cfg, idx)
        res[mode] = (rel_fro(e.grads()[0], q["dW"]), rel_fro(e.blobs(cfg, ip1_diff=True)["ip1_diff"], q["dY"]))
    print((B, C, Nn, F, D), "dense dw_q %.3e dy_q %.3e   dedup dw_q %.3e dy_q %.3e" % (res[0] + res[1]))
