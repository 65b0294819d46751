"""Debug session: gradient error per iteration (same W on both sides) with a small negative buffer, with and without a
negative dataset, dedup on/off."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import videovector_amd as vv
from oracle import oracle
from videovector_amd.synth import init_weights
oracle.build()


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


rng = np.random.default_rng(9)
B, C, Nn, F, D = 16, 5, 3, 96, 32
ns = rng.integers(3, 40, 31); vid = 1000 + 7 * np.arange(31); rb = np.concatenate([[0], np.cumsum(ns[:-1])])
table = (rng.integers(0, 32, (int(ns.sum()), F)) / 8).astype(np.float32)
nns = rng.integers(3, 40, 12); nvid = 1000 + 7 * np.arange(12); nrb = len(table) + np.concatenate([[0], np.cumsum(nns[:-1])])
ntable = (rng.integers(0, 32, (int(nns.sum()), F)) / 8).astype(np.float32)
mb = int(nns[:7].sum())
full = np.concatenate([table, ntable])
for dedup in ("1", "0"):
    os.environ["VV_DEDUP"] = dedup
    for neg in (False, True):
        kw = dict(batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=mb, negative_swap_percentage=50)
        if neg:
            kw["negatives"] = (nvid, nns, nrb)
        smp = oracle.Sampler(vid, ns, rb, **kw)
        eng = vv.Engine(0)
        eng.table_set(full)
        W, b = init_weights(4, D, F, std=0.02)
        cfg = vv.StepConfig(B, C, Nn)
        for it in range(4):
            idx = smp.next()[0]
            eng.params_set(W, b)
            eng.forward_backward(cfg, idx)
            dW, db = eng.grads()
            l, _ = eng.loss()
            Wh = W.astype(np.float16).astype(np.float32)
            r = oracle.forward_backward(full, idx, Wh, b, C_=C, Nn=Nn, want=("dW", "db"))
            print("dedup", dedup, "negds", neg, "it", it, "loss rel %.2e dW rel %.2e db rel %.2e rows>=main %d distinct %d"
                  % (abs(l - r["loss"]) / r["loss"], rel(dW, r["dW"]), rel(db, r["db"]), int((idx >= len(table)).sum()),
                     len(np.unique(idx))), flush=True)
            W = W - 0.5 * r["dW"]; b = b - 0.5 * r["db"]
        del eng

print("---- free-running trajectories, product sampler with prefetch, solver hyper-parameters of the facade test")
for dedup in ("1", "0"):
    os.environ["VV_DEDUP"] = dedup
    for neg in (False, True):
        kw = dict(batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=mb, negative_swap_percentage=50)
        if neg:
            kw["negatives"] = (nvid, nns, nrb)
        smp = oracle.Sampler(vid, ns, rb, **kw)
        ps = vv.Sampler(vid, ns, rb, **kw)
        ps.prefetch_start(depth=4, threads=3)
        eng = vv.Engine(0)
        eng.table_set(full)
        W0, b0 = init_weights(4, D, F, std=0.02)
        eng.params_set(W0, b0)
        Wq, bq = W0.copy(), b0.copy(); hW, hb = np.zeros_like(W0), np.zeros_like(b0)
        cfg = vv.StepConfig(B, C, Nn, momentum=0.9, weight_decay=5e-4, lr_mult=(1.0, 2.0), decay_mult=(1.0, 0.0))
        for it in range(6):
            idx = smp.next()[0]
            idx2 = ps.next()
            assert np.array_equal(idx, idx2)
            lr = oracle.learning_rate("inv", 0.01, 1e-3, 0.75, 0, it)
            cfg.set("lr", lr)
            eng.forward_backward(cfg, idx2)
            l, _ = eng.loss()
            eng.apply_update(cfg)
            r = oracle.forward_backward(full, idx, Wq.astype(np.float16).astype(np.float32), bq, C_=C, Nn=Nn, want=("dW", "db"))
            oracle.sgd_update(Wq, r["dW"], hW, lr, 1.0, 0.9, 5e-4, 1.0)
            oracle.sgd_update(bq, r["db"], hb, lr, 2.0, 0.9, 5e-4, 0.0)
            Wg, bg, _, _ = eng.params_get()
            print("dedup", dedup, "negds", neg, "it", it, "loss rel %.2e  W %.2e  b %.2e  dWtot %.2e"
                  % (abs(l - r["loss"]) / r["loss"], rel(Wg, Wq), rel(bg, bq), rel(Wg - W0, Wq - W0)), flush=True)
        ps.close()
        del eng
