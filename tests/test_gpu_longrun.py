"""Long-horizon parity: 2000 iterations of the HIP path (f16 operands, row de-duplication on) beside
the fp32 oracle's own trajectory -- what agreement on rounded operands (tests/test_gpu_parity.py) means for the trained
MODEL.  Cadence as Solver::Solve (src/caffe/solver.cpp:159-240): every iteration ForwardBackward + ComputeUpdateValue +
Update on the next batch of the reference sampler, shipped solver schedule (mednet_embedding_train_solver.prototxt:
base_lr 1e-3, inv policy, momentum .9, weight decay 5e-4).

What can be asserted.  This training problem amplifies perturbations: all embeddings share a large common component
(cos ~ 0.9999), the normalisation layers divide by small differences, hinge terms switch on and off.  Measured with the
oracle ALONE (tools/lab/trajectory_sensitivity.py): the fp32 oracle against itself with W perturbed by 1e-7 relative -- the
size of fp32 rounding, what another BLAS's summation order does to the reference itself -- differs by 3e-3 in the loss
after 100 free iterations; against itself with W rounded to f16 before each forward by 1e-3 after 10 and 9e-2 after
100.  So "per-iteration loss within 1e-3 over 100 free iterations" cannot hold for ANY implementation that differs from
the oracle in the last bits.  The test therefore runs three trajectories on the same batches -- the HIP path, the fp32
oracle, and the oracle on f16-rounded weights (the ENVELOPE: the same arithmetic perturbed exactly as the MFMA operand
rounding perturbs it) -- restarts the two perturbed ones from the oracle's state every 100 iterations (teacher forcing),
and asserts
  * per-iteration loss within 1e-3 of the fp32 oracle while the trajectories are still comparable (<= 3 free iterations),
  * at every distance from the last common state the HIP path is no further from the fp32 oracle than 3 x the envelope
    (+ 1e-3): it behaves like the reference under a perturbation of the size of its operand rounding, no worse,
  * the final models (100 free iterations after the last common state) agree through the TEST branch's retrieval
    statistics (retrieval_stats_layer.cpp:104-141): mAP within 5e-3; hit@1 / hit@5 (counts over 1000 queries) within 0.03."""
import numpy as np
import pytest

from videovector_amd.synth import SyntheticVideos, init_weights

pytestmark = pytest.mark.gpu


def lr_at(it):            # mednet_embedding_train_solver.prototxt:4-9: base_lr 1e-3, inv policy, gamma 1e-3, power .75
    return 1e-3 * (1.0 + 1e-3 * it) ** -0.75


@pytest.mark.parametrize("shape", ["cfg1", "d512"])
def test_2000_iterations_beside_the_fp32_oracle(oracle, shape):
    import videovector_amd as vv
    # BASELINE configs[0] shapes: 1k synthetic frames, 128-d -> 32-d, batch 32, 2 negatives; "d512": the same data
    # through the kernels the benchmark runs (segment-wise backward, D = 512), 10 negatives
    B, C, Nn, F, D = (32, 5, 2, 128, 32) if shape == "cfg1" else (32, 5, 10, 128, 512)
    ds = SyntheticVideos(seed=1701, n_videos=50)
    table = ds.table(F)
    kw = dict(batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=500, negative_swap_percentage=50)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    W0, b0 = init_weights(1, D, F, std=0.02)
    eng = vv.Engine(0, "f16")
    eng.table_set(table); eng.params_set(W0, b0); eng.set_dedup(True)
    cfg = vv.StepConfig(B, C, Nn)
    Wo, bo = W0.copy(), b0.copy(); hW, hb = np.zeros_like(Wo), np.zeros_like(bo)          # the fp32 oracle
    We, be = W0.copy(), b0.copy(); hWe, hbe = np.zeros_like(We), np.zeros_like(be)        # the envelope: oracle on f16-rounded W
    n_iter, seg = 2000, 100
    d_hip, d_env = np.zeros(seg), np.zeros(seg)
    first_loss = last_loss = None
    for it in range(n_iter):
        idx = smp.next()
        lr = lr_at(it)
        if it % seg == 0 and it > 0:                                 # teacher forcing: both restart from the oracle's state
            eng.params_set(Wo, bo, hW, hb)
            We[:] = Wo; be[:] = bo; hWe[:] = hW; hbe[:] = hb
        cfg.set("lr", lr)
        eng.step(cfg, idx)
        loss_g, _ = eng.loss()
        ref = oracle.forward_backward(table, idx, Wo, bo, C_=C, Nn=Nn, want=("dW", "db"))
        env = oracle.forward_backward(table, idx, We.astype(np.float16).astype(np.float32), be, C_=C, Nn=Nn, want=("dW", "db"))
        oracle.sgd_update(Wo, ref["dW"], hW, lr, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(bo, ref["db"], hb, lr, 2.0, 0.9, 5e-4, 0.0)
        oracle.sgd_update(We, env["dW"], hWe, lr, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(be, env["db"], hbe, lr, 2.0, 0.9, 5e-4, 0.0)
        k = it % seg
        d_hip[k] = max(d_hip[k], abs(loss_g - ref["loss"]) / abs(ref["loss"]))
        d_env[k] = max(d_env[k], abs(env["loss"] - ref["loss"]) / abs(ref["loss"]))
        if it == 0:
            first_loss = ref["loss"]
        last_loss = ref["loss"]
    Wg, bg, _, _ = eng.params_get()
    moved = np.linalg.norm(Wo - W0) / np.linalg.norm(W0)
    w_hip = np.linalg.norm(Wg - Wo) / np.linalg.norm(Wo)
    w_env = np.linalg.norm(We - Wo) / np.linalg.norm(Wo)
    # final models through the TEST branch: window-of-4 mean -> fc7 -> ReLU -> normalise -> within-batch retrieval statistics
    rng = np.random.default_rng(5)
    n = 1000
    vid = rng.integers(0, ds.n_videos, n).astype(np.int32)
    start = (rng.random(n) * (ds.n_shots[vid] - 4)).astype(np.int64)
    rows = ((ds.row_base[vid] + start)[:, None] + np.arange(4)[None, :]).astype(np.int32)
    emb_g = eng.embed_mean(rows, None, relu=True, l2norm=True)
    mean_rows = table[rows].mean(1).astype(np.float32)
    emb_o = oracle.embed(mean_rows, None, Wo, bo, relu=True, l2norm=True)
    cls = {int(v): int(v % 5) + 1 for v in range(ds.n_videos)}
    s_g = np.array(eng.retrieval_stats(emb_g, vid, cls))
    s_o = np.array(oracle.retrieval_stats(emb_o, vid, cls))
    repeats, _ = eng.grad_scale_stats()
    ks = (0, 1, 2, 3, 5, 10, 20, 30, 50, 99)
    print("LONGRUN %s: oracle loss %.4f -> %.4f, |W - W0| / |W0| = %.2f; guard repeats %d" % (shape, first_loss, last_loss, moved, repeats))
    print("LONGRUN %s: free iterations       %s" % (shape, " ".join("%8d" % k for k in ks)))
    print("LONGRUN %s: HIP vs fp32 oracle    %s" % (shape, " ".join("%8.1e" % d_hip[k] for k in ks)))
    print("LONGRUN %s: envelope (f16 W)      %s" % (shape, " ".join("%8.1e" % d_env[k] for k in ks)))
    print("LONGRUN %s: final W after 100 free iterations: HIP %.2e, envelope %.2e from the oracle's; retrieval (mAP, hit@1, hit@5) "
          "HIP %s oracle %s" % (shape, w_hip, w_env, s_g.round(4), s_o.round(4)))
    assert moved > 0.05, "the run did not train"
    assert d_hip[:4].max() <= 1e-3, d_hip[:4]
    env_run = np.maximum.accumulate(d_env)
    bad = [k for k in range(seg) if d_hip[k] > 3.0 * env_run[k] + 1e-3]
    assert not bad, "further from the oracle than 3 x the operand-rounding envelope at %s free iterations" % bad[:5]
    assert abs(s_g[0] - s_o[0]) <= 5e-3 and np.abs(s_g[1:] - s_o[1:]).max() <= 0.03
    smp.close(); eng.close()
