"""GPU parity tests: the HIP path (through the C ABI, include/videovec.h) against the CPU oracle
on identical inputs.  Tolerances (north-star: embeddings and loss within 1e-3 relative to the fp32
reference path; indices bit-exact):

  against the fp32 oracle (fp32 features, fp32 weights)
    f16 MFMA operands (default) : per-row relative L2 error of embeddings <= 1e-3, loss rel <= 1e-3,
                                   cosine scores abs <= 1e-3
    bf16 MFMA operands (opt-in) : 8-bit mantissas cannot reach 1e-3 on long dot products (DESIGN.md
                                   "Precision"): embeddings bounded at 1e-2, scores 5e-3
  gradients (ip1 diff, dW, db)
    d(loss)/d(embedding) is the component of the context vector ORTHOGONAL to the target/negative
    embedding.  With non-negative features all embeddings share a large common component at
    initialisation (cos ~ 0.999), so that component is a small difference of nearly equal vectors
    and any perturbation of the embeddings -- here the 2^-12 (f16) / 2^-9 (bf16) rounding of the
    weight copy the MFMA reads -- is amplified 10-100x in the gradient, in ANY implementation.
    The kernels are therefore checked (a) tightly against the oracle evaluated on the SAME rounded
    operands (tolerance 2e-3 f16, 1e-2 bf16: what is left is the 16-bit rounding of dY) and (b)
    against the fp32-operand oracle with the conditioning-limited bound 0.1 (f16) / 0.3 (bf16).
"""
import os

import numpy as np
import pytest

from videovector_amd.synth import SyntheticVideos, feature_rows, init_weights

pytestmark = pytest.mark.gpu

TOL = {"f16": dict(emb=1e-3, loss=1e-3, score=1e-3, grad_q=2e-3, grad=0.1),
       "bf16": dict(emb=1e-2, loss=1e-3, score=5e-3, grad_q=1e-2, grad=0.3)}


def round_operand(x, prec):
    """The value the MFMA reads for x: round-to-nearest-even to f16 after the product's power-of-two
    range scaling (max|x| * s in [2^11, 2^12)), or to bf16."""
    x = np.asarray(x, np.float32)
    if prec == "f16":
        m = float(np.abs(x).max())
        s = 2.0 ** (12 - np.frexp(m)[1]) if m > 0 else 1.0
        return ((x * np.float32(s)).astype(np.float16).astype(np.float32) / np.float32(s)).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def round_table(t, prec):
    if prec == "f16":       # the table keeps scale 1 while max|x| is inside [2^-8, 2^14]
        return t.astype(np.float16).astype(np.float32)
    return round_operand(t, "bf16")


def rel_rows(a, ref):
    den = np.maximum(np.linalg.norm(ref, axis=1), 1e-20)
    return float((np.linalg.norm(a - ref, axis=1) / den).max())


def rel_fro(a, ref):
    return float(np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30))


@pytest.fixture(scope="module")
def vv():
    import videovector_amd
    return videovector_amd


def make_case(seed, n_videos, B, C, Nn, F, D, wstd=1e-3, real=False):
    ds = SyntheticVideos(seed=seed, n_videos=n_videos)
    table = ds.table(F)
    if real:     # real-valued features: exercises the rounding of the table itself
        table = np.maximum(np.random.default_rng(seed).standard_normal(table.shape), 0).astype(np.float32) * 1.7
    rng = np.random.default_rng(seed + 1)
    idx = rng.integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
    W, b = init_weights(seed, D, F, std=wstd)
    b = (rng.standard_normal(D) * wstd).astype(np.float32)
    return ds, table, idx, W, b


def run_both(vv, oracle, prec, table, idx, W, b, C, Nn, dedup=None, **kw):
    eng = vv.Engine(0, prec)
    if dedup is not None:
        eng.set_dedup(dedup)
    eng.table_set(table)
    eng.params_set(W, b)
    B = idx.shape[0]
    cfg = vv.StepConfig(B, C, Nn, **{k: v for k, v in kw.items() if k != "last_src"})
    eng.forward_backward(cfg, idx)
    loss, viol = eng.loss()
    blobs = eng.blobs(cfg, ip1_diff=True)
    dW, db = eng.grads()
    okw = dict(C_=C, Nn=Nn, margin=cfg.c.margin, norm=cfg.c.norm, loss_weight=cfg.c.loss_weight,
               ctx_coeff=kw.get("ctx_coeff"), dropout_ratio=kw.get("dropout_ratio", 0.0),
               dropout_mask=kw.get("dropout_mask"), global_count=kw.get("global_count", 0),
               item_weight=kw.get("item_weight"), ip_regularization=kw.get("ip_regularization", 0.0),
               want=("H", "s_true", "s_bogus", "dY", "dW", "db"))
    ref = oracle.forward_backward(table, idx, W, b, **okw)
    okw["want"] = ("dY", "dW", "db")
    ref["q"] = oracle.forward_backward(round_table(table, prec), idx, round_operand(W, prec), b, **okw)
    return eng, cfg, dict(loss=loss, viol=viol, dW=dW, db=db, **blobs), ref


def check(got, ref, tol, tag=""):
    q = ref["q"]
    m = dict(
        emb=rel_rows(got["ip2"], ref["H"]),
        loss=abs(got["loss"] - ref["loss"]) / max(abs(ref["loss"]), 1e-30),
        score=max(np.abs(got["target_score"] - ref["s_true"]).max(),
                  np.abs(got["negative_scores"] - ref["s_bogus"]).max()),
        dy_q=rel_fro(got["ip1_diff"], q["dY"]), dw_q=rel_fro(got["dW"], q["dW"]),
        db_q=rel_fro(got["db"], q["db"]),
        dy=rel_fro(got["ip1_diff"], ref["dY"]), dw=rel_fro(got["dW"], ref["dW"]),
        db=rel_fro(got["db"], ref["db"]),
    )
    print("PARITY %s %s" % (tag, " ".join("%s=%.3e" % kv for kv in m.items())))
    for k in ("emb", "loss", "score"):
        assert m[k] <= tol[k], (k, m[k], tol[k])
    for k in ("dy_q", "dw_q", "db_q"):
        assert m[k] <= tol["grad_q"], (k, m[k], tol["grad_q"])
    for k in ("dy", "dw", "db"):
        assert m[k] <= tol["grad"], (k, m[k], tol["grad"])
    return m


def test_library_is_the_hip_build(vv):
    L = vv.load_library()
    assert b"gfx950" in L.vv_version()


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_table_synth_bit_exact_and_roundtrip(vv, prec):
    eng = vv.Engine(0, prec)
    eng.table_synth(1701, 300, 200)
    rows = np.array([0, 7, 299, 123], np.int32)
    assert np.array_equal(eng.table_get(rows), feature_rows(1701, rows, 200))
    t = feature_rows(5, np.arange(50), 72)
    eng2 = vv.Engine(0, prec)
    eng2.table_set(t)
    assert np.array_equal(eng2.table_get(n=50), t)          # synthetic values are exact in 16 bits


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("norm", ["L2", "L1"])
def test_config1_plumbing_shapes(vv, oracle, prec, norm):
    # BASELINE config 1: 1k frames, 128 -> 32, batch 32, 2 negatives
    ds, table, idx, W, b = make_case(3, 50, 32, 5, 2, 128, 32, wstd=0.02)
    _, _, got, ref = run_both(vv, oracle, prec, table, idx, W, b, 5, 2, norm=norm)
    check(got, ref, TOL[prec], "cfg1/%s/%s" % (prec, norm))
    assert got["viol"] == ref["violations"]


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_multi_tile_shapes(vv, oracle, prec):
    # several M / N / K tiles and split-K slabs: R = 64*16 = 1024 rows, 512 -> 512
    ds, table, idx, W, b = make_case(4, 40, 64, 5, 11, 512, 512)
    _, _, got, ref = run_both(vv, oracle, prec, table, idx, W, b, 5, 11)
    check(got, ref, TOL[prec], "multitile/%s" % prec)


@pytest.mark.parametrize("D", [2048, 1536, 3072])
@pytest.mark.parametrize("dedup", [False, True])
def test_wide_rows_small_batch_sixteen_wave_score_kernel(vv, oracle, dedup, D):
    """Few items of wide rows (D >= 2048, B <= 512) run k_score_loss with sixteen waves per item.  At D = 2048 its column-parallel
    backward phase has TWO row groups side by side, each with a [D] partial-sum area for dAh and one for db: ADVICE r4 found those areas
    sized for four waves (1024 floats) -- group 1 of dAh ran into group 0 of db and group 1 of db over the kernel's small arrays and past the
    allocation.  D = 1536 (four-wave form, two groups) and D = 3072 (one group) bracket the case."""
    B, C, Nn, F = 24, 3, 6, 256
    ds, table, idx, W, b = make_case(77 + D, 40, B, C, Nn, F, D)
    _, _, got, ref = run_both(vv, oracle, "f16", table, idx, W, b, C, Nn, dedup=dedup)
    check(got, ref, TOL["f16"], "wide-rows D=%d dedup=%s" % (D, dedup))
    assert got["viol"] == ref["violations"]


def test_ragged_shapes_and_empty_slots(vv, oracle):
    # D not a multiple of 4 (scalar store paths), F not a multiple of 64, idx == -1 (zero rows)
    ds, table, idx, W, b = make_case(5, 30, 7, 3, 4, 100, 30, wstd=0.05)
    idx[0, 3] = -1
    idx[6, 0] = -1
    t2 = np.concatenate([table, np.zeros((1, 100), np.float32)])
    idx_o = np.where(idx < 0, len(table), idx).astype(np.int32)
    eng = vv.Engine(0, "f16")
    eng.table_set(table); eng.params_set(W, b)
    cfg = vv.StepConfig(7, 3, 4)
    eng.forward_backward(cfg, idx)
    got = dict(loss=eng.loss()[0], **eng.blobs(cfg, ip1_diff=True))
    got["dW"], got["db"] = eng.grads()
    ref = oracle.forward_backward(t2, idx_o, W, b, C_=3, Nn=4,
                                  want=("H", "s_true", "s_bogus", "dY", "dW", "db"))
    ref["q"] = oracle.forward_backward(round_table(t2, "f16"), idx_o, round_operand(W, "f16"), b,
                                       C_=3, Nn=4, want=("dY", "dW", "db"))
    check(got, ref, TOL["f16"], "ragged")


def test_real_valued_features_f16(vv, oracle):
    # fp32 features that are NOT exactly representable in 16 bits: the table rounding counts too
    ds, table, idx, W, b = make_case(6, 40, 32, 5, 6, 1024, 256, real=True)
    _, _, got, ref = run_both(vv, oracle, "f16", table, idx, W, b, 5, 6)
    check(got, ref, TOL["f16"], "realfeat")


def test_wgrad_transposed_lds_reads_match_scalar_reads(vv):
    # the ds_read_b64_tr_b16 fragment loader against the layout-obvious 16-bit loader
    ds, table, idx, W, b = make_case(8, 40, 48, 5, 7, 512, 256)
    out = []
    for tr in ("1", "0"):
        os.environ["VV_WGRAD_TR"] = tr
        eng = vv.Engine(0, "f16")
        eng.table_set(table); eng.params_set(W, b)
        cfg = vv.StepConfig(48, 5, 7)
        eng.forward_backward(cfg, idx)
        out.append(eng.grads()[0])
    os.environ["VV_WGRAD_TR"] = "1"
    vv.Engine(0, "f16")
    assert np.array_equal(out[0], out[1])


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_dropout_with_explicit_mask_and_options(vv, oracle, prec):
    B, C, Nn, F, D = 16, 5, 3, 256, 64
    ds, table, idx, W, b = make_case(9, 30, B, C, Nn, F, D, wstd=0.01)
    mask = (np.random.default_rng(2).random(((C + Nn) * B, D)) > 0.5).astype(np.uint8)
    coeff = np.array([0.4, 0.3, 0.2, 0.1], np.float32)
    _, _, got, ref = run_both(vv, oracle, prec, table, idx, W, b, C, Nn, dropout_ratio=0.5,
                              dropout_mask=mask, ctx_coeff=coeff, loss_weight=0.7,
                              global_count=4 * B * Nn, margin=1.5)
    check(got, ref, TOL[prec], "dropout/%s" % prec)


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_dropout_on_the_deduplicated_path_matches_oracle(vv, oracle, prec, fp32_ip2):
    """drop2 sits behind fc7 + ReLU (mednet_embedding_train.prototxt:190-230), so equal table rows share their projection and only the mask
    is per instance: at D = 512 dropout rides the de-duplicated path (k_score_fwd masks every instance's row, k_seg_bwd sums the
    mask-weighted terms per distinct row).  Explicit mask, rows that repeat inside and across items, every blob against the oracle."""
    B, C, Nn, F, D = 16, 5, 7, 256, 512
    ds, table, idx, W, b = make_case(19, 30, B, C, Nn, F, D, wstd=0.01)
    rng = np.random.default_rng(3)
    idx = rng.integers(0, 60, size=(B, C + Nn)).astype(np.int32)          # 192 instances of <= 60 rows: segments of several instances
    mask = (rng.random(((C + Nn) * B, D)) > 0.6).astype(np.uint8)
    coeff = np.array([0.4, 0.3, 0.2, 0.1], np.float32)
    eng, _, got, ref = run_both(vv, oracle, prec, table, idx, W, b, C, Nn, dropout_ratio=0.6, dropout_mask=mask, ctx_coeff=coeff,
                                loss_weight=0.7, global_count=4 * B * Nn, margin=1.5)
    rows, uniq = eng.dedup_stats()
    assert rows == B * (C + Nn) and uniq == len(np.unique(idx)) < rows, "the de-duplicated path did not run"
    check(got, ref, TOL[prec], "dropout-dedup/%s" % prec)
    # the same step with the switch off: dense execution, same mask -> the same blobs to rounding
    eng2 = vv.Engine(0, prec)
    eng2.set_option("drop_dedup", 0)
    eng2.table_set(table); eng2.params_set(W, b)
    cfg = vv.StepConfig(B, C, Nn, dropout_ratio=0.6, dropout_mask=mask, ctx_coeff=coeff, loss_weight=0.7, global_count=4 * B * Nn, margin=1.5)
    eng2.forward_backward(cfg, idx)
    assert eng2.dedup_stats() == (rows, rows)
    b2 = eng2.blobs(cfg, ip1_diff=True)
    assert np.array_equal(b2["ip2"] != 0, got["ip2"] != 0) and np.allclose(b2["ip2"], got["ip2"], rtol=1e-5, atol=1e-6)      # (the dense epilogue folds 1 / (1 - ratio) into its descale: last-bit differences)
    assert abs(eng2.loss()[0] - got["loss"]) <= 1e-6 * got["loss"]
    assert rel_fro(eng2.grads()[0], got["dW"]) <= (4e-3 if prec == "bf16" else 2e-3)


def test_dropout_on_the_deduplicated_path_l1_weighted_pairwise(vv, oracle):
    """The same path under the options that change the backward's coefficients: L1 hinge, per-item loss weights (the weighted loss's third
    bottom), two channels in front of the negatives (C = 2: one context row, as PAIRWISE batches have), 30 negatives (the four-rows-per-wave form)."""
    B, C, Nn, F, D = 24, 2, 30, 256, 512
    ds, table, idx, W, b = make_case(29, 30, B, C, Nn, F, D, wstd=0.01)
    rng = np.random.default_rng(7)
    idx = rng.integers(0, 90, size=(B, C + Nn)).astype(np.int32)
    mask = (rng.random(((C + Nn) * B, D)) > 0.5).astype(np.uint8)
    iw = (0.5 + rng.random(B)).astype(np.float32)
    eng, _, got, ref = run_both(vv, oracle, "f16", table, idx, W, b, C, Nn, dropout_ratio=0.5, dropout_mask=mask, norm=1, item_weight=iw, margin=1.0)
    rows, uniq = eng.dedup_stats()
    assert uniq == len(np.unique(idx)) < rows
    check(got, ref, TOL["f16"], "dropout-dedup-l1w/f16")


def test_counter_based_dropout_dedup_equals_dense(vv, fp32_ip2):
    """Counter-hash masks (the product's own generator) at D = 512: the de-duplicated and the dense execution evaluate the same mask
    function, so they drop the same elements of every instance's row; loss equal to rounding, gradients to the reassociation of the sums."""
    B, C, Nn, F, D = 64, 5, 20, 512, 512
    ds, table, idx, W, b = make_case(23, 60, B, C, Nn, F, D, wstd=0.02)
    idx = np.random.default_rng(5).integers(0, 300, size=(B, C + Nn)).astype(np.int32)
    out = {}
    for dd in (1, 0):
        eng = vv.Engine(0, "f16")
        eng.set_option("drop_dedup", dd)
        eng.table_set(table); eng.params_set(W, b)
        cfg = vv.StepConfig(B, C, Nn, dropout_ratio=0.9, dropout_seed=4242)
        eng.forward_backward(cfg, idx)
        out[dd] = (eng.dedup_stats(), eng.loss(), eng.blobs(cfg)["ip2"], eng.grads())
    (st1, l1, h1, (dW1, db1)), (st0, l0, h0, (dW0, db0)) = out[1], out[0]
    assert st1[1] < st1[0] and st0[1] == st0[0]
    kept = h1 != 0
    assert np.array_equal(kept, h0 != 0) and 0.02 < kept.mean() < 0.12           # the same elements: ~10 % of the positive ones
    assert np.allclose(h1, h0, rtol=1e-5, atol=1e-6)
    assert abs(l1[0] - l0[0]) <= 1e-6 * l0[0] and l1[1] == l0[1]
    assert rel_fro(dW1, dW0) <= 2e-3 and rel_fro(db1, db0) <= 1e-4
    print("DROP-DEDUP rows %d distinct %d; loss %.6f / %.6f; dW %.2e db %.2e" % (st1 + (l1[0], l0[0], rel_fro(dW1, dW0), rel_fro(db1, db0))))


def test_counter_based_dropout_statistics(vv):
    # reference TestDropoutHalf (test_neuron_layer.cpp): kept fraction within 1.96 sigma of 1-p and
    # kept values scaled by 1/(1-p)
    B, C, Nn, F, D = 32, 5, 3, 128, 256
    ds, table, idx, W, b = make_case(10, 30, B, C, Nn, F, D, wstd=0.05)
    eng = vv.Engine(0, "f16")
    eng.table_set(table); eng.params_set(np.abs(W), np.ones(D, np.float32))
    base = vv.StepConfig(B, C, Nn)
    eng.forward_backward(base, idx)
    h0 = eng.blobs(base)["ip2"]
    cfg = vv.StepConfig(B, C, Nn, dropout_ratio=0.5, dropout_seed=77)
    eng.forward_backward(cfg, idx)
    h1 = eng.blobs(cfg)["ip2"]
    kept = h1 != 0
    n = h0.size
    assert abs(kept.sum() - 0.5 * n) <= 1.96 * np.sqrt(n * 0.25) * 1.5
    assert np.allclose(h1[kept], 2.0 * h0[kept], rtol=1e-6)


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_sgd_steps_match_oracle(vv, oracle, prec, fp32_slabs):
    # Four solver iterations (inv lr policy, momentum, L2 decay, lr_mult 1/2, decay_mult 1/0).
    # The oracle trajectory "q" takes each gradient at the operands the MFMA reads (rounded copy of
    # its own current W): weights, history and bias must then agree tightly.  The pure-fp32
    # trajectory differs by the conditioning-amplified gradient deviation (see module docstring).
    # This free-running comparison is made on the dense path: the case is chaotic (a 1e-6 relative change of
    # W at step 0 moves W by 5e-3 after four steps, tools/dbg_dedup3.py), so it only holds while the
    # summation order of the gradient is fixed; test_sgd_update_rule_teacher_forced covers both paths.
    B, C, Nn, F, D = 32, 5, 4, 256, 128
    ds, table, idx, W, b = make_case(11, 40, B, C, Nn, F, D, wstd=0.01)
    eng = vv.Engine(0, prec)
    eng.set_dedup(False)
    eng.table_set(table); eng.params_set(W, b)
    tq = round_table(table, prec)
    traj = {k: dict(W=W.copy(), b=b.copy(), hW=np.zeros_like(W), hb=np.zeros_like(b)) for k in "fq"}
    rng = np.random.default_rng(0)
    for it in range(4):
        lr = oracle.learning_rate("inv", 0.05, 1e-3, 0.75, 0, it)
        idx = rng.integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
        cfg = vv.StepConfig(B, C, Nn, lr=lr, momentum=0.9, weight_decay=5e-4)
        eng.step(cfg, idx)
        for k, t in traj.items():
            Wuse = round_operand(t["W"], prec) if k == "q" else t["W"]
            r = oracle.forward_backward(tq if k == "q" else table, idx, Wuse, t["b"], C_=C, Nn=Nn,
                                        want=("dW", "db"))
            oracle.sgd_update(t["W"], r["dW"], t["hW"], lr, 1.0, 0.9, 5e-4, 1.0)
            oracle.sgd_update(t["b"], r["db"], t["hb"], lr, 2.0, 0.9, 5e-4, 0.0)
            if k == "f":
                assert abs(eng.loss()[0] - r["loss"]) <= 1e-3 * abs(r["loss"])
    Wg, bg, hWg, hbg = eng.params_get()
    q, f = traj["q"], traj["f"]
    print("SGD %s vs q: W=%.3e hW=%.3e b=%.3e hb=%.3e | vs fp32: W=%.3e hW=%.3e" %
          (prec, rel_fro(Wg, q["W"]), rel_fro(hWg, q["hW"]), rel_fro(bg, q["b"]), rel_fro(hbg, q["hb"]),
           rel_fro(Wg, f["W"]), rel_fro(hWg, f["hW"])))
    # f16: tight.  bf16: the two trajectories round slightly different weights to 8 bits each step,
    # and those rounding flips are themselves amplified -- only a loose bound is meaningful.
    wtol, htol = (1e-3, 4e-3) if prec == "f16" else (2e-2, 6e-2)
    assert rel_fro(Wg, q["W"]) <= wtol and rel_fro(bg, q["b"]) <= 2 * wtol
    assert rel_fro(hWg, q["hW"]) <= htol and rel_fro(hbg, q["hb"]) <= htol
    assert rel_fro(Wg, f["W"]) <= TOL[prec]["grad"] and rel_fro(hWg, f["hW"]) <= TOL[prec]["grad"]


@pytest.mark.parametrize("dedup", [False, True])
@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_sgd_update_rule_teacher_forced(vv, oracle, prec, dedup):
    # The same four solver iterations, but every step starts from the oracle's state (weights, bias and both
    # momentum histories pushed into the engine): tests the fused update rule and the gradient of each step
    # without the chaotic compounding above, on the dense and on the de-duplicated path.
    B, C, Nn, F, D = 32, 5, 4, 256, 128
    ds, table, idx, W, b = make_case(11, 40, B, C, Nn, F, D, wstd=0.01)
    eng = vv.Engine(0, prec)
    eng.set_dedup(dedup)
    eng.table_set(table)
    tq = round_table(table, prec)
    t = dict(W=W.copy(), b=b.copy(), hW=np.zeros_like(W), hb=np.zeros_like(b))
    rng = np.random.default_rng(0)
    for it in range(4):
        lr = oracle.learning_rate("inv", 0.05, 1e-3, 0.75, 0, it)
        idx = rng.integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
        cfg = vv.StepConfig(B, C, Nn, lr=lr, momentum=0.9, weight_decay=5e-4)
        eng.params_set(t["W"], t["b"], t["hW"], t["hb"])
        eng.step(cfg, idx)
        r = oracle.forward_backward(tq, idx, round_operand(t["W"], prec), t["b"], C_=C, Nn=Nn, want=("dW", "db"))
        assert abs(eng.loss()[0] - r["loss"]) <= 1e-4 * abs(r["loss"])
        oracle.sgd_update(t["W"], r["dW"], t["hW"], lr, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(t["b"], r["db"], t["hb"], lr, 2.0, 0.9, 5e-4, 0.0)
        Wg, bg, hWg, hbg = eng.params_get()
        m = (rel_fro(Wg, t["W"]), rel_fro(hWg, t["hW"]), rel_fro(bg, t["b"]), rel_fro(hbg, t["hb"]))
        print("SGD-TF %s dedup=%d it %d: W=%.3e hW=%.3e b=%.3e hb=%.3e" % ((prec, dedup, it) + m))
        wtol, htol = (1e-4, 2e-3) if prec == "f16" else (1e-3, 2e-2)
        assert m[0] <= wtol and m[2] <= 2 * wtol and m[1] <= htol and m[3] <= htol


@pytest.mark.parametrize("solver,mom", [("NESTEROV", 0.9), ("ADAGRAD", 0.0), ("SGD", 0.9)])
def test_solver_types_and_ip_regularization_teacher_forced(vv, oracle, solver, mom):
    # NesterovSolver / AdaGradSolver (solver.cpp:599-655, 714-781) and InnerProduct.regularization
    # (inner_product_layer.cpp:80-90) in the fused update, each step started from the oracle's state; L1 decay.
    B, C, Nn, F, D = 32, 5, 4, 256, 128
    ds, table, idx, W, b = make_case(13, 40, B, C, Nn, F, D, wstd=0.02)
    eng = vv.Engine(0, "f16")
    eng.table_set(table)
    tq = round_table(table, "f16")
    # AdaGrad starts from a non-zero history: with h = 0 the first update is lr * sign(g), and a gradient element
    # within rounding of zero would flip a whole +-lr step
    h0 = 1e-4 if solver == "ADAGRAD" else 0.0
    t = dict(W=W.copy(), b=b.copy(), hW=np.full_like(W, h0), hb=np.full_like(b, h0))
    rng = np.random.default_rng(1)
    for it in range(4):
        lr = oracle.learning_rate("step", 0.05, 0.5, 0, 2, it)
        idx = rng.integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
        cfg = vv.StepConfig(B, C, Nn, lr=lr, momentum=mom, weight_decay=1e-3, solver_type=solver, delta=1e-6,
                            ip_regularization=0.5, reg="L1" if it % 2 else "L2")
        eng.params_set(t["W"], t["b"], t["hW"], t["hb"])
        eng.step(cfg, idx)
        r = oracle.forward_backward(tq, idx, round_operand(t["W"], "f16"), t["b"], C_=C, Nn=Nn, ip_regularization=0.5,
                                    want=("dW", "db"))
        g_gpu = eng.grads()[0]
        assert rel_fro(g_gpu, r["dW"]) <= 1e-3
        kw = dict(reg="L1" if it % 2 else "L2", solver=solver, delta=1e-6)
        oracle.sgd_update(t["W"], r["dW"], t["hW"], lr, 1.0, mom, 1e-3, 1.0, **kw)
        oracle.sgd_update(t["b"], r["db"], t["hb"], lr, 2.0, mom, 1e-3, 0.0, **kw)
        Wg, bg, hWg, hbg = eng.params_get()
        m = (rel_fro(Wg, t["W"]), rel_fro(hWg, t["hW"]), rel_fro(bg, t["b"]), rel_fro(hbg, t["hb"]))
        print("SOLVER %s it %d: W=%.3e hW=%.3e b=%.3e hb=%.3e" % ((solver, it) + m))
        # the update is O(lr) per element with AdaGrad, so the gradient's 2e-4 shows up in W undiminished
        assert m[0] <= (1e-3 if solver == "ADAGRAD" else 2e-4) and m[2] <= 4e-4 and m[1] <= 2e-3 and m[3] <= 2e-3
    cfg = vv.StepConfig(B, C, Nn, solver_type="ADAGRAD", momentum=0.9)
    with pytest.raises(vv.VVError, match="Momentum cannot be used with AdaGrad"):
        eng.step(cfg, idx)


@pytest.mark.parametrize("norm", ["L1", "L2"])
def test_weighted_loss_third_bottom(vv, oracle, norm):
    # MAX_MARGIN_LOSS with per-item term weights (max_margin_loss_layer.cpp:82-97, 152-186), zeros included
    B, C, Nn, F, D = 48, 5, 6, 256, 96
    ds, table, idx, W, b = make_case(17, 30, B, C, Nn, F, D, wstd=0.03)
    w = np.random.default_rng(3).uniform(0, 3, B).astype(np.float32)
    w[::7] = 0.0
    eng, cfg, got, ref = run_both(vv, oracle, "f16", table, idx, W, b, C, Nn, item_weight=w, norm=norm, margin=1.5)
    check(got, ref, TOL["f16"], "weighted-" + norm)
    plain = oracle.forward_backward(table, idx, W, b, C_=C, Nn=Nn, norm=2 if norm == "L2" else 1, margin=1.5)
    assert abs(plain["loss"] - ref["loss"]) > 1e-3 * ref["loss"]          # the weights do change the loss
    with pytest.raises(vv.VVError, match="All weights should be greater than 0"):
        eng.forward_backward(vv.StepConfig(B, C, Nn, item_weight=-w - 1), idx)


def test_q1_same_video_negatives(vv, oracle):
    # shipped setting max_same_video_negs: 6 -- the data layer copies those rows without their last
    # feature (video_sampled_shots_data_layer.cpp:492); indices from the product sampler
    B, C, Nn, F, D = 16, 5, 10, 256, 64
    ds = SyntheticVideos(seed=13, n_videos=60)
    table = ds.table(F)
    W, b = init_weights(13, D, F, std=0.01)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                     max_buffer_size=300, negative_swap_percentage=50, max_same_video_negs=6)
    eng = vv.Engine(0, "f16")
    eng.table_set(table); eng.params_set(W, b)
    cfg = vv.StepConfig(B, C, Nn)
    for _ in range(3):            # later batches inherit last features from earlier ones
        idx, last, _ = smp.next(want_last=True, want_label=True)
    assert (idx != last).sum() >= B
    eng.forward_backward_q1(cfg, idx, last)
    got = dict(loss=eng.loss()[0], **eng.blobs(cfg, ip1_diff=True))
    got["dW"], got["db"] = eng.grads()
    kw = dict(C_=C, Nn=Nn, last_src=last)
    ref = oracle.forward_backward(table, idx, W, b, want=("H", "s_true", "s_bogus", "dY", "dW", "db"), **kw)
    ref["q"] = oracle.forward_backward(round_table(table, "f16"), idx, round_operand(W, "f16"), b,
                                       want=("dY", "dW", "db"), **kw)
    check(got, ref, TOL["f16"], "q1")
    # and it differs from ignoring the quirk
    eng.forward_backward(cfg, idx)
    assert abs(eng.loss()[0] - got["loss"]) > 0


def test_pipelined_trainer_single_gpu_matches_delayed_gradient_oracle(vv, oracle, fp32_slabs):
    # the schedule bench.py uses for N > 1 (videovector_amd/dist.py PipelinedTrainer), run with world 1:
    # gradients of iteration t+1 are taken before the update with g_t is applied
    from videovector_amd.dist import GpuBackend, PipelinedTrainer
    B, C, Nn, F, D = 32, 5, 4, 256, 128
    ds, table, idx0, W, b = make_case(17, 40, B, C, Nn, F, D, wstd=0.01)
    eng = vv.Engine(0, "f16")
    eng.table_set(table); eng.params_set(W, b)
    cfg = vv.StepConfig(B, C, Nn, momentum=0.9, weight_decay=5e-4)
    tr = PipelinedTrainer(GpuBackend(eng, cfg), None, Nn, dist=None, rank=0, world=1)
    tq = round_table(table, "f16")
    Wo, bo, hW, hb = W.copy(), b.copy(), np.zeros_like(W), np.zeros_like(b)
    rng = np.random.default_rng(5)
    pending = None

    def apply(r):
        oracle.sgd_update(Wo, r["dW"], hW, 0.05, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(bo, r["db"], hb, 0.05, 2.0, 0.9, 5e-4, 0.0)
    for it in range(5):
        idx = rng.integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
        tr.step(0.05, idx_local=idx, global_batch=B)
        r = oracle.forward_backward(tq, idx, round_operand(Wo, "f16"), bo, C_=C, Nn=Nn, want=("dW", "db"))
        if pending is not None:
            apply(pending)
        pending = r
    tr.flush(); apply(pending)
    Wg, bg, hWg, _ = eng.params_get()
    print("PIPELINED W=%.3e hW=%.3e" % (rel_fro(Wg, Wo), rel_fro(hWg, hW)))
    assert rel_fro(Wg, Wo) <= 1e-3 and rel_fro(hWg, hW) <= 4e-3 and rel_fro(bg, bo) <= 2e-3


def test_embed_matches_oracle(vv, oracle):
    ds, table, idx, W, b = make_case(12, 20, 4, 3, 2, 512, 96)
    eng = vv.Engine(0, "f16")
    eng.table_set(table); eng.params_set(W, b)
    rows = np.array([5, 1, 200, 17, 17], np.int32)
    for l2 in (False, True):
        e = eng.embed(rows, relu=True, l2norm=l2)
        r = oracle.embed(table, rows, W, b, relu=True, l2norm=l2)
        assert rel_rows(e, r) <= 1e-3


def test_error_reporting(vv):
    eng = vv.Engine(0, "f16")
    with pytest.raises(vv.VVError):           # parameters before the table
        eng.F = 4
        eng.params_set(np.zeros((2, 4), np.float32))
    eng.table_synth(1, 10, 8)
    eng.params_set(np.zeros((4, 8), np.float32))
    with pytest.raises(vv.VVError):           # index out of range
        eng.forward_backward(vv.StepConfig(2, 3, 1), np.full((2, 4), 10, np.int32))
    with pytest.raises(vv.VVError):           # context_size < 2 (reference CHECK_GE, ...data_layer.cpp:207)
        eng.forward_backward(vv.StepConfig(2, 1, 1), np.zeros((2, 2), np.int32))
