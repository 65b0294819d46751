"""Randomised differential test on the GPU: odd shapes (D not a multiple of 4, F not a multiple of 8, single negative,
two-frame context, tiny and ragged batches, empty slots, heavy repetition), de-duplicated against dense execution of the
same library (bit-identical forward, gradients equal up to the reassociated sum) and both against the oracle."""
import numpy as np
import pytest

from tests.test_gpu_parity import TOL, check, rel_fro, run_both, vv  # noqa: F401
from videovector_amd.synth import init_weights

pytestmark = pytest.mark.gpu


def random_case(rng):
    B = int(rng.choice([1, 2, 7, 33, 64, 130]))
    C = int(rng.choice([2, 3, 5, 7]))
    Nn = int(rng.choice([1, 2, 5, 17, 60]))
    F = int(rng.choice([7, 64, 100, 256, 515, 1024]))
    D = int(rng.choice([1, 3, 30, 64, 250, 512, 520]))
    n_rows = int(rng.choice([3, 40, 500, 5000]))
    table = (rng.integers(0, 32, (n_rows, F)) / 8).astype(np.float32) * (rng.random((n_rows, F)) < 0.6)
    idx = rng.integers(0, n_rows, (B, C + Nn)).astype(np.int32)
    if rng.random() < 0.4:
        idx[rng.random(idx.shape) < 0.05] = -1
    W, _ = init_weights(int(rng.integers(1 << 30)), D, F, std=0.05)
    b = (rng.standard_normal(D) * 0.05).astype(np.float32)
    return B, C, Nn, F, D, table.astype(np.float32), idx, W, b


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_dedup_dense_oracle(vv, oracle, seed, fp32_ip2):
    rng = np.random.default_rng(1000 + seed)
    B, C, Nn, F, D, table, idx, W, b = random_case(rng)
    kw = dict(norm="L1" if seed % 5 == 0 else "L2", margin=float(rng.choice([0.5, 2.0])))
    outs = {}
    for mode in (False, True):
        eng = vv.Engine(0, "f16")
        eng.set_dedup(mode)
        eng.table_set(table); eng.params_set(W, b)
        cfg = vv.StepConfig(B, C, Nn, **kw)
        eng.forward_backward(cfg, idx)
        bl = eng.blobs(cfg, ip1_diff=True)
        dW, db = eng.grads()
        outs[mode] = (eng.loss(), bl, dW.copy(), db.copy(), eng.dedup_stats())
    (l0, b0, dW0, db0, _), (l1, b1, dW1, db1, st) = outs[False], outs[True]
    assert st == (B * (C + Nn), len(np.unique(np.where(idx < 0, -1, idx))))
    if D in (512, 1024):
        # the shape of the segment-wise backward (tests/test_gpu_segbwd.py): another kernel computes the scores (last-bit
        # differences) and the bias gradient is summed per distinct row instead of per item
        assert abs(l0[0] - l1[0]) <= 1e-6 * abs(l0[0]) and l0[1] == l1[1]
        assert np.array_equal(b0["ip2"], b1["ip2"])
        for k in ("target_score", "negative_scores"):
            assert np.abs(b0[k] - b1[k]).max() <= 2.5e-7, (k, B, C, Nn, F, D)
        assert rel_fro(b1["ip1_diff"], b0["ip1_diff"]) <= 1e-6 and rel_fro(db1, db0) <= 1e-5
    else:
        assert l0 == l1
        for k in ("ip2", "target_score", "negative_scores", "ip1_diff"):
            assert np.array_equal(b0[k], b1[k]), (k, B, C, Nn, F, D)
        assert np.array_equal(db0, db1)
    scale = max(np.abs(dW0).max(), 1e-30)
    assert np.abs(dW1 - dW0).max() <= 2e-3 * scale, (B, C, Nn, F, D)
    if seed % 3 == 0:
        eng, cfg, got, ref = run_both(vv, oracle, "f16", table, idx, W, b, C, Nn, **kw)
        tol = dict(TOL["f16"], grad_q=5e-3)
        # degenerate shapes (a single embedding dimension, all-zero rows) make relative row errors meaningless
        if D >= 30 and np.isfinite(ref["loss"]) and ref["loss"] > 0:
            check(got, ref, tol, "fuzz-%d" % seed)


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_dropout_on_the_deduplicated_path(vv, oracle, seed, fp32_ip2):
    """Dropout at D = 512 on random shapes (ragged batches, empty slots, heavy repetition, 1..60 negatives, 2..7 channels in front of them):
    the de-duplicated execution (mask per instance on the shared projection) against the dense one (mask in the GEMM's epilogue) with the
    product's counter-hash masks, and -- every third seed -- against the oracle with an explicit mask.  Shapes the register-resident score
    kernel does not take (more than 55 negatives) must fall back to the dense execution by themselves."""
    rng = np.random.default_rng(5000 + seed)
    B, C, Nn, F, _, table, idx, W0, b0 = random_case(rng)
    D = 512
    W, _ = init_weights(int(rng.integers(1 << 30)), D, F, std=0.05)
    b = (rng.standard_normal(D) * 0.05).astype(np.float32)
    ratio = float(rng.choice([0.3, 0.5, 0.9]))
    kw = dict(norm="L1" if seed % 4 == 0 else "L2", margin=float(rng.choice([0.5, 2.0])), dropout_ratio=ratio)
    rides = (C - 1 <= 6) and (1 + Nn <= 56)
    outs = {}
    for dd in (0, 1):
        eng = vv.Engine(0, "f16")
        eng.set_option("drop_dedup", dd)
        eng.table_set(table); eng.params_set(W, b)
        cfg = vv.StepConfig(B, C, Nn, dropout_seed=99 + seed, **kw)
        eng.forward_backward(cfg, idx)
        bl = eng.blobs(cfg, ip1_diff=True)
        dW, db = eng.grads()
        outs[dd] = (eng.loss(), bl, dW.copy(), db.copy(), eng.dedup_stats())
    (l0, b0_, dW0, db0, st0), (l1, b1_, dW1, db1, st1) = outs[0], outs[1]
    n_uniq = len(np.unique(np.where(idx < 0, -1, idx)))
    assert st0 == (B * (C + Nn), B * (C + Nn))
    assert st1 == (B * (C + Nn), n_uniq if rides else B * (C + Nn)), (st1, rides)
    assert np.array_equal(b0_["ip2"] != 0, b1_["ip2"] != 0)                       # the same elements dropped
    assert np.allclose(b0_["ip2"], b1_["ip2"], rtol=1e-5, atol=1e-6)
    if np.isfinite(l0[0]) and l0[0] > 0:
        assert abs(l0[0] - l1[0]) <= 2e-6 * abs(l0[0]) and l0[1] == l1[1]
    for k in ("target_score", "negative_scores"):
        assert np.abs(b0_[k] - b1_[k]).max() <= 1e-6, (k, B, C, Nn, F)
    scale = max(np.abs(dW0).max(), 1e-30)
    assert np.abs(dW1 - dW0).max() <= 2e-3 * scale, (B, C, Nn, F)
    assert rel_fro(b1_["ip1_diff"], b0_["ip1_diff"]) <= 2e-3 or np.abs(b0_["ip1_diff"]).max() == 0
    if seed % 3 == 0:
        mask = (rng.random(((C + Nn) * B, D)) > ratio).astype(np.uint8)
        kw2 = dict(kw, dropout_mask=mask)
        eng, cfg, got, ref = run_both(vv, oracle, "f16", table, idx, W, b, C, Nn, **kw2)
        assert eng.dedup_stats()[1] == (n_uniq if rides else B * (C + Nn))
        if np.isfinite(ref["loss"]) and ref["loss"] > 0:
            check(got, ref, dict(TOL["f16"], grad_q=5e-3), "fuzz-dropout-%d" % seed)
