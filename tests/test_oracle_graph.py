"""The oracle's layer-by-layer ForwardBackward (restating Net::ForwardBackward over
projects/videovec_embedding/mednet_embedding_train.prototxt) against an independent closed-form
float64 formulation (tests/pyref.fused_step) and against finite differences of
the loss -- the reference's own way of validating layers (test_gradient_check_util.hpp)."""
import numpy as np
import pytest

from tests.pyref import fused_step
from videovector_amd.synth import SyntheticVideos, init_weights


def _case(B=8, C=5, Nn=3, F=24, D=10, seed=0, wstd=0.05):
    rng = np.random.default_rng(seed)
    ds = SyntheticVideos(seed=5, n_videos=12)
    table = ds.table(F)
    idx = rng.integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
    W, b = init_weights(seed, D, F, std=wstd)
    b = (rng.standard_normal(D) * 0.01).astype(np.float32)
    return table, idx, W, b


@pytest.mark.parametrize("norm", [1, 2])
@pytest.mark.parametrize("drop", [0.0, 0.5])
def test_forward_backward_matches_closed_form(oracle, norm, drop):
    B, C, Nn, F, D = 8, 5, 3, 24, 10
    table, idx, W, b = _case(B, C, Nn, F, D)
    mask = None
    if drop > 0:
        mask = (np.random.default_rng(9).random(((C + Nn) * B, D)) > drop).astype(np.uint8)
    want = ("Y", "H", "ctx", "posneg", "s_true", "s_bogus", "dY", "dW", "db")
    o = oracle.forward_backward(table, idx, W, b, C_=C, Nn=Nn, margin=2.0, norm=norm,
                                dropout_ratio=drop, dropout_mask=mask, want=want)
    r = fused_step(table, idx, W, b, C, Nn, 2.0, norm, mask=mask, drop=drop)
    assert abs(o["loss"] - r["loss"]) <= 1e-5 * max(1, abs(r["loss"]))
    assert o["violations"] == r["violations"]
    for k in want:
        scale = max(1e-30, np.abs(r[k]).max())
        assert np.abs(o[k] - r[k]).max() <= 2e-5 * scale, k


def test_loss_weight_global_count_and_coeffs(oracle):
    B, C, Nn, F, D = 6, 3, 4, 16, 8
    table, idx, W, b = _case(B, C, Nn, F, D, seed=3)
    coeff = np.array([0.7, 0.1], np.float32)
    o = oracle.forward_backward(table, idx, W, b, C_=C, Nn=Nn, loss_weight=0.5, ctx_coeff=coeff,
                                global_count=4 * B * Nn, want=("dW", "db"))
    r = fused_step(table, idx, W, b, C, Nn, loss_weight=0.5, coeff=coeff, global_count=4 * B * Nn)
    assert abs(o["loss"] - r["loss"]) <= 1e-5
    assert np.abs(o["dW"] - r["dW"]).max() <= 2e-5 * np.abs(r["dW"]).max()
    assert np.abs(o["db"] - r["db"]).max() <= 2e-5 * np.abs(r["db"]).max()


def test_gradient_wrt_weights_by_finite_differences(oracle):
    # whole-graph check in the style of GradientChecker (central differences on the scalar loss)
    B, C, Nn, F, D = 4, 3, 2, 6, 4
    table, idx, W, b = _case(B, C, Nn, F, D, seed=2, wstd=0.3)
    r = fused_step(table, idx, W, b, C, Nn)
    o = oracle.forward_backward(table, idx, W, b, C_=C, Nn=Nn, want=("dW", "db"))
    W64 = W.astype(np.float64)
    step = 1e-5
    for i in range(D):
        for j in range(F):
            Wp, Wm = W64.copy(), W64.copy()
            Wp[i, j] += step
            Wm[i, j] -= step
            est = (fused_step(table, idx, Wp, b, C, Nn)["loss"] -
                   fused_step(table, idx, Wm, b, C, Nn)["loss"]) / (2 * step)
            assert abs(est - r["dW"][i, j]) <= 1e-6 + 1e-4 * abs(est)
            assert abs(est - o["dW"][i, j]) <= 1e-5 + 1e-3 * abs(est)


def test_q1_last_feature_override(oracle):
    B, C, Nn, F, D = 4, 3, 2, 8, 4
    table, idx, W, b = _case(B, C, Nn, F, D, seed=4)
    last = idx.copy()
    last[:, C] = -1                              # slot never fully written: last feature is 0
    last[1, C + 1] = idx[0, 0]
    o = oracle.forward_backward(table, idx, W, b, C_=C, Nn=Nn, last_src=last, want=("Y",))
    t2 = np.concatenate([table, table[idx[:, C]], table[idx[1:2, C + 1]]])
    n0 = table.shape[0]
    t2[n0:n0 + B, F - 1] = 0
    t2[n0 + B, F - 1] = table[idx[0, 0], F - 1]
    idx2 = idx.copy()
    idx2[:, C] = n0 + np.arange(B)
    idx2[1, C + 1] = n0 + B
    o2 = oracle.forward_backward(t2, idx2, W, b, C_=C, Nn=Nn, want=("Y",))
    assert np.array_equal(o["Y"], o2["Y"]) and o["loss"] == o2["loss"]


def test_embed(oracle):
    table, idx, W, b = _case(F=16, D=8)
    rows = np.array([3, 0, 7], np.int32)
    e = oracle.embed(table, rows, W, b, relu=True, l2norm=True)
    y = np.maximum(table[rows].astype(np.float64) @ W.T.astype(np.float64) + b, 0)
    y = y / (np.linalg.norm(y, axis=1, keepdims=True) + 1e-10)
    assert np.abs(e - y).max() <= 1e-6
