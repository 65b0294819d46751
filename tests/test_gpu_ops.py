"""The per-layer device operators (vv_op_*, include/videovec.h) -- what the reference's Forward_gpu / Backward_gpu compute
for the hot-path layer classes -- checked the way the reference checks its layers:

  * forward values against the oracle / plain numpy (the assertions of test_neuron_layer.cpp, test_eltwise_layer.cpp,
    test_sum_layer.cpp, test_normalization_layer.cpp, test_max_margin_loss_layer.cpp, test_slice/concat/split_layer.cpp,
    test_inner_product_layer.cpp);
  * backward values with the reference's GradientChecker recipe (test_gradient_check_util.hpp:18-254): central
    differences of the FORWARD operator with step 1e-2, analytic and numeric derivative within threshold 1e-3 relative to
    max(|a|, |n|, 1) -- here the finite differences are taken in float64 on the host forward (numpy restatement already
    checked against the device forward), so that the device backward is what is being tested.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STEP, THR = 1e-2, 1e-3


@pytest.fixture(scope="module")
def eng():
    import videovector_amd as vv
    from videovector_amd.synth import SyntheticVideos, init_weights
    e = vv.Engine(0, "f16")
    ds = SyntheticVideos(seed=3, n_videos=20)
    e.table_synth(ds.seed, ds.n_rows, 64)
    W, b = init_weights(3, 24, 64, std=0.05)
    b = (np.random.default_rng(1).standard_normal(24) * 0.05).astype(np.float32)
    e.params_set(W, b)
    e._case = (ds, W, b)
    return e


def numeric_grad(f, x, top_diff):
    """d(sum(f(x) * top_diff)) / dx by central differences (GradientChecker::CheckGradientSingle with the objective
    sum of top * top_diff)."""
    x = x.astype(np.float64)
    g = np.zeros_like(x)
    flat, gf = x.reshape(-1), g.reshape(-1)
    for i in range(flat.size):
        old = flat[i]
        flat[i] = old + STEP; p = (f(x) * top_diff).sum()
        flat[i] = old - STEP; m = (f(x) * top_diff).sum()
        flat[i] = old
        gf[i] = (p - m) / (2 * STEP)
    return g


def assert_grad(analytic, numeric, kink=None):
    scale = np.maximum(np.maximum(np.abs(analytic), np.abs(numeric)), 1.0)
    bad = np.abs(analytic - numeric) > THR * scale
    if kink is not None:
        bad &= ~kink                      # elements within `step` of a kink are skipped, as the reference does
    assert not bad.any(), (analytic[bad][:5], numeric[bad][:5])


def test_relu_and_leaky_relu(eng, oracle):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((6, 50)).astype(np.float32)
    for slope in (0.0, 0.01):
        X, Y = eng.dev(x), eng.dev(x.shape)
        eng.op("relu", x.size, X, Y, slope)
        y = Y.get()
        assert np.array_equal(y, np.where(x > 0, x, x * np.float32(slope)))        # test_neuron_layer.cpp: TestReLU / WithNegativeSlope
        assert np.array_equal(y, oracle.relu_fwd(x, slope))
        dy = rng.standard_normal(x.shape).astype(np.float32)
        DY, DX = eng.dev(dy), eng.dev(x.shape)
        eng.op("relu_bwd", x.size, X, DY, DX, slope)
        assert np.array_equal(DX.get(), oracle.relu_bwd(x, dy, slope))
        num = numeric_grad(lambda v: np.where(v > 0, v, v * slope), x, dy)
        assert_grad(DX.get().astype(np.float64), num, kink=np.abs(x) < STEP)


def test_dropout_train_statistics_and_backward(eng, oracle):
    n, ratio = 200000, 0.6
    x = np.ones(n, np.float32)
    X, Y, M = eng.dev(x), eng.dev((n,)), eng.dev(np.zeros(n, np.uint8))
    eng.op("dropout", n, X, Y, M, ratio, 12345, 1)
    y, m = Y.get(), M.get()
    scale = 1.0 / (1.0 - ratio)
    assert np.all((y == 0) | (np.abs(y - scale) < 1e-6))                          # test_neuron_layer.cpp: TestDropoutForward
    kept = (y != 0).mean()
    assert abs(kept - (1 - ratio)) <= 1.96 * np.sqrt(ratio * (1 - ratio) / n) * 2  # the reference's 1.96 sigma check, doubled
    assert np.array_equal(m != 0, y != 0)
    assert np.array_equal(y, oracle.dropout_fwd(x, m != 0, ratio))                # same mask -> the oracle's values
    dy = np.random.default_rng(1).standard_normal(n).astype(np.float32)
    DY, DX = eng.dev(dy), eng.dev((n,))
    eng.op("dropout", n, DY, DX, M, ratio, 0, 0)                                  # backward = the same mask on the diff
    assert np.allclose(DX.get(), dy * m * np.float32(scale))
    assert np.allclose(DX.get(), oracle.dropout_bwd(dy, m != 0, ratio), rtol=1e-6, atol=0)


def test_eltwise_sum_coeff_and_prod(eng, oracle):
    rng = np.random.default_rng(2)
    a, b, c3 = [rng.standard_normal((4, 30)).astype(np.float32) for _ in range(3)]
    A, B, C3, Y = eng.dev(a), eng.dev(b), eng.dev(c3), eng.dev(a.shape)
    # SUM with coefficients 1, -0.5, 2 (test_eltwise_layer.cpp: TestSumCoeff)
    eng.op("axpby", a.size, 1.0, A, 0.0, Y)
    eng.op("axpby", a.size, -0.5, B, 1.0, Y)
    eng.op("axpby", a.size, 2.0, C3, 1.0, Y)
    assert np.allclose(Y.get(), a - 0.5 * b + 2 * c3, rtol=1e-6, atol=1e-6)
    assert np.allclose(Y.get(), oracle.eltwise_fwd("SUM", [a, b, c3], [1, -0.5, 2]), rtol=1e-6, atol=1e-7)  # fma contraction
    # PROD of two bottoms (TestProd) and its stable backward: dA = B * dY
    eng.op("mul", a.size, A, B, Y, 0)
    assert np.array_equal(Y.get(), oracle.eltwise_fwd("PROD", [a, b]))
    dy = rng.standard_normal(a.shape).astype(np.float32)
    DY, DA = eng.dev(dy), eng.dev(a.shape)
    eng.op("mul", a.size, B, DY, DA, 0)
    assert np.array_equal(DA.get(), oracle.eltwise_bwd("PROD", [a, b], dy, 0))
    assert_grad(DA.get().astype(np.float64), numeric_grad(lambda v: v * b, a, dy))


def test_slice_concat_split_copies(eng, oracle):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5, 7, 6)).astype(np.float32)                          # (num, channels, dim)
    X = eng.dev(x)
    # SLICE dim 1 into channels [0,3) and [3,7) (test_slice_layer.cpp: TestSliceAcrossChannels): per item a strided block
    T0, T1 = eng.dev((5, 3, 6)), eng.dev((5, 4, 6))
    eng.op("copy2d", X, 7 * 6, T0, 3 * 6, 5, 3 * 6, 0)
    import ctypes
    off = ctypes.c_void_p(X.ptr.value + 3 * 6 * 4)
    eng.op("copy2d", off, 7 * 6, T1, 4 * 6, 5, 4 * 6, 0)
    assert np.array_equal(T0.get(), x[:, :3]) and np.array_equal(T1.get(), x[:, 3:])
    o0, o1 = oracle.slice_fwd(x, 1, [3, 4])
    assert np.array_equal(T0.get(), o0[..., 0]) and np.array_equal(T1.get(), o1[..., 0])
    # CONCAT dim 1 puts them back (test_concat_layer.cpp); SLICE backward is the same copy
    Z = eng.dev(x.shape)
    eng.op("copy2d", T0, 3 * 6, Z, 7 * 6, 5, 3 * 6, 0)
    eng.op("copy2d", T1, 4 * 6, ctypes.c_void_p(Z.ptr.value + 3 * 6 * 4), 7 * 6, 5, 4 * 6, 0)
    assert np.array_equal(Z.get(), x)
    assert np.array_equal(Z.get(), oracle.concat_fwd([o0, o1], 1)[..., 0])
    # SPLIT backward (split_layer.cu:18-33): bottom diff = sum of the top diffs
    d0, d1, d2 = [rng.standard_normal(x.shape).astype(np.float32) for _ in range(3)]
    D0, D1, D2, S = eng.dev(d0), eng.dev(d1), eng.dev(d2), eng.dev(x.shape)
    eng.op("copy2d", D0, x.size, S, x.size, 1, x.size, 0)
    eng.op("copy2d", D1, x.size, S, x.size, 1, x.size, 1)
    eng.op("copy2d", D2, x.size, S, x.size, 1, x.size, 1)
    assert np.array_equal(S.get(), oracle.split_bwd([d0, d1, d2]))


@pytest.mark.parametrize("num_output", [1, 10])
def test_sum_layer(eng, oracle, num_output):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((6, 20)).astype(np.float32)
    X, Y = eng.dev(x), eng.dev((6, num_output))
    eng.op("rowsum", 6, 20, X, num_output, Y)
    assert np.allclose(Y.get(), oracle.sum_fwd(x, num_output), rtol=1e-5, atol=1e-5)   # test_sum_layer.cpp:39-78
    dy = rng.standard_normal((6, num_output)).astype(np.float32)
    DY, DX = eng.dev(dy), eng.dev(x.shape)
    eng.op("rowsum_bwd", 6, 20, num_output, DY, DX)
    assert np.allclose(DX.get(), oracle.sum_bwd(dy, 20), rtol=1e-5, atol=1e-5)
    num = numeric_grad(lambda v: np.repeat(v.sum(1, keepdims=True), num_output, 1), x, dy)
    assert_grad(DX.get().astype(np.float64), num)


def test_normalization_layer(eng, oracle):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((7, 33)).astype(np.float32)
    x[3] = 0                                                                       # all-zero row: output 0, gradient 0 (quirk Q6)
    X, Y = eng.dev(x), eng.dev(x.shape)
    eng.op("normalize", 7, 33, X, Y)
    y = Y.get()
    assert np.allclose(y, oracle.normalize_fwd(x), rtol=1e-5, atol=1e-6)
    n = np.linalg.norm(y, axis=1)
    assert np.allclose(np.delete(n, 3), 1.0, atol=1e-5) and n[3] == 0                # test_normalization_layer.cpp:39-60
    dy = rng.standard_normal(x.shape).astype(np.float32)
    DY, DX = eng.dev(dy), eng.dev(x.shape)
    eng.op("normalize_bwd", 7, 33, X, DY, DX)
    dx = DX.get()
    assert np.allclose(dx, oracle.normalize_bwd(x, dy), rtol=1e-4, atol=1e-6) and np.all(dx[3] == 0)
    f = lambda v: v / (np.sqrt((v * v).sum(1, keepdims=True)) + 1e-10)
    num = numeric_grad(f, np.delete(x, 3, 0), np.delete(dy, 3, 0))
    assert_grad(np.delete(dx, 3, 0).astype(np.float64), num)


@pytest.mark.parametrize("norm,weighted", [(2, False), (1, False), (2, True), (1, True)])
def test_max_margin_loss_layer(eng, oracle, norm, weighted):
    rng = np.random.default_rng(6)
    cnt, margin, lw = 60, 2.0, 1.5
    st = rng.standard_normal(cnt).astype(np.float32)
    sb = rng.standard_normal(cnt).astype(np.float32)
    w = (rng.random(cnt).astype(np.float32) * 2) if weighted else None
    ST, SB = eng.dev(st), eng.dev(sb)
    Wd = eng.dev(w) if weighted else None
    import ctypes
    l, v = ctypes.c_float(), ctypes.c_float()
    eng.op("max_margin", cnt, ST, SB, Wd, margin, norm, ctypes.byref(l), ctypes.byref(v))
    rl, rv = oracle.max_margin_fwd(st, sb, margin, norm, weight=w)
    assert abs(l.value - rl) <= 1e-5 * max(1, abs(rl)) and v.value == rv          # test_max_margin_loss_layer.cpp:53-77 (brute force)
    DT, DB = eng.dev((cnt,)), eng.dev((cnt,))
    eng.op("max_margin_bwd", cnt, ST, SB, Wd, margin, norm, lw, DT, DB)
    rt, rb = oracle.max_margin_bwd(st, sb, margin, norm, loss_weight=lw, weight=w)
    assert np.allclose(DT.get(), rt, rtol=1e-5, atol=1e-7) and np.allclose(DB.get(), rb, rtol=1e-5, atol=1e-7)
    if norm == 2:                                                                 # smooth enough for the finite-difference check
        def loss(b_):
            h = np.maximum(0, margin - (st.astype(np.float64) - b_))
            if w is not None:
                h = h * np.sqrt(w)
            return np.array([lw * (h * h).sum() / cnt])
        num = numeric_grad(loss, sb, np.ones(1))
        assert_grad(DB.get().astype(np.float64), num, kink=np.abs(margin - (st - sb)) < 2 * STEP)


def test_data_layer_gather_and_inner_product(eng, oracle):
    ds, W, b = eng._case
    rng = np.random.default_rng(7)
    idx = rng.integers(0, ds.n_rows, size=300).astype(np.int32)
    idx[5] = -1
    X = eng.dev((300, 64))
    eng.op("gather_rows", idx, 300, X)
    x = X.get()
    ref = ds.table(64)[np.maximum(idx, 0)]
    ref[5] = 0
    assert np.array_equal(x, ref)                                                   # synthetic features are exact in f16
    Y = eng.dev((300, 24))
    eng.op("inner_product", X, 300, Y)
    y = Y.get()
    yr = x.astype(np.float64) @ W.astype(np.float64).T + b                          # test_inner_product_layer.cpp:43-75
    assert np.abs(y - yr).max() <= 1e-3 * np.abs(yr).max()
    dy = (rng.standard_normal((300, 24)) * 1e-3).astype(np.float32)
    DY = eng.dev(dy)
    eng.op("inner_product_bwd", DY, 300, 0.0)
    dW, db = eng.grads()
    dWr, dbr = dy.astype(np.float64).T @ x.astype(np.float64), dy.astype(np.float64).sum(0)
    assert np.linalg.norm(dW - dWr) <= 2e-3 * np.linalg.norm(dWr) and np.abs(db - dbr).max() <= 1e-5 * np.abs(dbr).max() + 1e-8
    eng.op("inner_product_bwd", DY, 300, 0.5)                                       # regularization 0.5: dW * 1.25 (inner_product_layer.cpp:80-90)
    dW2, _ = eng.grads()
    assert np.allclose(dW2, 1.25 * dW, rtol=1e-5, atol=1e-9)
