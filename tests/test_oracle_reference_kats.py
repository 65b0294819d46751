"""The property / known-answer assertions of the reference's own unit tests for the layers on the
hot path, applied to the oracle.  The reference's fixtures are seeded
boost-RNG fills (not reproducible without boost), so each test re-creates the SAME SHAPES AND
DISTRIBUTIONS with numpy and applies the SAME assertion with the SAME tolerances; the finite
difference checker mirrors include/caffe/test/test_gradient_check_util.hpp:74-200 (central
differences, scale floor 1, optional kink band).
"""
import numpy as np
import pytest


def fd_check(f_obj, grad, x, step, thr, kink=0.0, kink_range=-1.0):
    """GradientChecker::CheckGradientSingle for one scalar objective."""
    x = x.copy()
    flat = x.reshape(-1)
    g = grad.reshape(-1)
    for i in range(flat.size):
        feat = float(flat[i])
        flat[i] = feat + step
        pos = f_obj(x)
        flat[i] = feat - step
        neg = f_obj(x)
        flat[i] = feat
        est = (pos - neg) / step / 2.0
        if kink - kink_range > abs(feat) or abs(feat) > kink + kink_range:
            scale = max(abs(g[i]), abs(est), 1.0)
            assert abs(g[i] - est) <= thr * scale, (i, g[i], est)


# ---------------------------------------------------------------- MaxMarginLoss --------------
@pytest.fixture
def mm_bottoms():
    # test_max_margin_loss_layer.cpp:24-33: two (10,5,1,1) blobs, gaussian std 10
    rng = np.random.default_rng(1701)
    return (rng.standard_normal((10, 5)) * 10).astype(np.float32), \
           (rng.standard_normal((10, 5)) * 10).astype(np.float32)


def test_max_margin_forward_l1(oracle, mm_bottoms):
    # test_max_margin_loss_layer.cpp:53-79 (defaults: norm L1, margin 1; caffe.proto:864,867)
    t, b = mm_bottoms
    loss, _ = oracle.max_margin_fwd(t, b, 1.0, 1)
    d = t.astype(np.float64) - b.astype(np.float64)
    expect = np.where(d < 1, 1 - d, 0).sum() / d.size
    assert abs(loss - expect) <= 1e-3


def test_max_margin_violations(oracle, mm_bottoms):
    # max_margin_loss_layer.cpp:78-80,123-126: second top = number of (true - bogus) < 0
    t, b = mm_bottoms
    _, viol = oracle.max_margin_fwd(t, b, 1.0, 1)
    assert viol == float(((t - b) < 0).sum())


def test_max_margin_gradient_l1(oracle, mm_bottoms):
    # test_max_margin_loss_layer.cpp:81-88: GradientChecker(1e-2, 2e-3, 1701, kink 1, range 0.01),
    # bottom 0 only; objective = loss * 2 (GetObjAndGradient loss_weight 2, :242-246)
    t, b = mm_bottoms
    dt, _ = oracle.max_margin_bwd(t, b, 1.0, 1, loss_weight=2.0)
    # the kink of the hinge sits at (t - b) == margin, the reference's band is on |t|; excluding
    # elements within 2 steps of the hinge kink keeps the check meaningful for any fill
    near = np.abs((t - b) - 1.0) < 2e-2
    assert near.sum() == 0
    fd_check(lambda x: 2.0 * oracle.max_margin_fwd(x, b, 1.0, 1)[0], dt, t, 1e-2, 2e-3, 1.0, 0.01)


def test_max_margin_gradient_l2(oracle, mm_bottoms):
    # test_max_margin_loss_layer.cpp:90-100: norm L2, GradientChecker(1e-2, 1e-2, 1701)
    t, b = mm_bottoms
    dt, db = oracle.max_margin_bwd(t, b, 1.0, 2, loss_weight=2.0)
    fd_check(lambda x: 2.0 * oracle.max_margin_fwd(x, b, 1.0, 2)[0], dt, t, 1e-2, 1e-2)
    assert np.array_equal(db, -dt)     # max_margin_loss_layer.cpp:211


def test_max_margin_weighted(oracle, mm_bottoms):
    # max_margin_loss_layer.cpp:82-97 (use_direct_weight): L2 scales the hinge by sqrt(w)
    t, b = mm_bottoms
    w = np.abs(np.random.default_rng(3).standard_normal(t.shape)).astype(np.float32)
    loss, _ = oracle.max_margin_fwd(t, b, 2.0, 2, weight=w)
    h = np.maximum(0, 2.0 - (t.astype(np.float64) - b))
    assert abs(loss - (w * h * h).sum() / h.size) <= 1e-3 * max(1.0, loss)


# ---------------------------------------------------------------- Normalization --------------
def test_normalization_forward_unit_norm(oracle):
    # test_normalization_layer.cpp:39-73: (2,3,4,5) gaussian; every row of top has |.|^2 == 1 +-1e-3
    x = np.random.default_rng(1701).standard_normal((2, 60)).astype(np.float32)
    y = oracle.normalize_fwd(x)
    assert np.all(np.abs((y.astype(np.float64) ** 2).sum(1) - 1) <= 1e-3)


def test_normalization_gradient(oracle):
    # test_normalization_layer.cpp:76-83: exhaustive GradientChecker(1e-2, 1e-3)
    x = np.random.default_rng(1701).standard_normal((2, 60)).astype(np.float32)
    y = oracle.normalize_fwd(x)
    for j in range(0, y.size, 7):          # every 7th top element keeps the CPU suite fast
        dy = np.zeros_like(y)
        dy.reshape(-1)[j] = 2.0
        dx = oracle.normalize_bwd(x, dy)
        fd_check(lambda v: 2.0 * float(oracle.normalize_fwd(v).reshape(-1)[j]), dx, x, 1e-2, 1e-3)


def test_normalization_zero_row(oracle):
    # quirk Q6: eps placement makes an all-zero row produce 0 output and 0 gradient
    x = np.zeros((1, 8), np.float32)
    assert not np.any(oracle.normalize_fwd(x))
    assert not np.any(oracle.normalize_bwd(x, np.ones_like(x)))


# ---------------------------------------------------------------- Sum ------------------------
@pytest.mark.parametrize("num_output", [1, 10])
def test_sum_forward_and_gradient(oracle, num_output):
    # test_sum_layer.cpp:39-117: (10,5,1,1); forward == row sums replicated; gradients
    x = np.random.default_rng(1701).standard_normal((10, 5)).astype(np.float32)
    y = oracle.sum_fwd(x, num_output)
    assert np.allclose(y, np.repeat(x.sum(1, keepdims=True), num_output, 1), atol=1e-4)
    for j in range(y.size):
        dy = np.zeros_like(y)
        dy.reshape(-1)[j] = 2.0
        dx = oracle.sum_bwd(dy, 5)
        fd_check(lambda v: 2.0 * float(oracle.sum_fwd(v, num_output).reshape(-1)[j]), dx, x,
                 1e-2, 1e-3)


# ---------------------------------------------------------------- SGD solver -----------------
@pytest.mark.parametrize("lr,wd,mom,iters", [(1.0, 0.0, 0.0, 1), (0.1, 0.0, 0.0, 1),
                                             (1.0, 0.5, 0.0, 1), (0.01, 0.1, 0.5, 4),
                                             (0.1, 0.0, 0.9, 4)])
def test_sgd_update_matches_least_squares_algebra(oracle, lr, wd, mom, iters):
    # test_gradient_based_solver.cpp:140-250,310-374: after K iterations the (K+1)-th update is
    #   update = lr * (grad + wd * w) + momentum * history ;  w_new = w - update   (1e-2 rel)
    rng = np.random.default_rng(1701)
    X = rng.standard_normal((5, 300)).astype(np.float64)
    yv = rng.standard_normal(5)
    w = rng.standard_normal(300).astype(np.float32)
    hist = np.zeros_like(w)
    w64, h64 = w.astype(np.float64), np.zeros(300)
    for _ in range(iters + 1):
        grad64 = X.T @ (X @ w64 - yv) / 5
        grad = (X.T @ (X @ w.astype(np.float64) - yv) / 5).astype(np.float32)
        oracle.sgd_update(w, grad, hist, lr, 1.0, mom, wd, 1.0)
        upd = lr * (grad64 + wd * w64) + mom * h64
        w64, h64 = w64 - upd, upd
        assert np.allclose(hist, h64, rtol=1e-2, atol=1e-4)
        assert np.allclose(w, w64, rtol=1e-2, atol=1e-4)


@pytest.mark.parametrize("solver,lr,wd,mom,iters", [
    ("NESTEROV", 1.0, 0.0, 0.0, 1), ("NESTEROV", 0.1, 0.0, 0.0, 1), ("NESTEROV", 1.0, 0.5, 0.0, 1),
    ("NESTEROV", 1.0, 0.0, 0.5, 2), ("NESTEROV", 0.1, 0.0, 0.9, 4), ("NESTEROV", 0.01, 0.1, 0.9, 4),
    ("ADAGRAD", 1.0, 0.0, 0.0, 1), ("ADAGRAD", 0.1, 0.0, 0.0, 1), ("ADAGRAD", 1.0, 0.5, 0.0, 1),
    ("ADAGRAD", 0.01, 0.1, 0.0, 4)])
def test_nesterov_adagrad_updates_match_least_squares_algebra(oracle, solver, lr, wd, mom, iters):
    # test_gradient_based_solver.cpp:392-485 (AdaGradSolverTest / NesterovSolverTest cases) with the expected update
    # of ComputeLeastSquaresUpdate (:196-211):
    #   NESTEROV: u = lr*g + m*h ; update = (1+m)*u - m*h      ADAGRAD: update = lr*g / (sqrt(h + g^2) + delta)
    # (g includes the weight decay term; precision 1e-2 relative as CheckLeastSquaresUpdate :227-236)
    delta = 1e-8
    rng = np.random.default_rng(1701)
    X = rng.standard_normal((5, 300)).astype(np.float64)
    yv = rng.standard_normal(5)
    w = rng.standard_normal(300).astype(np.float32)
    hist = np.zeros_like(w)
    w64, h64 = w.astype(np.float64), np.zeros(300)
    for _ in range(iters + 1):
        g64 = X.T @ (X @ w64 - yv) / 5 + wd * w64
        grad = (X.T @ (X @ w.astype(np.float64) - yv) / 5).astype(np.float32)
        oracle.sgd_update(w, grad, hist, lr, 1.0, mom, wd, 1.0, solver=solver, delta=delta)
        if solver == "NESTEROV":
            u = lr * g64 + mom * h64
            upd, h64 = (1 + mom) * u - mom * h64, u
        else:
            h64 = h64 + g64 * g64
            upd = lr * g64 / (np.sqrt(h64) + delta)
        w64 = w64 - upd
        assert np.allclose(hist, h64, rtol=1e-2, atol=1e-4)
        assert np.allclose(w, w64, rtol=1e-2, atol=1e-4)


def test_sgd_lr_and_decay_multipliers_and_l1(oracle):
    # solver.cpp:502-531: local_rate = rate*lr_mult, local_decay = wd*decay_mult; L1 uses sign(w)
    w = np.array([1.0, -2.0, 0.0, 3.0], np.float32)
    g = np.array([0.5, 0.5, 0.5, 0.5], np.float32)
    h = np.array([1.0, 1.0, 1.0, 1.0], np.float32)
    w2, g2, h2 = w.copy(), g.copy(), h.copy()
    oracle.sgd_update(w2, g2, h2, 0.1, 2.0, 0.9, 0.01, 1.0, reg="L1")
    eh = 0.2 * (g + 0.01 * np.sign(w)) + 0.9 * h
    assert np.allclose(h2, eh) and np.allclose(w2, w - eh) and np.allclose(g2, eh)
    w3, g3, h3 = w.copy(), g.copy(), h.copy()
    oracle.sgd_update(w3, g3, h3, 0.1, 2.0, 0.9, 0.01, 0.0)      # decay_mult 0 (the bias blob)
    assert np.allclose(h3, 0.2 * g + 0.9 * h)


def test_learning_rate_policies(oracle):
    # solver.cpp:440-460
    assert oracle.learning_rate("fixed", 0.01, 0.1, 0.75, 10, 77) == np.float32(0.01)
    assert np.isclose(oracle.learning_rate("step", 0.01, 0.1, 0, 10, 25), 0.01 * 0.1 ** 2)
    assert np.isclose(oracle.learning_rate("exp", 0.01, 0.99, 0, 0, 10), 0.01 * 0.99 ** 10)
    # shipped solver: inv, gamma 1e-3, power .75 (mednet_embedding_train_solver.prototxt:12-22)
    assert np.isclose(oracle.learning_rate("inv", 1e-3, 1e-3, 0.75, 0, 2000), 1e-3 * 3.0 ** -0.75)


# ---------------------------------------------------------------- sgemm (BLAS stand-in) ------
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_sgemm_against_numpy(oracle, ta, tb):
    # test_util_blas.cpp TestGemm: caffe_cpu_gemm trans combinations
    rng = np.random.default_rng(5)
    M, N, K = 37, 70, 53
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    out = oracle.sgemm(ta, tb, A, B, alpha=0.5, beta=2.0, Cmat=C0.copy())
    ref = 0.5 * ((A.T if ta else A).astype(np.float64) @ (B.T if tb else B)) + 2.0 * C0
    assert np.allclose(out, ref, rtol=1e-5, atol=1e-4)


# ---------------------------------------------------------------- RetrievalStats -------------
def test_retrieval_stats_known_answer(oracle):
    # test_retrieval_stats_layer.cpp:34-39,71-86: 5 samples x 2 features, video ids 2..6 with classes
    # {1,2,1,2,2} (comment on line 39) -> mAP 0.7833333, hit@1 0.60, hit@5 0.32 (+-1e-3)
    feat = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 0.06], [0.0, 1.0], [1.0, 0.1]], np.float32)
    m, h1, h5 = oracle.retrieval_stats(feat, [2, 3, 4, 5, 6], {2: 1, 3: 2, 4: 1, 5: 2, 6: 2})
    assert abs(m - 0.7833333) <= 1e-3 and abs(h1 - 0.60) <= 1e-3 and abs(h5 - 0.32) <= 1e-3


def test_retrieval_stats_options(oracle):
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((40, 8)).astype(np.float32)
    feat /= np.linalg.norm(feat, axis=1, keepdims=True)
    vids = np.repeat(np.arange(10), 4)
    cls = {v: v % 3 - (v == 9) * 5 for v in range(10)}        # video 9 has a negative class: skipped
    a = oracle.retrieval_stats(feat, vids, cls, True)
    b = oracle.retrieval_stats(feat, vids, cls, False)
    assert all(0 <= x <= 1 for x in a + b) and a != b


@pytest.mark.parametrize("force_avx2", [False, True])
def test_blocked_sgemm_edges_and_both_kernels(oracle, force_avx2):
    """The blocked sgemm of the CPU baseline: every transpose combination, sizes that are not multiples of the register
    block, the cache blocks or the thread count, alpha / beta, on the AVX-512 kernel (when the CPU has it) and the AVX2 one."""
    try:
        oracle.sgemm_isa(force_avx2=force_avx2)
        oracle.set_threads(8)
        rng = np.random.default_rng(5)
        # (100, 700, 900) and (200, 4100, 37): few row blocks and enough work for several threads -- the arrangement
        # with one shared packed slab of A (the weight gradient's shape); (1000, 64, 500): many row blocks
        for (M, N, K) in [(1, 1, 1), (7, 33, 5), (97, 50, 401), (200, 4100, 37), (13, 17, 800), (64, 64, 64),
                          (100, 700, 900), (1000, 64, 500)]:
            for ta in (False, True):
                for tb in (False, True):
                    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
                    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
                    C0 = rng.standard_normal((M, N)).astype(np.float32)
                    ref = 0.75 * ((A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)) - 1.5 * C0
                    out = oracle.sgemm(ta, tb, A, B, alpha=0.75, beta=-1.5, Cmat=C0.copy())
                    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (M, N, K, ta, tb)
                    out0 = oracle.sgemm(ta, tb, A, B)
                    ref0 = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
                    assert np.abs(out0 - ref0).max() <= 2e-5 * max(1.0, np.abs(ref0).max())
    finally:
        oracle.sgemm_isa(force_avx2=False)
        oracle.set_threads(0)


# ---------------------------------------------------------------- the plain layers -----------
def layer_gradient(fwd, bwd, bottoms, step, thr, eltwise=False, kink=0.0, kink_range=-1.0, check=None):
    """GradientChecker::CheckGradientExhaustive / CheckGradientEltwise (test_gradient_check_util.hpp:74-200):
    for every top element j the objective is 2 * top[j] (GetObjAndGradient with loss_weight 2); every bottom
    element is compared with a central difference; with `eltwise` only bottom element j is differenced and every
    other analytic gradient must be exactly zero (:141-147)."""
    bottoms = [b.copy() for b in bottoms]
    tops = fwd(bottoms)
    for t, top in enumerate(tops):
        for j in range(top.size):
            dtops = [np.zeros_like(x) for x in tops]
            dtops[t].reshape(-1)[j] = 2.0
            grads = bwd(bottoms, dtops)
            for bi in (range(len(bottoms)) if check is None else check):
                g = grads[bi].reshape(-1)
                flat = bottoms[bi].reshape(-1)
                for i in range(flat.size):
                    if eltwise and i != j:
                        assert g[i] == 0.0
                        continue
                    feat = float(flat[i])
                    flat[i] = feat + step
                    pos = 2.0 * float(fwd(bottoms)[t].reshape(-1)[j])
                    flat[i] = feat - step
                    neg = 2.0 * float(fwd(bottoms)[t].reshape(-1)[j])
                    flat[i] = feat
                    est = (pos - neg) / step / 2.0
                    if kink - kink_range > abs(feat) or abs(feat) > kink + kink_range:
                        scale = max(abs(g[i]), abs(est), 1.0)
                        assert abs(g[i] - est) <= thr * scale, (t, j, bi, i, g[i], est)


@pytest.fixture
def neuron_bottom():
    # test_neuron_layer.cpp:20-30: (2,3,4,5), gaussian(0,1)
    return np.random.default_rng(1701).standard_normal((2, 3, 4, 5)).astype(np.float32)


@pytest.mark.parametrize("slope", [0.0, 0.01])
def test_relu_forward(oracle, neuron_bottom, slope):
    # test_neuron_layer.cpp:96-109.  The reference's "negative slope" variant (:120-134) hands a TEXT-format string
    # to ParseFromString (binary wire format), so its layer still runs with slope 0 and asserts the same thing; the
    # slope 0.01 case here asserts the formula of relu_layer.cpp:16-19 itself.
    y = oracle.relu_fwd(neuron_bottom, slope)
    x = neuron_bottom
    if slope == 0.0:
        assert np.all(y >= 0)
        assert np.all((y == 0) | (y == x))
    assert np.array_equal(y, np.maximum(x, 0) + np.float32(slope) * np.minimum(x, 0))


@pytest.mark.parametrize("slope", [0.0, 0.01])
def test_relu_gradient(oracle, neuron_bottom, slope):
    # test_neuron_layer.cpp:111-118, 136-144: GradientChecker(1e-2, 1e-3, 1701, kink 0, range 0.01), eltwise
    layer_gradient(lambda b: [oracle.relu_fwd(b[0], slope)],
                   lambda b, d: [oracle.relu_bwd(b[0], d[0], slope)],
                   [neuron_bottom], 1e-2, 1e-3, eltwise=True, kink=0.0, kink_range=0.01)


@pytest.mark.parametrize("ratio", [0.5, 0.75])
def test_dropout_forward_train(oracle, neuron_bottom, ratio):
    # test_neuron_layer.cpp:38-68 (TestDropoutHalf / ThreeQuarters): kept entries == bottom * 1/(1-ratio), the
    # rest are zero.  The Bernoulli draw itself is an input of the oracle (the product draws its own on the GPU;
    # tests/test_gpu_ops.py applies the reference's 1.96 sigma test to that).
    x = neuron_bottom
    mask = (np.random.default_rng(7).random(x.shape) >= ratio).astype(np.uint8)
    y = oracle.dropout_fwd(x, mask, ratio, train=True)
    scale = np.float32(1.0 / (1.0 - ratio))
    kept = y != 0
    assert np.array_equal(y[kept], (x * scale)[kept])
    assert np.array_equal(kept, (mask != 0) & (x != 0))


def test_dropout_forward_test_phase(oracle, neuron_bottom):
    # test_neuron_layer.cpp:214-229
    mask = np.zeros(neuron_bottom.shape, np.uint8)        # ignored in the TEST phase
    assert np.array_equal(oracle.dropout_fwd(neuron_bottom, mask, 0.5, train=False), neuron_bottom)


@pytest.mark.parametrize("train", [True, False])
def test_dropout_gradient(oracle, neuron_bottom, train):
    # test_neuron_layer.cpp:231-249 (the checker re-seeds before every forward, so the mask is fixed)
    mask = (np.random.default_rng(7).random(neuron_bottom.shape) >= 0.5).astype(np.uint8)
    layer_gradient(lambda b: [oracle.dropout_fwd(b[0], mask, 0.5, train)],
                   lambda b, d: [oracle.dropout_bwd(d[0], mask, 0.5, train)],
                   [neuron_bottom], 1e-2, 1e-3, eltwise=True)


@pytest.fixture
def eltwise_bottoms():
    # test_eltwise_layer.cpp:21-38: three (2,3,4,5) blobs, uniform [0,1)
    rng = np.random.default_rng(1701)
    return [rng.random((2, 3, 4, 5)).astype(np.float32) for _ in range(3)]


def test_eltwise_prod(oracle, eltwise_bottoms):
    a, b, c = eltwise_bottoms                     # test_eltwise_layer.cpp:68-85
    assert np.array_equal(oracle.eltwise_fwd("PROD", [a, b, c]), a * b * c)


def test_eltwise_sum(oracle, eltwise_bottoms):
    a, b, c = eltwise_bottoms                     # test_eltwise_layer.cpp:87-104
    assert np.array_equal(oracle.eltwise_fwd("SUM", [a, b, c]), a + b + c)


def test_eltwise_sum_coeff(oracle, eltwise_bottoms):
    a, b, c = eltwise_bottoms                     # test_eltwise_layer.cpp:106-127
    y = oracle.eltwise_fwd("SUM", [a, b, c], coeff=[1, -0.5, 2])
    assert np.all(np.abs(y - (a - 0.5 * b + 2 * c)) <= 1e-4)


def test_eltwise_max(oracle, eltwise_bottoms):
    a, b, c = eltwise_bottoms                     # test_eltwise_layer.cpp:181-199
    assert np.array_equal(oracle.eltwise_fwd("MAX", [a, b, c]), np.maximum(a, np.maximum(b, c)))


@pytest.mark.parametrize("op,coeff,stable", [("PROD", None, True), ("PROD", None, False), ("SUM", None, True),
                                             ("SUM", [1, -0.5, 2], True), ("MAX", None, True)])
def test_eltwise_gradients(oracle, eltwise_bottoms, op, coeff, stable):
    # test_eltwise_layer.cpp:129-179: GradientChecker(1e-2, 1e-3), CheckGradientEltwise; MAX (:196-207) steps 1e-4
    n = len(eltwise_bottoms)
    step = 1e-4 if op == "MAX" else 1e-2
    layer_gradient(lambda b: [oracle.eltwise_fwd(op, b, coeff)],
                   lambda b, d: [oracle.eltwise_bwd(op, b, d[0], k, coeff, stable) for k in range(n)],
                   eltwise_bottoms, step, 1e-3, eltwise=True)


@pytest.fixture
def slice_bottom():
    # test_slice_layer.cpp:21-37: (6,12,2,3) gaussian
    return np.random.default_rng(1701).standard_normal((6, 12, 2, 3)).astype(np.float32)


def test_slice_across_num(oracle, slice_bottom):
    # test_slice_layer.cpp:62-74, 91-119: two tops of num 3
    t0, t1 = oracle.slice_fwd(slice_bottom, 0, [3, 3])
    assert t0.shape == (3, 12, 2, 3) and t1.shape == (3, 12, 2, 3)
    assert np.array_equal(t0, slice_bottom[:3]) and np.array_equal(t1, slice_bottom[3:])


def test_slice_across_channels(oracle, slice_bottom):
    # test_slice_layer.cpp:121-162: slice points 2, 8 -> channels 2, 6, 4
    t0, t1, t2 = oracle.slice_fwd(slice_bottom, 1, [2, 6, 4])
    assert (t0.shape[1], t1.shape[1], t2.shape[1]) == (2, 6, 4)
    assert np.array_equal(t0, slice_bottom[:, :2]) and np.array_equal(t1, slice_bottom[:, 2:8])
    assert np.array_equal(t2, slice_bottom[:, 8:])


@pytest.mark.parametrize("dim,widths", [(0, [2, 2]), (1, [4, 1])])
def test_slice_gradient(oracle, dim, widths):
    # test_slice_layer.cpp:164-187: bottom reduced to (4,5,2,2); GradientChecker(1e-2, 1e-3), exhaustive
    x = np.random.default_rng(1701).standard_normal((4, 5, 2, 2)).astype(np.float32)
    layer_gradient(lambda b: oracle.slice_fwd(b[0], dim, widths),
                   lambda b, d: [oracle.concat_fwd(d, dim)], [x], 1e-2, 1e-3)


def test_concat_forward_and_shapes(oracle):
    # test_concat_layer.cpp:21-43 (constant fills 1, 2, 3), :61-85 shapes, :87-111 values
    b0 = np.full((2, 3, 6, 5), 1, np.float32)
    b1 = np.full((2, 5, 6, 5), 2, np.float32)
    b2 = np.full((5, 3, 6, 5), 3, np.float32)
    top = oracle.concat_fwd([b0, b2], 0)
    assert top.shape == (7, 3, 6, 5)
    assert np.array_equal(top[:2], b0) and np.array_equal(top[2:], b2)
    top = oracle.concat_fwd([b0, b1], 1)
    assert top.shape == (2, 8, 6, 5)
    assert np.array_equal(top[:, :3], b0) and np.array_equal(top[:, 3:], b1)


@pytest.mark.parametrize("dim", [0, 1])
def test_concat_gradient(oracle, dim):
    # test_concat_layer.cpp:113-120: GradientChecker(1e-2, 1e-2); the blobs are cut down to keep the run short
    rng = np.random.default_rng(1701)
    shapes = [(2, 3, 2, 2), (2, 5, 2, 2)] if dim == 1 else [(2, 3, 2, 2), (3, 3, 2, 2)]
    bs = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    layer_gradient(lambda b: [oracle.concat_fwd(b, dim)],
                   lambda b, d: oracle.slice_fwd(d[0], dim, [s[dim] for s in shapes]), bs, 1e-2, 1e-2)


def test_split_forward_and_gradient(oracle):
    # test_split_layer.cpp:66-77 (the tops ARE the bottom: split_layer.cpp:28-34 shares the data) and :79-86
    x = np.random.default_rng(1701).standard_normal((2, 3, 6, 5)).astype(np.float32)
    layer_gradient(lambda b: [b[0], b[0]], lambda b, d: [oracle.split_bwd(d)], [x], 1e-2, 1e-2, eltwise=True)
    d = [np.random.default_rng(k).standard_normal(x.shape).astype(np.float32) for k in range(3)]
    assert np.array_equal(oracle.split_bwd(d), (d[0] + d[1]) + d[2])


def test_flatten_is_a_view(oracle):
    # test_flatten_layer.cpp:40-63: (2,3,6,5) -> (2,90,1,1), top(n, c) == bottom(n, c / 30, (c / 5) % 6, c % 5): the
    # row-major reshape the oracle relies on when it reads data-layer channels as rows of F values.
    x = np.random.default_rng(1701).standard_normal((2, 3, 6, 5)).astype(np.float32)
    top = x.reshape(2, 90)
    for c in range(90):
        assert top[0, c] == x[0, c // 30, (c // 5) % 6, c % 5]
        assert top[1, c] == x[1, c // 30, (c // 5) % 6, c % 5]


@pytest.fixture
def ip_bottom():
    # test_inner_product_layer.cpp:24-33: (2,3,4,5) uniform [0,1)
    return np.random.default_rng(1701).random((2, 3, 4, 5)).astype(np.float32)


def test_inner_product_forward(oracle, ip_bottom):
    # test_inner_product_layer.cpp:43-56 (top is (2,10,1,1)) and :58-86 (uniform weights in [0,1), bias in
    # [1,2): every output >= 1)
    rng = np.random.default_rng(3)
    W = rng.random((10, 60)).astype(np.float32)
    b = (1 + rng.random(10)).astype(np.float32)
    y = oracle.inner_product_fwd(ip_bottom, W, b)
    assert y.shape == (2, 10)
    assert np.all(y >= 1.0)
    ref = ip_bottom.reshape(2, 60).astype(np.float64) @ W.T.astype(np.float64) + b
    assert np.allclose(y, ref, rtol=1e-5, atol=1e-5)


def test_inner_product_gradient(oracle, ip_bottom):
    # test_inner_product_layer.cpp:88-111: gaussian weights and bias, GradientChecker(1e-2, 1e-3), exhaustive over
    # the bottom AND the two parameter blobs (test_gradient_check_util.hpp:86-93 adds the layer's blobs)
    rng = np.random.default_rng(5)
    W = rng.standard_normal((10, 60)).astype(np.float32)
    b = rng.standard_normal(10).astype(np.float32)

    def fwd(bl):
        return [oracle.inner_product_fwd(bl[0], bl[1], bl[2])]

    def bwd(bl, d):
        dW, db, dX = oracle.inner_product_bwd(bl[0], bl[1], d[0])
        return [dX.reshape(bl[0].shape), dW, db]

    layer_gradient(fwd, bwd, [ip_bottom, W, b], 1e-2, 1e-3)
