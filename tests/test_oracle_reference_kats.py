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
        rng = np.random.default_rng(5)
        for (M, N, K) in [(1, 1, 1), (7, 33, 5), (97, 50, 401), (200, 4100, 37), (13, 17, 800), (64, 64, 64)]:
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
