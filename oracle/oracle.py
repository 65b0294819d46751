"""ctypes front-end of oracle/liboracle.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (videovector_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_PIN = os.path.join(_HERE, "pin", "stdlib_pin")


def build(force=False):
    """Compile liboracle.so and pin/stdlib_pin with the Makefile in this directory."""
    if force or not (os.path.exists(_LIB) and os.path.exists(_PIN)) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "vv_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "-s", "all"], check=True)
    return _LIB


class _Rng(C.Structure):
    _fields_ = [("r", C.c_int32 * 31), ("f", C.c_int), ("b", C.c_int)]


class _Dataset(C.Structure):
    _fields_ = [("n_videos", C.c_int32), ("video_id", C.c_void_p), ("n_shots", C.c_void_p),
                ("row_base", C.c_void_p), ("shot_ids", C.c_void_p), ("shot_off", C.c_void_p)]


class _SamplerParam(C.Structure):
    _fields_ = [("batch_size", C.c_int32), ("context_size", C.c_int32),
                ("num_negative_samples", C.c_int32), ("max_buffer_size", C.c_int32),
                ("negative_swap_percentage", C.c_int32), ("max_same_video_negs", C.c_int32),
                ("max_tries_for_negs", C.c_int32), ("context_type", C.c_int32), ("initial_cursor", C.c_int32),
                ("output_shot_distance", C.c_int32), ("max_shot_distance", C.c_float)]


CONTEXT_TYPES = {"WINDOW": 0, "PAST": 1, "PAST_CONTINUOUS": 2, "PAST_CONTINUOUS_FIXED": 3, "PAIRWISE": 4}


class _StepCfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("Nn", C.c_int32), ("F", C.c_int32),
                ("D", C.c_int32), ("margin", C.c_float), ("norm", C.c_int32),
                ("loss_weight", C.c_float), ("ctx_coeff", C.c_void_p),
                ("dropout_ratio", C.c_float), ("dropout_mask", C.c_void_p),
                ("relu_negative_slope", C.c_float), ("ip_regularization", C.c_float),
                ("global_count", C.c_int64), ("item_weight", C.c_void_p)]


class _StepOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("Y", "H", "ctx", "posneg", "s_true", "s_bogus", "dY", "dW", "db")] + \
               [("loss", C.c_float), ("violations", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_rand.restype = C.c_int32
        L.orc_sampler_create.restype = C.c_void_p
        L.orc_sampler_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.orc_sampler_create_neg.restype = C.c_void_p
        L.orc_sampler_create_neg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
        L.orc_sampler_destroy.argtypes = [C.c_void_p]
        L.orc_sampler_next.argtypes = [C.c_void_p] * 4
        L.orc_sampler_buffer_rows.restype = C.c_void_p
        L.orc_sampler_buffer_rows.argtypes = [C.c_void_p]
        L.orc_sampler_buffer_ids.restype = C.c_void_p
        L.orc_sampler_buffer_ids.argtypes = [C.c_void_p]
        L.orc_sampler_cursor.restype = C.c_int32
        L.orc_sampler_cursor.argtypes = [C.c_void_p]
        L.orc_sampler_rand_calls.restype = C.c_int64
        L.orc_sampler_rand_calls.argtypes = [C.c_void_p]
        L.orc_sgemm.argtypes = [C.c_int] * 5 + [C.c_float, C.c_void_p, C.c_void_p, C.c_float,
                                                C.c_void_p]
        L.orc_get_threads.restype = C.c_int
        L.orc_learning_rate.restype = C.c_float
        L.orc_learning_rate.argtypes = [C.c_char_p, C.c_float, C.c_float, C.c_float, C.c_int,
                                        C.c_int]
        L.orc_sgd_update.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]
        L.orc_solver_update.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                        C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float]
        V = C.c_void_p
        L.orc_relu_fwd.argtypes = [C.c_int64, V, C.c_float, V]
        L.orc_relu_bwd.argtypes = [C.c_int64, V, V, C.c_float, V]
        L.orc_dropout_fwd.argtypes = [C.c_int64, V, V, C.c_float, C.c_int, V]
        L.orc_dropout_bwd.argtypes = [C.c_int64, V, V, C.c_float, C.c_int, V]
        L.orc_eltwise_fwd.argtypes = [C.c_int, C.c_int64, C.c_int, V, V, V]
        L.orc_eltwise_bwd.argtypes = [C.c_int, C.c_int64, C.c_int, V, V, V, V, C.c_int, C.c_int, V]
        L.orc_split_pieces.argtypes = [C.c_int, C.c_int64, C.c_int, V, V, V]
        L.orc_join_pieces.argtypes = [C.c_int, C.c_int64, C.c_int, V, V, V]
        L.orc_split_bwd.argtypes = [C.c_int64, C.c_int, V, V]
        L.orc_inner_product_fwd.argtypes = [C.c_int] * 3 + [V] * 4
        L.orc_inner_product_bwd.argtypes = [C.c_int] * 3 + [V] * 6
        L.orc_normalize_fwd.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_normalize_bwd.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_max_margin_fwd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                         C.c_int, C.c_void_p, C.c_void_p]
        L.orc_max_margin_bwd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                         C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_sum_fwd.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_sum_bwd.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_forward_backward.argtypes = [C.c_void_p] * 7
        L.orc_embed.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_retrieval_stats.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 3
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------------------- RNG ---------
class Rand:
    """glibc rand() clone (seed 1 == never seeded)."""

    def __init__(self, seed=1):
        self.g = _Rng()
        lib().orc_srand(C.byref(self.g), C.c_uint(seed))

    def next(self):
        return lib().orc_rand(C.byref(self.g))

    def random_unique(self, a, n):
        a = np.ascontiguousarray(a, dtype=np.int32)
        lib().orc_random_unique(C.byref(self.g), _p(a), C.c_int(len(a)), C.c_int(n))
        return a

    def random_shuffle(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        lib().orc_random_shuffle(C.byref(self.g), _p(a), C.c_int(len(a)))
        return a


def stdlib_pin(*args):
    """Run oracle/pin/stdlib_pin (real glibc / libstdc++) and return its stdout lines."""
    build()
    out = subprocess.run([_PIN] + [str(a) for a in args], check=True, capture_output=True,
                         text=True).stdout
    return out.strip().split("\n")


# ------------------------------------------------------------------------------- sampler -----
class Sampler:
    """VideoSampledShotsDataLayer restated (context_type WINDOW / PAST / PAST_CONTINUOUS / PAST_CONTINUOUS_FIXED /
    PAIRWISE).  negatives = (video_id, n_shots, row_base[, shot_ids]) of a `negative_dataset`."""

    @staticmethod
    def _dataset(video_id, n_shots, row_base, shot_ids=None):
        vid = np.ascontiguousarray(video_id, dtype=np.int32)
        ns = np.ascontiguousarray(n_shots, dtype=np.int32)
        rb = np.ascontiguousarray(row_base, dtype=np.int64)
        sid = None if shot_ids is None else np.ascontiguousarray(shot_ids, dtype=np.int32)
        soff = None if sid is None else np.concatenate([[0], np.cumsum(ns[:-1])]).astype(np.int64)
        return _Dataset(len(vid), _p(vid), _p(ns), _p(rb), _p(sid), _p(soff)), (vid, ns, rb, sid, soff)

    def __init__(self, video_id, n_shots, row_base, *, batch_size, context_size,
                 num_negative_samples, max_buffer_size, negative_swap_percentage,
                 max_same_video_negs=0, max_tries_for_negs=100, shot_ids=None, seed=1, context_type="WINDOW", initial_cursor=0,
                 output_shot_distance=False, max_shot_distance=5.0, negatives=None):
        self._vid = np.ascontiguousarray(video_id, dtype=np.int32)
        self._ns = np.ascontiguousarray(n_shots, dtype=np.int32)
        self._rb = np.ascontiguousarray(row_base, dtype=np.int64)
        self._sid = None if shot_ids is None else np.ascontiguousarray(shot_ids, dtype=np.int32)
        self._soff = None
        if self._sid is not None:
            self._soff = np.concatenate([[0], np.cumsum(self._ns[:-1])]).astype(np.int64)
        self.ds = _Dataset(len(self._vid), _p(self._vid), _p(self._ns), _p(self._rb),
                           _p(self._sid), _p(self._soff))
        self.p = _SamplerParam(batch_size, context_size, num_negative_samples, max_buffer_size,
                               negative_swap_percentage, max_same_video_negs, max_tries_for_negs,
                               CONTEXT_TYPES[context_type], initial_cursor, int(output_shot_distance),
                               max_shot_distance)
        if negatives is None:
            self.h = lib().orc_sampler_create(C.byref(self.ds), C.byref(self.p), seed)
        else:
            self.neg, self._neg_keep = self._dataset(*negatives)
            self.h = lib().orc_sampler_create_neg(C.byref(self.ds), C.byref(self.neg), C.byref(self.p), seed)
        if not self.h:
            raise ValueError("reference would CHECK-fail for these sampler parameters")
        if context_type == "PAIRWISE":
            context_size = 2
        self.B, self.CN = batch_size, context_size + num_negative_samples
        self.max_buffer = max_buffer_size if num_negative_samples > 0 else 0

    def next(self):
        idx = np.empty((self.B, self.CN), np.int32)
        last = np.empty((self.B, self.CN), np.int32)
        label = np.empty((self.B,), np.int32)
        lib().orc_sampler_next(self.h, _p(idx), _p(last), _p(label))
        return idx, last, label

    def buffer_rows(self):
        p = lib().orc_sampler_buffer_rows(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), (self.max_buffer,)).copy()

    def buffer_ids(self):
        p = lib().orc_sampler_buffer_ids(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), (self.max_buffer,)).copy()

    def cursor(self):
        return lib().orc_sampler_cursor(self.h)

    def rand_calls(self):
        return lib().orc_sampler_rand_calls(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_sampler_destroy(self.h)
            self.h = None


# ------------------------------------------------------------------------------- layers ------
def sgemm(transA, transB, A, B, alpha=1.0, beta=0.0, Cmat=None):
    A, B = _f32(A), _f32(B)
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    out = np.zeros((M, N), np.float32) if Cmat is None else _f32(Cmat)
    lib().orc_sgemm(int(transA), int(transB), M, N, K, alpha, _p(A), _p(B), beta, _p(out))
    return out


def sgemm_isa(force_avx2=None):
    """Micro-kernel of the built-in sgemm on this CPU; force_avx2=True / False selects / releases the AVX2 kernel."""
    L = lib()
    L.orc_sgemm_isa.restype = C.c_char_p
    if force_avx2 is not None:
        L.orc_sgemm_force_isa(C.c_int(1 if force_avx2 else 0))
    return L.orc_sgemm_isa().decode()


def gemm_gflops(R, D, F, reps=2):
    """GFLOP/s of the two fc7 GEMMs of one iteration through orc_sgemm (whatever BLAS is selected): the forward
    X[R][F] W[D][F]^T (inner_product_layer.cpp:62-64) and the weight gradient dY[R][D]^T X[R][F] (:85-86)."""
    import time
    rng = np.random.default_rng(0)
    X = rng.standard_normal((R, F), dtype=np.float32)
    W = rng.standard_normal((D, F), dtype=np.float32)
    dY = rng.standard_normal((R, D), dtype=np.float32)
    sgemm(False, True, X[:64], W); sgemm(True, False, dY[:64], X[:64])
    best_f = best_w = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); sgemm(False, True, X, W); best_f = min(best_f, time.perf_counter() - t0)
        t0 = time.perf_counter(); sgemm(True, False, dY, X); best_w = min(best_w, time.perf_counter() - t0)
    fl = 2.0 * R * D * F / 1e9
    return {"forward": fl / best_f, "weight_gradient": fl / best_w, "kernel": "external cblas_sgemm" if lib().orc_has_blas() else sgemm_isa()}


def set_blas(path):
    """Route orc_sgemm through an external BLAS (a shared object exporting cblas_sgemm); None = built-in kernel.
    Returns True when the library was loaded."""
    L = lib()
    L.orc_set_blas.argtypes = [C.c_char_p]
    L.orc_set_blas.restype = C.c_int
    return L.orc_set_blas(None if not path else path.encode()) == 0 and bool(path)


def find_blas():
    """A cblas_sgemm provider of this machine, if any: MKL's single dynamic library or OpenBLAS."""
    import glob
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")      # liboracle itself uses GNU OpenMP
    for pat in ("/opt/conda/lib/libmkl_rt.so*", "/usr/lib/x86_64-linux-gnu/libopenblas.so*",
                "/usr/lib/x86_64-linux-gnu/libmkl_rt.so*", "/opt/conda/lib/libopenblas.so*"):
        for f in sorted(glob.glob(pat)):
            return f
    return None


def set_threads(n):
    lib().orc_set_threads(C.c_int(n))


def get_threads():
    return lib().orc_get_threads()


def normalize_fwd(x):
    x = _f32(x); n = x.shape[0]; y = np.empty_like(x)
    lib().orc_normalize_fwd(n, x.size // n, _p(x), _p(y))
    return y


def normalize_bwd(x, dy):
    x, dy = _f32(x), _f32(dy); n = x.shape[0]; dx = np.empty_like(x)
    lib().orc_normalize_bwd(n, x.size // n, _p(x), _p(dy), _p(dx))
    return dx


def max_margin_fwd(s_true, s_bogus, margin, norm, weight=None):
    s_true, s_bogus = _f32(s_true), _f32(s_bogus)
    w = None if weight is None else _f32(weight)
    loss, viol = C.c_float(), C.c_float()
    lib().orc_max_margin_fwd(s_true.size, _p(s_true), _p(s_bogus), _p(w), margin, norm,
                             C.byref(loss), C.byref(viol))
    return loss.value, viol.value


def max_margin_bwd(s_true, s_bogus, margin, norm, loss_weight=1.0, weight=None):
    s_true, s_bogus = _f32(s_true), _f32(s_bogus)
    w = None if weight is None else _f32(weight)
    dt, db = np.empty_like(s_true), np.empty_like(s_bogus)
    lib().orc_max_margin_bwd(s_true.size, _p(s_true), _p(s_bogus), _p(w), margin, norm,
                             loss_weight, _p(dt), _p(db))
    return dt, db


def sum_fwd(x, num_output):
    x = _f32(x); n = x.shape[0]; y = np.empty((n, num_output), np.float32)
    lib().orc_sum_fwd(n, x.size // n, num_output, _p(x), _p(y))
    return y


def sum_bwd(dy, dim):
    dy = _f32(dy); n, no = dy.shape; dx = np.empty((n, dim), np.float32)
    lib().orc_sum_bwd(n, dim, no, _p(dy), _p(dx))
    return dx


def _pp(arrs):
    """float** over a list of float32 arrays."""
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def relu_fwd(x, slope=0.0):
    x = _f32(x); y = np.empty_like(x)
    lib().orc_relu_fwd(x.size, _p(x), C.c_float(slope), _p(y))
    return y


def relu_bwd(x, dy, slope=0.0):
    x = _f32(x); dy = _f32(dy); dx = np.empty_like(x)
    lib().orc_relu_bwd(x.size, _p(x), _p(dy), C.c_float(slope), _p(dx))
    return dx


def dropout_fwd(x, mask, ratio=0.5, train=True):
    x = _f32(x); mask = np.ascontiguousarray(mask, np.uint8); y = np.empty_like(x)
    lib().orc_dropout_fwd(x.size, _p(x), _p(mask), C.c_float(ratio), int(train), _p(y))
    return y


def dropout_bwd(dy, mask, ratio=0.5, train=True):
    dy = _f32(dy); mask = np.ascontiguousarray(mask, np.uint8); dx = np.empty_like(dy)
    lib().orc_dropout_bwd(dy.size, _p(dy), _p(mask), C.c_float(ratio), int(train), _p(dx))
    return dx


ELTWISE_OPS = {"PROD": 0, "SUM": 1, "MAX": 2}


def eltwise_fwd(op, bottoms, coeff=None):
    bs = [_f32(b) for b in bottoms]; top = np.empty_like(bs[0])
    cf = _f32(coeff) if coeff is not None else None
    lib().orc_eltwise_fwd(ELTWISE_OPS[op], top.size, len(bs), _pp(bs), _p(cf) if cf is not None else None, _p(top))
    return top


def eltwise_bwd(op, bottoms, dtop, which, coeff=None, stable=True):
    bs = [_f32(b) for b in bottoms]; dtop = _f32(dtop); top = eltwise_fwd(op, bs, coeff)
    cf = _f32(coeff) if coeff is not None else None
    out = np.empty_like(bs[0])
    lib().orc_eltwise_bwd(ELTWISE_OPS[op], top.size, len(bs), _pp(bs), _p(cf) if cf is not None else None,
                          _p(top), _p(dtop), which, int(stable), _p(out))
    return out


def _piece_geometry(shape, dim):
    shape = tuple(shape) + (1,) * (4 - len(shape))
    outer = 1 if dim == 0 else shape[0]
    inner = int(np.prod(shape[dim + 1:]))
    return shape, outer, inner


def slice_fwd(bottom, dim, widths):
    """SliceLayer::Forward_cpu: a 4-d blob cut along dim 0 or 1 into tops of the given widths."""
    bottom = _f32(bottom); shape, outer, inner = _piece_geometry(bottom.shape, dim)
    assert sum(widths) == shape[dim]
    tops = [np.empty(shape[:dim] + (w,) + shape[dim + 1:], np.float32) for w in widths]
    wd = np.asarray(widths, np.int32)
    lib().orc_split_pieces(outer, inner, len(tops), _p(wd), _p(bottom), _pp(tops))
    return tops


def concat_fwd(bottoms, dim):
    """ConcatLayer::Forward_cpu along dim 0 or 1 (also SliceLayer::Backward_cpu on diffs)."""
    bs = [_f32(b).reshape(tuple(b.shape) + (1,) * (4 - b.ndim)) for b in bottoms]
    widths = [b.shape[dim] for b in bs]
    shape, outer, inner = _piece_geometry(bs[0].shape, dim)
    top = np.empty(shape[:dim] + (sum(widths),) + shape[dim + 1:], np.float32)
    wd = np.asarray(widths, np.int32)
    lib().orc_join_pieces(outer, inner, len(bs), _p(wd), _pp(bs), _p(top))
    return top


def split_bwd(dtops):
    ds = [_f32(d) for d in dtops]; out = np.empty_like(ds[0])
    lib().orc_split_bwd(out.size, len(ds), _pp(ds), _p(out))
    return out


def inner_product_fwd(X, W, b=None):
    X = _f32(X); W = _f32(W); M = X.shape[0]; K = X.size // M; N = W.shape[0]
    Y = np.empty((M, N), np.float32); bb = _f32(b) if b is not None else None
    lib().orc_inner_product_fwd(M, N, K, _p(X), _p(W), _p(bb) if bb is not None else None, _p(Y))
    return Y


def inner_product_bwd(X, W, dY):
    X = _f32(X); W = _f32(W); dY = _f32(dY); M = X.shape[0]; K = X.size // M; N = W.shape[0]
    dW = np.empty((N, K), np.float32); db = np.empty(N, np.float32); dX = np.empty((M, K), np.float32)
    lib().orc_inner_product_bwd(M, N, K, _p(X), _p(W), _p(dY), _p(dW), _p(db), _p(dX))
    return dW, db, dX


def learning_rate(policy, base_lr, gamma, power, stepsize, it):
    return lib().orc_learning_rate(policy.encode(), base_lr, gamma, power, stepsize, it)


SOLVER_TYPES = {"SGD": 0, "NESTEROV": 1, "ADAGRAD": 2}


def sgd_update(w, grad, hist, rate, lr_mult, momentum, weight_decay, decay_mult, reg="L2", solver="SGD",
               delta=1e-8):
    """In place on float32 contiguous arrays (as Blob::Update does).  solver: SGD / NESTEROV / ADAGRAD."""
    for a in (w, grad, hist):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    lib().orc_solver_update(w.size, _p(w), _p(grad), _p(hist), rate, lr_mult, momentum, weight_decay,
                            decay_mult, 2 if reg == "L2" else 1, SOLVER_TYPES[solver], delta)


def forward_backward(table, idx, W, b, *, C_, Nn, margin=2.0, norm=2, loss_weight=1.0,
                     ctx_coeff=None, dropout_ratio=0.0, dropout_mask=None, last_src=None,
                     global_count=0, ip_regularization=0.0, item_weight=None, want=("dW", "db")):
    """One Net::ForwardBackward.  Returns a dict with loss, violations and the requested arrays
    (names of orc_step_out)."""
    table, W = _f32(table), _f32(W)
    b = None if b is None else _f32(b)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    B, CN = idx.shape
    assert CN == C_ + Nn
    D, F = W.shape
    assert table.shape[1] == F
    coeff = _f32(np.full(C_ - 1, 1.0 / (C_ - 1)) if ctx_coeff is None else ctx_coeff)
    mask = None if dropout_mask is None else np.ascontiguousarray(dropout_mask, dtype=np.uint8)
    last = None if last_src is None else np.ascontiguousarray(last_src, dtype=np.int32)
    iw = None if item_weight is None else _f32(item_weight)
    cfg = _StepCfg(B, C_, Nn, F, D, margin, norm, loss_weight, _p(coeff), dropout_ratio, _p(mask),
                   0.0, ip_regularization, global_count, _p(iw))
    R, Q = CN * B, 1 + Nn
    shapes = dict(Y=(R, D), H=(R, D), ctx=(B, D), posneg=(Q * B, D), s_true=(B, Nn),
                  s_bogus=(B, Nn), dY=(R, D), dW=(D, F), db=(D,))
    arrs = {k: np.zeros(shapes[k], np.float32) for k in want}
    out = _StepOut()
    for k, a in arrs.items():
        setattr(out, k, a.ctypes.data)
    lib().orc_forward_backward(C.byref(cfg), _p(table), _p(idx), _p(last), _p(W), _p(b),
                               C.byref(out))
    res = dict(arrs)
    res["loss"], res["violations"] = out.loss, out.violations
    return res


def embed(table, rows, W, b, relu=True, l2norm=False):
    table, W = _f32(table), _f32(W)
    b = None if b is None else _f32(b)
    rows = None if rows is None else np.ascontiguousarray(rows, dtype=np.int32)
    n = table.shape[0] if rows is None else len(rows)
    D, F = W.shape
    out = np.empty((n, D), np.float32)
    lib().orc_embed(n, F, D, _p(table), _p(rows), _p(W), _p(b), int(relu), int(l2norm), _p(out))
    return out


def retrieval_stats(feat, video_ids, id2class, exclude_same_video=True):
    """RetrievalStatsLayer forward: returns (mAP, hit@1, hit@5).  id2class: dict video_id -> class."""
    feat = _f32(feat)
    vid = np.ascontiguousarray(video_ids, dtype=np.int32)
    mi = np.ascontiguousarray(list(id2class.keys()), dtype=np.int32)
    mc = np.ascontiguousarray(list(id2class.values()), dtype=np.int32)
    out = [C.c_float(), C.c_float(), C.c_float()]
    lib().orc_retrieval_stats(feat.shape[0], feat.shape[1], _p(feat), _p(vid), _p(mi), _p(mc), len(mi),
                              int(exclude_same_video), *[C.byref(o) for o in out])
    return tuple(o.value for o in out)
