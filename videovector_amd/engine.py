"""ctypes binding of include/videovec.h and a small host-side engine object.

Mirrors the reference's solver-side view of the path (Solver::Solve loop body, solver.cpp:194-220):
`forward_backward()` == Net::ForwardBackward, `apply_update()` == ComputeUpdateValue + Net::Update.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PREC = {"f16": 0, "bf16": 1}


class VVError(RuntimeError):
    pass


def lib_path():
    """The product library.  VV_LIB names another build of the same ABI (tools/lab/*: the -DVV_LAB build with the ablated
    kernels; A/B of two builds); it must exist -- nothing falls back."""
    return os.environ.get("VV_LIB") or os.path.join(_HERE, "lib", "libvideovec.so")


class _StepCfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("Nn", C.c_int32),
                ("margin", C.c_float), ("norm", C.c_int32), ("loss_weight", C.c_float),
                ("ctx_coeff", C.c_void_p),
                ("dropout_ratio", C.c_float), ("dropout_mask", C.c_void_p),
                ("dropout_seed", C.c_uint64), ("global_count", C.c_int64),
                ("lr", C.c_float), ("momentum", C.c_float), ("weight_decay", C.c_float),
                ("lr_mult", C.c_float * 2), ("decay_mult", C.c_float * 2), ("reg", C.c_int32),
                ("solver_type", C.c_int32), ("delta", C.c_float), ("ip_regularization", C.c_float),
                ("item_weight", C.c_void_p)]


_lib = None


class _BoxProbe(C.Structure):
    _fields_ = [("gemm_tflops", C.c_double), ("gemm_ms", C.c_double), ("gemm_clock_mhz", C.c_double),
                ("copy_tbs", C.c_double), ("copy_ms", C.c_double),
                ("gemm_rows", C.c_int32), ("gemm_k", C.c_int32), ("gemm_n", C.c_int32), ("gemm_launches", C.c_int32),
                ("copy_bytes", C.c_int64)]


def load_library():
    """Load libvideovec.so; raises (never falls back) when the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own libamdhip64 with the same SONAME as /opt/rocm's.  Whichever is
    # loaded first serves the whole process, and torch fails ("no ROCm-capable device") when it finds the
    # system runtime already loaded: import torch first so streams / tensors / RCCL and this library
    # share ONE HIP runtime.
    if "torch" not in __import__("sys").modules and not os.environ.get("VV_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    p = lib_path()
    if not os.path.exists(p):
        raise VVError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
    L = C.CDLL(p)
    L.vv_last_error.restype = C.c_char_p
    L.vv_version.restype = C.c_char_p
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    sigs = {
        "vv_create": [C.c_int, C.c_int, C.POINTER(vp)],
        "vv_destroy": [vp], "vv_set_stream": [vp, vp], "vv_synchronize": [vp],
        "vv_device_query": [C.c_int, C.c_char_p, C.c_size_t],
        "vv_set_option": [vp, C.c_char_p, C.c_double], "vv_get_option": [vp, C.c_char_p, C.POINTER(C.c_double)],
        "vv_set_dedup": [vp, C.c_int], "vv_dedup_stats": [vp, C.POINTER(i64), C.POINTER(i64)],
        "vv_grad_scale_stats": [vp, C.POINTER(i64), C.POINTER(C.c_float)],
        "vv_table_set": [vp, vp, i64, i32], "vv_table_synth": [vp, C.c_uint64, i64, i32],
        "vv_table_get": [vp, vp, i64, vp],
        "vv_params_set": [vp, i32, vp, vp, vp, vp], "vv_params_get": [vp, vp, vp, vp, vp],
        "vv_step_cfg_default": [vp],
        "vv_forward_backward": [vp, vp, vp, C.c_int], "vv_apply_update": [vp, vp],
        "vv_forward_backward_q1": [vp, vp, vp, vp],
        "vv_step": [vp, vp, vp, C.c_int],
        "vv_update_hint": [vp, vp],
        "vv_forward_backward_ring": [vp, vp, vp, i32, i32, vp, C.c_double],
        "vv_dev_alloc": [vp, C.c_size_t, C.POINTER(vp)], "vv_dev_free": [vp, vp],
        "vv_dev_upload": [vp, vp, vp, C.c_size_t], "vv_dev_download": [vp, vp, vp, C.c_size_t],
        "vv_dev_memset": [vp, vp, C.c_int, C.c_size_t],
        "vv_op_copy2d": [vp, vp, i64, vp, i64, i64, i64, C.c_int],
        "vv_op_axpby": [vp, i64, f32, vp, f32, vp], "vv_op_mul": [vp, i64, vp, vp, vp, C.c_int],
        "vv_op_relu": [vp, i64, vp, vp, f32], "vv_op_relu_bwd": [vp, i64, vp, vp, vp, f32],
        "vv_op_dropout": [vp, i64, vp, vp, vp, f32, C.c_uint64, C.c_int],
        "vv_op_rowsum": [vp, i64, i32, vp, i32, vp], "vv_op_rowsum_bwd": [vp, i64, i32, i32, vp, vp],
        "vv_op_normalize": [vp, i64, i32, vp, vp], "vv_op_normalize_bwd": [vp, i64, i32, vp, vp, vp],
        "vv_op_max_margin": [vp, i32, vp, vp, vp, f32, i32, C.POINTER(f32), C.POINTER(f32)],
        "vv_op_max_margin_bwd": [vp, i32, vp, vp, vp, f32, i32, f32, vp, vp],
        "vv_op_gather_rows": [vp, vp, i64, vp],
        "vv_op_inner_product": [vp, vp, i64, vp], "vv_op_inner_product_bwd": [vp, vp, i64, f32],
        "vv_comm_init": [vp, i32, i32, C.c_char_p, i32], "vv_comm_overlap": [vp, C.c_int], "vv_comm_schedule": [vp, C.c_int],
        "vv_allreduce_grads": [vp], "vv_comm_destroy": [vp],
        "vv_loss_get": [vp, C.POINTER(f32), C.POINTER(f32)],
        "vv_grads_device": [vp, C.POINTER(vp), C.POINTER(i64)], "vv_grads_get": [vp, vp, vp],
        "vv_grads_bind": [vp, vp],
        "vv_blobs_get": [vp, vp, vp, vp, vp],
        "vv_embed": [vp, vp, i64, C.c_int, C.c_int, vp],
        "vv_embed_mean": [vp, vp, i64, i32, vp, C.c_int, C.c_int, vp],
        "vv_retrieval_stats": [vp, vp, i32, i32, vp, vp, vp, i32, C.c_int, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)],
        "vv_profile_enable": [vp, C.c_int],
        "vv_profile_select": [vp, C.c_char_p],
        "vv_profile_get": [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)],
        "vv_box_probe": [vp, C.POINTER(_BoxProbe)],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = None if name == "vv_step_cfg_default" else C.c_int
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class StepConfig:
    """vv_step_cfg with the shipped project defaults (vv_step_cfg_default)."""

    def __init__(self, B, C_, Nn, **kw):
        self.c = _StepCfg()
        load_library().vv_step_cfg_default(C.byref(self.c))
        self.c.B, self.c.C, self.c.Nn = B, C_, Nn
        self._keep = {}
        for k, v in kw.items():
            self.set(k, v)

    def set(self, k, v):
        if k == "ctx_coeff":
            a = None if v is None else np.ascontiguousarray(v, dtype=np.float32)
            self._keep[k] = a
            self.c.ctx_coeff = None if a is None else a.ctypes.data
        elif k == "dropout_mask":
            a = None if v is None else np.ascontiguousarray(v, dtype=np.uint8)
            self._keep[k] = a
            self.c.dropout_mask = None if a is None else a.ctypes.data
        elif k == "item_weight":
            a = None if v is None else np.ascontiguousarray(v, dtype=np.float32)
            self._keep[k] = a
            self.c.item_weight = None if a is None else a.ctypes.data
        elif k in ("lr_mult", "decay_mult"):
            getattr(self.c, k)[0], getattr(self.c, k)[1] = float(v[0]), float(v[1])
        elif k == "reg":
            self.c.reg = {"L1": 1, "L2": 2}.get(v, v)
        elif k == "solver_type":
            self.c.solver_type = {"SGD": 0, "NESTEROV": 1, "ADAGRAD": 2}.get(v, v)
        elif k == "norm":
            self.c.norm = {"L1": 1, "L2": 2}.get(v, v)
        else:
            if not hasattr(self.c, k):
                raise AttributeError(k)
            setattr(self.c, k, v)
        return self


class Engine:
    """One context per process / GPU (vv_create)."""

    def __init__(self, device=0, prec="f16"):
        self.L = load_library()
        self.h = C.c_void_p()
        self.prec = prec
        self._chk(self.L.vv_create(device, PREC[prec], C.byref(self.h)))
        self.F = self.D = 0
        self.n_rows = 0

    def _chk(self, rc):
        if rc != 0:
            raise VVError("videovec error %d: %s" % (rc, self.L.vv_last_error().decode()))

    def close(self):
        if getattr(self, "h", None) and self.h:
            self.L.vv_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup
    def set_stream(self, raw_stream):
        self._chk(self.L.vv_set_stream(self.h, C.c_void_p(raw_stream)))

    def synchronize(self):
        self._chk(self.L.vv_synchronize(self.h))

    def set_option(self, name, value):
        """Per-context execution switch by name (include/videovec.h: vv_set_option)."""
        self._chk(self.L.vv_set_option(self.h, name.encode(), float(value)))

    def get_option(self, name):
        v = C.c_double(0)
        self._chk(self.L.vv_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def set_dedup(self, on):
        """Row de-duplication of the batch (include/videovec.h: vv_set_dedup); default on."""
        self._chk(self.L.vv_set_dedup(self.h, int(bool(on))))

    def dedup_stats(self):
        """(rows, distinct rows) of the last forward/backward pass."""
        r, u = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.vv_dedup_stats(self.h, C.byref(r), C.byref(u)))
        return r.value, u.value

    def grad_scale_stats(self):
        """(steps whose 16-bit gradients had to be produced again at a smaller scale, current scale); videovec.h."""
        r, sc = C.c_int64(0), C.c_float(0)
        self._chk(self.L.vv_grad_scale_stats(self.h, C.byref(r), C.byref(sc)))
        return r.value, sc.value

    def table_set(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        self.n_rows, self.F = rows.shape
        self._chk(self.L.vv_table_set(self.h, _ptr(rows), rows.shape[0], rows.shape[1]))

    def table_synth(self, seed, n_rows, F):
        self.n_rows, self.F = n_rows, F
        self._chk(self.L.vv_table_synth(self.h, seed, n_rows, F))

    def table_get(self, rows=None, n=None):
        r = None if rows is None else np.ascontiguousarray(rows, dtype=np.int32)
        n = len(r) if r is not None else n
        out = np.empty((n, self.F), np.float32)
        self._chk(self.L.vv_table_get(self.h, _ptr(r), n, _ptr(out)))
        return out

    def params_set(self, W, b=None, hW=None, hb=None):
        W = np.ascontiguousarray(W, dtype=np.float32)
        self.D = W.shape[0]
        assert W.shape[1] == self.F, "W must be D x F with the table's F"
        cv = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        b, hW, hb = cv(b), cv(hW), cv(hb)
        self._chk(self.L.vv_params_set(self.h, self.D, _ptr(W), _ptr(b), _ptr(hW), _ptr(hb)))

    def params_get(self):
        W = np.empty((self.D, self.F), np.float32); b = np.empty(self.D, np.float32)
        hW = np.empty_like(W); hb = np.empty_like(b)
        self._chk(self.L.vv_params_get(self.h, _ptr(W), _ptr(b), _ptr(hW), _ptr(hb)))
        return W, b, hW, hb

    # ---- iteration
    def forward_backward(self, cfg, idx=None, idx_dev_ptr=None, idx_ready=False):
        """idx: host array [B][C+Nn]; or idx_dev_ptr: device indices produced on the context's stream (ordered behind it),
        idx_ready=True when they are complete already (static batches: no ordering added)."""
        if idx_dev_ptr is not None:
            self._chk(self.L.vv_forward_backward(self.h, C.byref(cfg.c), C.c_void_p(int(idx_dev_ptr)), 2 if idx_ready else 1))
        else:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            assert idx.shape == (cfg.c.B, cfg.c.C + cfg.c.Nn)
            self._chk(self.L.vv_forward_backward(self.h, C.byref(cfg.c), _ptr(idx), 0))

    def forward_backward_q1(self, cfg, idx, last_src):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        last_src = np.ascontiguousarray(last_src, dtype=np.int32)
        assert idx.shape == last_src.shape == (cfg.c.B, cfg.c.C + cfg.c.Nn)
        self._chk(self.L.vv_forward_backward_q1(self.h, C.byref(cfg.c), _ptr(idx), _ptr(last_src)))

    def forward_backward_ring(self, cfg, ring, consumer=0, item_begin=0, label_out=None, timeout_s=60.0):
        """Next batch of the sampler's prefetch ring -> pinned staging -> async H2D -> forward/backward."""
        self._chk(self.L.vv_forward_backward_ring(self.h, C.byref(cfg.c), ring.h, consumer, item_begin, _ptr(label_out),
                                                  float(timeout_s)))

    # ---- per-layer operators (vv_dev_* / vv_op_*): device buffers are DevBuf objects
    def dev(self, array_or_shape):
        """A device buffer: from a float32 / uint8 array (uploaded) or a shape (zeros, float32)."""
        return DevBuf(self, array_or_shape)

    def op(self, name, *args):
        """Call vv_op_<name>(ctx, ...); DevBuf arguments pass their device pointer."""
        conv = [a.ptr if isinstance(a, DevBuf) else (_ptr(a) if isinstance(a, np.ndarray) else a) for a in args]
        self._chk(getattr(self.L, "vv_op_" + name)(self.h, *conv))

    # ---- data parallel (vv_comm_*)
    def comm_init(self, world, rank, id_path, transport="rccl"):
        kinds = {"rccl": 0, "shm": 1, "peer": 2}
        if transport not in kinds:
            raise VVError("comm_init: unknown transport %r (rccl, shm, peer)" % (transport,))
        self._chk(self.L.vv_comm_init(self.h, world, rank, None if id_path is None else id_path.encode(), kinds[transport]))

    def comm_overlap(self, on=True):
        self._chk(self.L.vv_comm_overlap(self.h, int(bool(on))))

    def comm_schedule(self, name):
        """'sync', 'overlap' or 'sharded' (include/videovec.h: vv_comm_schedule)."""
        self._chk(self.L.vv_comm_schedule(self.h, {"sync": 0, "overlap": 1, "sharded": 2}[name]))

    def allreduce_grads(self):
        self._chk(self.L.vv_allreduce_grads(self.h))

    def comm_destroy(self):
        self._chk(self.L.vv_comm_destroy(self.h))

    def update_hint(self, cfg):
        """The next forward_backward* is followed by apply_update(cfg) and nothing reads the gradient in between (vv_update_hint)."""
        self._chk(self.L.vv_update_hint(self.h, C.byref(cfg.c)))

    def apply_update(self, cfg):
        self._chk(self.L.vv_apply_update(self.h, C.byref(cfg.c)))

    def step(self, cfg, idx=None, idx_dev_ptr=None):
        self.update_hint(cfg)                      # (what vv_step does: nothing reads the gradient between the two calls)
        self.forward_backward(cfg, idx, idx_dev_ptr)
        self.apply_update(cfg)

    def loss(self):
        l, v = C.c_float(), C.c_float()
        self._chk(self.L.vv_loss_get(self.h, C.byref(l), C.byref(v)))
        return l.value, v.value

    def grads_device(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.vv_grads_device(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def grads_bind(self, dev_ptr):
        self._chk(self.L.vv_grads_bind(self.h, C.c_void_p(dev_ptr)))

    def grads(self):
        dW = np.empty((self.D, self.F), np.float32); db = np.empty(self.D, np.float32)
        self._chk(self.L.vv_grads_get(self.h, _ptr(dW), _ptr(db)))
        return dW, db

    def blobs(self, cfg, ip2=True, scores=True, ip1_diff=False):
        B, CN, Nn = cfg.c.B, cfg.c.C + cfg.c.Nn, cfg.c.Nn
        out = {}
        a = np.empty((CN * B, self.D), np.float32) if ip2 else None
        st = np.empty((B, Nn), np.float32) if scores else None
        sn = np.empty((B, Nn), np.float32) if scores else None
        dy = np.empty((CN * B, self.D), np.float32) if ip1_diff else None
        self._chk(self.L.vv_blobs_get(self.h, _ptr(a), _ptr(st), _ptr(sn), _ptr(dy)))
        if ip2: out["ip2"] = a
        if scores: out["target_score"], out["negative_scores"] = st, sn
        if ip1_diff: out["ip1_diff"] = dy
        return out

    def embed(self, rows=None, n=None, relu=True, l2norm=False):
        r = None if rows is None else np.ascontiguousarray(rows, dtype=np.int32)
        n = len(r) if r is not None else n
        out = np.empty((n, self.D), np.float32)
        self._chk(self.L.vv_embed(self.h, _ptr(r), n, int(relu), int(l2norm), _ptr(out)))
        return out

    def embed_mean(self, rows, coeff=None, relu=True, l2norm=False):
        """TEST-branch embedding: fc7(+ReLU)(+normalise) of the weighted sum of k rows per sample."""
        r = np.ascontiguousarray(rows, dtype=np.int32)
        n, k = r.shape
        cf = None if coeff is None else np.ascontiguousarray(coeff, dtype=np.float32)
        out = np.empty((n, self.D), np.float32)
        self._chk(self.L.vv_embed_mean(self.h, _ptr(r), n, k, _ptr(cf), int(relu), int(l2norm), _ptr(out)))
        return out

    def retrieval_stats(self, feat, video_ids, id2class, exclude_same_video=True):
        """RetrievalStatsLayer forward: (mAP, hit@1, hit@5)."""
        feat = np.ascontiguousarray(feat, dtype=np.float32)
        vid = np.ascontiguousarray(video_ids, dtype=np.int32)
        mi = np.ascontiguousarray(list(id2class.keys()), dtype=np.int32)
        mc = np.ascontiguousarray(list(id2class.values()), dtype=np.int32)
        o = [C.c_float(), C.c_float(), C.c_float()]
        self._chk(self.L.vv_retrieval_stats(self.h, _ptr(feat), feat.shape[0], feat.shape[1], _ptr(vid), _ptr(mi),
                                            _ptr(mc), len(mi), int(exclude_same_video), *[C.byref(x) for x in o]))
        return tuple(x.value for x in o)

    # ---- profiling
    def profile_enable(self, on=True):
        self._chk(self.L.vv_profile_enable(self.h, int(on)))

    def profile_select(self, kernels=None):
        """Time only these kernels (iterable of names); None = all."""
        self._chk(self.L.vv_profile_select(self.h, ",".join(kernels).encode() if kernels else None))

    def profile_get(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self.L.vv_profile_get(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def box_probe(self):
        """vv_box_probe: what this device delivers right now on two fixed probes (the benchmark's forward GEMM instantiation on contiguous
        random rows; a 1 GiB streaming copy) -- the calibration record of a bench line."""
        r = _BoxProbe()
        self._chk(self.L.vv_box_probe(self.h, C.byref(r)))
        return {"gemm_tflops": r.gemm_tflops, "gemm_ms": r.gemm_ms, "gemm_clock_mhz": r.gemm_clock_mhz,
                "copy_tbs": r.copy_tbs, "copy_ms": r.copy_ms,
                "gemm_shape": [r.gemm_rows, r.gemm_k, r.gemm_n], "gemm_launches": r.gemm_launches, "copy_bytes": r.copy_bytes}


class DevBuf:
    """Device memory of an Engine's context (vv_dev_alloc): the GPU side of a Blob."""

    def __init__(self, eng, array_or_shape):
        self.eng = eng
        if isinstance(array_or_shape, np.ndarray):
            a = np.ascontiguousarray(array_or_shape)
            self.shape, self.dtype = a.shape, a.dtype
        else:
            a = None
            self.shape, self.dtype = tuple(np.atleast_1d(array_or_shape)), np.dtype(np.float32)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        eng._chk(eng.L.vv_dev_alloc(eng.h, max(self.nbytes, 4), C.byref(p)))
        self.ptr = C.c_void_p(p.value)
        if a is not None and self.nbytes:
            eng._chk(eng.L.vv_dev_upload(eng.h, self.ptr, _ptr(a), self.nbytes))

    def get(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            self.eng._chk(self.eng.L.vv_dev_download(self.eng.h, _ptr(out), self.ptr, self.nbytes))
        return out

    def set(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.shape == self.shape
        self.eng._chk(self.eng.L.vv_dev_upload(self.eng.h, self.ptr, _ptr(a), self.nbytes))

    def free(self):
        if self.ptr is not None and getattr(self.eng, "h", None):
            self.eng.L.vv_dev_free(self.eng.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _SamplerParam(C.Structure):
    _fields_ = [("batch_size", C.c_int32), ("context_size", C.c_int32),
                ("num_negative_samples", C.c_int32), ("max_buffer_size", C.c_int32),
                ("negative_swap_percentage", C.c_int32), ("max_same_video_negs", C.c_int32),
                ("max_tries_for_negs", C.c_int32), ("context_type", C.c_int32), ("initial_cursor", C.c_int32),
                ("output_shot_distance", C.c_int32), ("max_shot_distance", C.c_float), ("rand_seed", C.c_int32)]


CONTEXT_TYPES = {"WINDOW": 0, "PAST": 1, "PAST_CONTINUOUS": 2, "PAST_CONTINUOUS_FIXED": 3, "PAIRWISE": 4}


class Sampler:
    """Host-side triplet sampler of the product library (vv_sampler_*): the reference's
    VideoSampledShotsDataLayer with row indices instead of feature copies.  negatives = (video_id, n_shots,
    row_base[, shot_ids]) of a `negative_dataset` whose rows live in the same feature table."""

    def __init__(self, video_id, n_shots, row_base, *, batch_size, context_size,
                 num_negative_samples, max_buffer_size=5000, negative_swap_percentage=50,
                 max_same_video_negs=0, max_tries_for_negs=100, shot_ids=None, context_type="WINDOW", initial_cursor=0,
                 output_shot_distance=False, max_shot_distance=5.0, negatives=None, rand_seed=1):
        L = load_library()
        L.vv_sampler_create_neg.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_void_p)]
        L.vv_sampler_next.argtypes = [C.c_void_p] * 4
        L.vv_sampler_destroy.argtypes = [C.c_void_p]
        L.vv_sampler_prefetch_start.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]
        L.vv_sampler_prefetch_stop.argtypes = [C.c_void_p]
        L.vv_sampler_ring.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        self.L = L
        vid = np.ascontiguousarray(video_id, dtype=np.int32)
        ns = np.ascontiguousarray(n_shots, dtype=np.int32)
        rb = np.ascontiguousarray(row_base, dtype=np.int64)
        sid = None if shot_ids is None else np.ascontiguousarray(shot_ids, dtype=np.int32)
        p = _SamplerParam(batch_size, context_size, num_negative_samples, max_buffer_size,
                          negative_swap_percentage, max_same_video_negs, max_tries_for_negs,
                          CONTEXT_TYPES[context_type], initial_cursor, int(output_shot_distance), max_shot_distance, rand_seed)
        self.h = C.c_void_p()
        nvid = nns = nrb = nsid = None
        if negatives is not None:
            nvid = np.ascontiguousarray(negatives[0], dtype=np.int32)
            nns = np.ascontiguousarray(negatives[1], dtype=np.int32)
            nrb = np.ascontiguousarray(negatives[2], dtype=np.int64)
            nsid = np.ascontiguousarray(negatives[3], dtype=np.int32) if len(negatives) > 3 and negatives[3] is not None else None
        rc = L.vv_sampler_create_neg(C.byref(p), len(vid), _ptr(vid), _ptr(ns), _ptr(rb), _ptr(sid),
                                     0 if nvid is None else len(nvid), _ptr(nvid), _ptr(nns), _ptr(nrb), _ptr(nsid),
                                     C.byref(self.h))
        if rc != 0:
            raise VVError("vv_sampler_create failed (%d): the reference would CHECK-fail on these "
                          "parameters" % rc)
        if context_type == "PAIRWISE":
            context_size = 2
        self.B, self.CN = batch_size, context_size + num_negative_samples

    def next(self, want_last=False, want_label=False):
        idx = np.empty((self.B, self.CN), np.int32)
        last = np.empty((self.B, self.CN), np.int32) if want_last else None
        label = np.empty((self.B,), np.int32) if want_label else None
        rc = self.L.vv_sampler_next(self.h, _ptr(idx), _ptr(last), _ptr(label))
        if rc != 0:
            raise VVError("vv_sampler_next failed (%d)" % rc)
        if want_last or want_label:
            return idx, last, label
        return idx

    def prefetch_start(self, depth=4, threads=4, shm_name=None, consumers=1):
        """Background threads keep `depth` batches ahead (BasePrefetchingDataLayer, base_data_layer.cpp:52-95); next()
        then pops finished batches.  shm_name: publish the ring in POSIX shared memory for `consumers` processes
        (BatchRing.attach)."""
        rc = self.L.vv_sampler_prefetch_start(self.h, depth, threads, None if shm_name is None else shm_name.encode(), consumers)
        if rc != 0:
            raise VVError("vv_sampler_prefetch_start failed (%d)" % rc)

    def stat(self, which):
        self.L.vv_sampler_stat.argtypes = [C.c_void_p, C.c_int32]
        self.L.vv_sampler_stat.restype = C.c_int64
        return self.L.vv_sampler_stat(self.h, which)

    def prefetch_stop(self):
        self.L.vv_sampler_prefetch_stop(self.h)

    def ring(self):
        """The producer process's own consumer handle of the prefetch ring."""
        r = C.c_void_p()
        if self.L.vv_sampler_ring(self.h, C.byref(r)) != 0:
            raise VVError("vv_sampler_ring: prefetch is not running")
        return BatchRing(r, owned=False, keep=self)

    def close(self):
        if getattr(self, "h", None) and self.h:
            self.L.vv_sampler_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchRing:
    """A consumer's view of a sampler's prefetch ring (vv_batch_ring_*): one sampler per node, every data-parallel
    rank takes its items of the same global batch."""

    def __init__(self, handle, owned, keep=None):
        self.L = load_library()
        self.L.vv_batch_ring_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 4
        self.L.vv_batch_ring_next.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double]
        self.L.vv_batch_ring_detach.argtypes = [C.c_void_p]
        self.h, self.owned, self._keep = handle, owned, keep
        v = [C.c_int32() for _ in range(4)]
        self.L.vv_batch_ring_info(self.h, *[C.byref(x) for x in v])
        self.batch_size, self.CN, self.consumers, self.depth = [x.value for x in v]

    @classmethod
    def attach(cls, shm_name, timeout_s=60.0):
        L = load_library()
        L.vv_batch_ring_attach.argtypes = [C.c_char_p, C.c_double, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        rc = L.vv_batch_ring_attach(shm_name.encode(), float(timeout_s), C.byref(h))
        if rc != 0:
            raise VVError("vv_batch_ring_attach(%r) failed (%d)" % (shm_name, rc))
        return cls(h, owned=True)

    def next(self, consumer=0, item_begin=0, item_count=None, want_label=False, timeout_s=0.0, out=None):
        n = self.batch_size - item_begin if item_count is None else item_count
        idx = out if out is not None else np.empty((n, self.CN), np.int32)
        label = np.empty((n,), np.int32) if want_label else None
        rc = self.L.vv_batch_ring_next(self.h, consumer, item_begin, n, _ptr(idx), _ptr(label), float(timeout_s))
        if rc != 0:
            raise VVError("vv_batch_ring_next failed (%d): producer gone or timeout" % rc)
        return (idx, label) if want_label else idx

    def close(self):
        if self.owned and self.h:
            self.L.vv_batch_ring_detach(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
