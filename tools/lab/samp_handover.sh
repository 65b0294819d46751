#!/bin/bash
# Lab: which part of the hand-over between the sampler's stage threads costs the walk its time.  Builds a lab copy of the host library's
# sampler (-DVV_SAMPLER_LAB: parts switched off by VV_SAMPLER_LAB bits, results wrong by design) and times the 8192-item pipeline.
cd "$(dirname "$0")/../.."
g++ -O3 -march=x86-64-v3 -std=c++17 -fPIC -pthread -DVV_SAMPLER_LAB -shared -o /tmp/libvv_sampler_lab.so videovector_amd/csrc/sampler.cc -lrt || exit 1
python3 - <<'PY'
import ctypes as C, os, time, numpy as np, sys
sys.path.insert(0, ".")
from videovector_amd.synth import SyntheticVideos
L = C.CDLL("/tmp/libvv_sampler_lab.so")
class P(C.Structure):
    _fields_ = [("batch_size", C.c_int32), ("context_size", C.c_int32), ("num_negative_samples", C.c_int32), ("max_buffer_size", C.c_int32),
                ("negative_swap_percentage", C.c_int32), ("max_same_video_negs", C.c_int32), ("max_tries_for_negs", C.c_int32), ("context_type", C.c_int32),
                ("initial_cursor", C.c_int32), ("output_shot_distance", C.c_int32), ("max_shot_distance", C.c_float), ("rand_seed", C.c_int32)]
ds = SyntheticVideos(seed=1701, n_videos=2048)
vid = np.ascontiguousarray(ds.video_id, np.int32); ns = np.ascontiguousarray(ds.n_shots, np.int32); rb = np.ascontiguousarray(ds.row_base, np.int64)
L.vv_sampler_create.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
L.vv_sampler_next.argtypes = [C.c_void_p] * 4
L.vv_sampler_prefetch_start.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]
L.vv_sampler_destroy.argtypes = [C.c_void_p]
B = 8192
names = {0: "everything on", 1: "slot stage: no event replay", 2: "slot stage: does not read the record's words", 4: "frame stage: idle", 8: "walk: no copy of the stream words",
         16: "walk: no event log", 3: "slot stage: neither", 7: "slot stage neither, frame stage idle", 24: "walk writes neither", 31: "all of it"}
for lab in (0, 1, 2, 4, 8, 16, 3, 7, 24, 31, 0):
    os.environ["VV_SAMPLER_LAB"] = str(lab)
    ms = []
    for r in range(5):
        p = P(); L.vv_sampler_param_default(C.byref(p)); p.batch_size = B; p.context_size = 5; p.num_negative_samples = 50; p.max_buffer_size = 5000; p.negative_swap_percentage = 50
        h = C.c_void_p()
        assert L.vv_sampler_create(C.byref(p), len(vid), vid.ctypes.data, ns.ctypes.data, rb.ctypes.data, None, C.byref(h)) == 0
        assert L.vv_sampler_prefetch_start(h, 8, 4, None, 1) == 0
        idx = np.empty((B, 55), np.int32)
        for _ in range(4): L.vv_sampler_next(h, idx.ctypes.data, None, None)
        t0 = time.perf_counter()
        for _ in range(50): L.vv_sampler_next(h, idx.ctypes.data, None, None)
        ms.append((time.perf_counter() - t0) / 50 * 1e3)
        L.vv_sampler_destroy(h)
    a = np.sort(ms)
    print("%2d %-48s min %.3f median %.3f max %.3f ms per 8192 items" % (lab, names[lab], a[0], np.median(a), a[-1]), flush=True)
PY
