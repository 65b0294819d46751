"""CPU tests of the product's host side: the C-ABI library loads, exports every symbol declared in
include/videovec.h (no compute without a GPU), fails loudly without a device, and its triplet
sampler is bit-identical to the oracle's restatement of the reference sampler."""
import os
import re

import numpy as np
import pytest

import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "videovec.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(vv_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    L = vv.load_library()
    for n in sorted(names):
        assert hasattr(L, n), n


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(vv.VVError) as e:
        vv.Engine(0)
    assert "no HIP device" in str(e.value) or "no CPU path" in str(e.value)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "videovector_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "vv_oracle" not in src and "liboracle" not in src, f
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


@pytest.mark.parametrize("B,C,Nn,mb,swap,max_same", [
    (32, 5, 2, 100, 50, 0), (16, 5, 10, 200, 50, 6), (8, 3, 4, 64, 99, 2), (8, 7, 3, 50, 0, 0),
    (64, 5, 50, 1000, 50, 0),
])
def test_sampler_bit_exact_vs_oracle(oracle, B, C, Nn, mb, swap, max_same):
    ds = SyntheticVideos(seed=7, n_videos=120, lo=2, span=40)
    kw = dict(batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=mb,
              negative_swap_percentage=swap, max_same_video_negs=max_same)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    for _ in range(8):
        i1, l1, y1 = a.next(want_last=True, want_label=True)
        i2, l2, y2 = o.next()
        assert np.array_equal(i1, i2) and np.array_equal(l1, l2) and np.array_equal(y1, y2)


@pytest.mark.parametrize("ctype", ["PAST", "PAST_CONTINUOUS", "PAST_CONTINUOUS_FIXED"])
def test_sampler_past_context_modes_bit_exact_vs_oracle(oracle, ctype):
    ds = SyntheticVideos(seed=7, n_videos=120, lo=2, span=40)
    for (B, C, Nn, mb, swap, max_same) in [(16, 5, 10, 200, 50, 6), (8, 4, 4, 64, 99, 2), (32, 2, 3, 50, 0, 3)]:
        kw = dict(batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=mb,
                  negative_swap_percentage=swap, max_same_video_negs=max_same, context_type=ctype)
        a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
        o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
        for _ in range(6):
            i1, l1, y1 = a.next(want_last=True, want_label=True)
            i2, l2, y2 = o.next()
            assert np.array_equal(i1, i2) and np.array_equal(l1, l2) and np.array_equal(y1, y2)


def test_sampler_initial_cursor_bit_exact_vs_oracle(oracle):
    ds = SyntheticVideos(seed=7, n_videos=120, lo=2, span=40)
    for skip in (1, 77, 500):
        kw = dict(batch_size=16, context_size=5, num_negative_samples=10, max_buffer_size=200, negative_swap_percentage=50,
                  max_same_video_negs=6, initial_cursor=skip)
        a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
        o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
        for _ in range(4):
            i1, l1, y1 = a.next(want_last=True, want_label=True)
            i2, l2, y2 = o.next()
            assert np.array_equal(i1, i2) and np.array_equal(l1, l2) and np.array_equal(y1, y2)


def test_sampler_with_explicit_shot_ids_and_errors(oracle):
    ds = SyntheticVideos(seed=9, n_videos=30)
    sid = np.concatenate([np.arange(n)[::-1] * 3 for n in ds.n_shots]).astype(np.int32)
    kw = dict(batch_size=8, context_size=5, num_negative_samples=4, max_buffer_size=60,
              negative_swap_percentage=30)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, shot_ids=sid, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, shot_ids=sid, **kw)
    for _ in range(5):
        assert np.array_equal(a.next(), o.next()[0])
    with pytest.raises(vv.VVError):       # even context size (...data_layer.cpp:434)
        vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=4, context_size=4,
                   num_negative_samples=2, max_buffer_size=10)
    with pytest.raises(vv.VVError):       # buffer larger than the number of unique shots (:344)
        vv.Sampler(ds.video_id[:2], ds.n_shots[:2], ds.row_base[:2], batch_size=4, context_size=3,
                   num_negative_samples=2, max_buffer_size=1000)
