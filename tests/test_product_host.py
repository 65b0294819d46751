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


def test_sampler_rand_seed_bit_exact_vs_oracle(oracle):
    """Per-rank samplers: rand_seed = srand() argument of the draw stream (the oracle's stream is pinned against the
    real libc's srand / rand for other seeds in test_oracle_rng.py); cursor offset as a rank would use it."""
    ds = SyntheticVideos(seed=7, n_videos=160, lo=2, span=40)
    for seed, skip, same in [(2, 0, 0), (5, 40, 0), (123456789, 77, 3), (2147483646, 3, 0)]:
        kw = dict(batch_size=32, context_size=5, num_negative_samples=10, max_buffer_size=300, negative_swap_percentage=50,
                  max_same_video_negs=same, initial_cursor=skip)
        a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, rand_seed=seed, **kw)
        b = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, rand_seed=seed, **kw)
        b.prefetch_start(depth=4, threads=3)
        o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, seed=seed, **kw)
        for _ in range(6):
            i1, l1, y1 = a.next(want_last=True, want_label=True)
            i2, l2, y2 = o.next()
            assert np.array_equal(i1, i2) and np.array_equal(l1, l2) and np.array_equal(y1, y2)
            assert np.array_equal(b.next(), i2)
        b.prefetch_stop()
    # seed 0 and seed 1 are the reference's never-seeded stream
    kw = dict(batch_size=8, context_size=5, num_negative_samples=4, max_buffer_size=64, negative_swap_percentage=50)
    ref = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw).next()
    for seed in (0, 1):
        assert np.array_equal(vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, rand_seed=seed, **kw).next(), ref)
    assert not np.array_equal(vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, rand_seed=2, **kw).next(), ref)
    with pytest.raises(vv.VVError):
        vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, rand_seed=-3, **kw)


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


def test_more_same_video_negatives_than_negative_slots_is_rejected(oracle):
    """video_sampled_shots_data_layer.cpp:484-502 never bounds `added` by num_negative_samples: with the shipped
    max_same_video_negs 6 and BASELINE config 1's 2 negatives the reference writes past the item's channels.  Both the
    product and the oracle refuse the pair instead of corrupting the next item's slots."""
    ds = SyntheticVideos(seed=7, n_videos=120, lo=8, span=40)
    kw = dict(batch_size=32, context_size=5, num_negative_samples=2, max_buffer_size=100, negative_swap_percentage=50,
              max_same_video_negs=6)
    with pytest.raises(vv.VVError):
        vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    with pytest.raises(ValueError):
        oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    kw["max_same_video_negs"] = 2        # the largest legal value fills every negative slot from the video itself
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    for _ in range(6):
        i1, l1, _ = a.next(want_last=True, want_label=True)
        i2, l2, _ = o.next()
        assert np.array_equal(i1, i2) and np.array_equal(l1, l2)


@pytest.mark.parametrize("threads", [1, 2, 3, 4])
@pytest.mark.parametrize("ctype", ["WINDOW", "PAST", "PAST_CONTINUOUS", "PAST_CONTINUOUS_FIXED"])
def test_prefetch_pipeline_is_the_same_stream(oracle, threads, ctype):
    """vv_sampler_prefetch_start (BasePrefetchingDataLayer's thread, base_data_layer.cpp:52-95): whatever the number of
    stage threads, the popped batches are the oracle's, bit for bit, labels included."""
    ds = SyntheticVideos(seed=11, n_videos=300, lo=2, span=70)      # some videos shorter than C, some longer than 64 shots
    kw = dict(batch_size=48, context_size=5, num_negative_samples=20, max_buffer_size=700,
              negative_swap_percentage=50, context_type=ctype)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    assert a.stat(1) == 1
    a.prefetch_start(depth=3, threads=threads)
    assert a.stat(2) == min(threads, 4)
    for _ in range(40):
        i1, l1, y1 = a.next(want_last=True, want_label=True)
        i2, l2, y2 = o.next()
        assert np.array_equal(i1, i2) and np.array_equal(l1, l2) and np.array_equal(y1, y2)
    a.prefetch_stop()
    a.close()


def test_prefetch_general_path_and_restarts(oracle):
    """(a) same-video negatives take the general path: prefetch then runs whole batches on one thread, last_src included.
    (b) a buffer that holds most of a small dataset makes swap-ins evict later shots of the video being walked: the
    bit-parallel walk must restart there (stat 0 counts it) and still match the oracle."""
    ds = SyntheticVideos(seed=5, n_videos=40, lo=6, span=30)
    kw = dict(batch_size=16, context_size=5, num_negative_samples=10, max_buffer_size=200, negative_swap_percentage=50,
              max_same_video_negs=6)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    assert a.stat(1) == 0
    a.prefetch_start(depth=2, threads=3)
    assert a.stat(2) == 1
    for _ in range(10):
        i1, l1, y1 = a.next(want_last=True, want_label=True)
        i2, l2, y2 = o.next()
        assert np.array_equal(i1, i2) and np.array_equal(l1, l2) and np.array_equal(y1, y2)
    a.close()
    total = int(ds.n_shots.sum())
    kw = dict(batch_size=32, context_size=3, num_negative_samples=8, max_buffer_size=int(total * 0.8),
              negative_swap_percentage=90)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    for _ in range(60):
        assert np.array_equal(a.next(), o.next()[0])
    assert a.stat(0) > 0, "the restart path was not exercised"
    a.prefetch_start(depth=2, threads=2)
    for _ in range(30):
        assert np.array_equal(a.next(), o.next()[0])
    a.close()


@pytest.mark.parametrize("width", ["avx2", "avx512"])
@pytest.mark.parametrize("case", ["long_videos_always_swap", "crowded_buffer", "odd_slot_count", "buffer_equals_slots", "benchmark_shape"])
def test_both_vector_widths_of_the_walk_and_the_slot_draw_are_the_oracle_stream(oracle, monkeypatch, width, case):
    """sampler.cc has two forms of the swap-in walk and of the negative-slot draw: 256-bit (any x86-64-v3 host) and 512-bit
    (chosen at creation when the host has AVX-512 F/BW/DQ/VL/VBMI2; VV_SAMPLER_AVX512=0 forces the first).  Both must be the
    oracle's stream, bit for bit, serial and pipelined -- on the shapes where they differ most: videos of more than 64 and more
    than 128 shots with a 99 % swap-in (nearly every test taken: up to 64 taken tests per chunk, position words up to word 127 of
    the chunk), a buffer that holds most of a small dataset, so that swap-ins keep evicting later shots of the video being walked (restarts), slot counts that are not
    a multiple of the vector length, a buffer of exactly the slot count (the last divisor of the draw is 1)."""
    monkeypatch.setenv("VV_SAMPLER_AVX512", "1" if width == "avx512" else "0")
    cfg = {
        "long_videos_always_swap": (dict(seed=3, n_videos=60, lo=50, span=120), dict(batch_size=24, context_size=5, num_negative_samples=20, max_buffer_size=3000, negative_swap_percentage=99)),
        "crowded_buffer": (dict(seed=4, n_videos=40, lo=6, span=90), dict(batch_size=16, context_size=3, num_negative_samples=7, max_buffer_size=1500, negative_swap_percentage=90)),
        "odd_slot_count": (dict(seed=5, n_videos=200, lo=5, span=70), dict(batch_size=32, context_size=5, num_negative_samples=37, max_buffer_size=900, negative_swap_percentage=35)),
        "buffer_equals_slots": (dict(seed=6, n_videos=80, lo=8, span=40), dict(batch_size=8, context_size=5, num_negative_samples=33, max_buffer_size=33, negative_swap_percentage=50)),
        "benchmark_shape": (dict(seed=1701, n_videos=512), dict(batch_size=128, context_size=5, num_negative_samples=50, max_buffer_size=5000, negative_swap_percentage=50)),
    }[case]
    ds = SyntheticVideos(**cfg[0])
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **cfg[1])
    if width == "avx512" and a.stat(7) != 1:
        a.close()
        pytest.skip("this host has no AVX-512 VBMI2")
    assert a.stat(7) == (1 if width == "avx512" else 0)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **cfg[1])
    for _ in range(12):
        assert np.array_equal(a.next(), o.next()[0])
    a.prefetch_start(depth=3, threads=4)
    for _ in range(24):
        assert np.array_equal(a.next(), o.next()[0])
    if case == "crowded_buffer":
        assert a.stat(0) > 0, "the restart path was not exercised"
    a.close()


@pytest.mark.parametrize("width", ["avx2", "avx512"])
def test_stream_blocks_of_a_dozen_items_and_a_stalling_consumer(oracle, monkeypatch, width):
    """Four stage threads, the stream generated by its own thread in blocks that change hands with the walk: blocks of a dozen
    items (VV_SAMPLER_BLOCK; far fewer than the 32 items after which the walk publishes its progress), a consumer that pauses so
    that every stage runs into a full ring, a stop while everything is blocked -- the stream stays the oracle's."""
    import time
    monkeypatch.setenv("VV_SAMPLER_AVX512", "1" if width == "avx512" else "0")
    monkeypatch.setenv("VV_SAMPLER_BLOCK", "1024")
    ds = SyntheticVideos(seed=21, n_videos=150, lo=4, span=60)
    kw = dict(batch_size=40, context_size=5, num_negative_samples=16, max_buffer_size=600, negative_swap_percentage=60)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    a.prefetch_start(depth=2, threads=4)
    for k in range(50):
        if k % 10 == 3:
            time.sleep(0.05)
        assert np.array_equal(a.next(), o.next()[0])
    time.sleep(0.05)                     # ring full, every stage waiting: the stop must still get through
    a.prefetch_stop()
    a.close()
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    a.prefetch_start(depth=4, threads=4)
    for _ in range(30):
        assert np.array_equal(a.next(), o.next()[0])
    a.close()


def test_two_samplers_never_pin_their_stage_threads_to_the_same_cores(oracle):
    """Each stage thread of a pipeline gets a core of its own when the caller's group of eight has four to spare -- claimed through
    a lock per CPU, so that a second sampler (another rank of an unbound job, a second data layer) takes OTHER cores or, when there
    are none left, shares the group as before; the claim ends with the pipeline.  Streams unaffected."""
    ds = SyntheticVideos(seed=31, n_videos=100, lo=6, span=30)
    kw = dict(batch_size=32, context_size=5, num_negative_samples=8, max_buffer_size=300, negative_swap_percentage=50)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    b = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    oa = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    ob = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    a.prefetch_start(depth=2, threads=4)
    b.prefetch_start(depth=2, threads=4)
    held = sorted(os.readlink("/proc/self/fd/" + f) for f in os.listdir("/proc/self/fd")
                  if os.path.exists("/proc/self/fd/" + f) and "vv_sampler_cpu_" in os.readlink("/proc/self/fd/" + f))
    assert a.stat(9) in (0, 4) and b.stat(9) in (0, 4)
    assert len(held) == a.stat(9) + b.stat(9) and len(set(held)) == len(held)       # distinct CPUs
    for _ in range(6):
        assert np.array_equal(a.next(), oa.next()[0]) and np.array_equal(b.next(), ob.next()[0])
    a.close(); b.close()
    assert not [f for f in os.listdir("/proc/self/fd") if os.path.exists("/proc/self/fd/" + f) and "vv_sampler_cpu_" in os.readlink("/proc/self/fd/" + f)]


def _ring_consumer(name, consumer, world, n_batches, q):
    import videovector_amd as vv2
    r = vv2.BatchRing.attach(name, timeout_s=30.0)
    b = r.batch_size // world
    out = [r.next(consumer, consumer * b, b, want_label=True, timeout_s=30.0) for _ in range(n_batches)]
    q.put((consumer, [o[0].copy() for o in out], [o[1].copy() for o in out]))
    r.close()


def test_one_sampler_per_node_through_shared_memory(oracle):
    """ONE logical sampler draws the global batch; every data-parallel rank takes its items.  Here the
    producer process publishes the ring in POSIX shared memory and two other processes (ranks 1 and 2 of 3) attach."""
    import multiprocessing as mp
    ds = SyntheticVideos(seed=3, n_videos=200, lo=8, span=40)
    world, Bg = 3, 96
    kw = dict(batch_size=Bg, context_size=5, num_negative_samples=12, max_buffer_size=500, negative_swap_percentage=50)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    name = "vv_test_ring_%d" % os.getpid()
    a.prefetch_start(depth=2, threads=3, shm_name=name, consumers=world)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_batches = 12                                      # > depth: the ring wraps and every consumer's release matters
    procs = [ctx.Process(target=_ring_consumer, args=(name, c, world, n_batches, q)) for c in (1, 2)]
    for pr in procs:
        pr.start()
    mine = a.ring()
    b = Bg // world
    got = {0: ([], [])}
    for _ in range(n_batches):
        i, y = mine.next(0, 0, b, want_label=True, timeout_s=30.0)
        got[0][0].append(i.copy()); got[0][1].append(y.copy())
    for _ in range(2):
        c, ii, yy = q.get(timeout=60)
        got[c] = (ii, yy)
    for pr in procs:
        pr.join(30)
        assert pr.exitcode == 0
    for k in range(n_batches):
        ref_i, _, ref_y = o.next()
        for c in range(world):
            assert np.array_equal(got[c][0][k], ref_i[c * b:(c + 1) * b])
            assert np.array_equal(got[c][1][k], ref_y[c * b:(c + 1) * b])
    a.close()


@pytest.mark.parametrize("threads", [0, 1, 3])
@pytest.mark.parametrize("dist", [False, True])
def test_pairwise_context_bit_exact_vs_oracle(oracle, threads, dist):
    """CONTEXT_PAIRWISE (...data_layer.cpp:396-422): two random frames per record in draw order, context_size forced to 2,
    one-shot records skipped; with output_shot_distance the label is the clamped frame distance.  Serial, one prefetch
    thread and the three-stage pipeline give the oracle's stream."""
    ds = SyntheticVideos(seed=3, n_videos=200, lo=1, span=40)
    kw = dict(batch_size=32, context_size=9, num_negative_samples=10, max_buffer_size=300, negative_swap_percentage=50,
              context_type="PAIRWISE", output_shot_distance=dist, max_shot_distance=6.5)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    assert a.CN == 12 and a.stat(1) == 1
    if threads:
        a.prefetch_start(depth=3, threads=threads)
    for _ in range(30):
        i1, l1, y1 = a.next(want_last=True, want_label=True)
        i2, l2, y2 = o.next()
        assert np.array_equal(i1, i2) and np.array_equal(l1, l2) and np.array_equal(y1, y2)
    if dist:
        assert y1.max() <= 6 and np.array_equal(y1, np.minimum(np.abs(i1[:, 0] - i1[:, 1]), 6))
    a.close()
    # no negatives at all: the pair only
    kw.update(num_negative_samples=0, max_buffer_size=0)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    for _ in range(5):
        assert np.array_equal(a.next(), o.next()[0])
    a.close()


def test_negative_dataset_bit_exact_vs_oracle(oracle):
    """negative_dataset (...data_layer.cpp:105-151, 253-286, 325-341): the buffer starts as every shot of the negative
    dataset's first records (no draw, main cursor untouched); swap-ins then bring the main dataset's shots in.  Same
    video ids in both datasets share keys, as the reference's "vid:shot" strings do."""
    ds = SyntheticVideos(seed=9, n_videos=80, lo=3, span=20)
    total = int(ds.n_shots.sum())
    nns = np.array([6, 9, 5, 12, 8, 10]); nvid = np.array([3, 1000, 1001, 7, 1002, 1003])
    nrb = total + np.concatenate([[0], np.cumsum(nns[:-1])])
    kw = dict(batch_size=16, context_size=5, num_negative_samples=6, max_buffer_size=int(nns[:5].sum()),
              negative_swap_percentage=60)
    a = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, negatives=(nvid, nns, nrb), **kw)
    o = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, negatives=(nvid, nns, nrb), **kw)
    assert a.stat(1) == 0                        # keys are not rows any more: the general path
    for k in range(25):
        i1, l1, y1 = a.next(want_last=True, want_label=True)
        i2, l2, y2 = o.next()
        assert np.array_equal(i1, i2) and np.array_equal(l1, l2) and np.array_equal(y1, y2)
        if k == 0:
            assert i1[0, 5:].min() >= total      # before any swap-in the negatives are the negative dataset's rows
    a.close()
    for mb in (int(nns[:5].sum()) - 1, int(nns.sum()) + 1):       # overshoot / never full
        with pytest.raises(vv.VVError):
            vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, negatives=(nvid, nns, nrb), **dict(kw, max_buffer_size=mb))
        with pytest.raises(ValueError):
            oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, negatives=(nvid, nns, nrb), **dict(kw, max_buffer_size=mb))
