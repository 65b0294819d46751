"""Sampler restatement: C oracle (glibc clone) vs an independent pure-Python restatement driven by
the REAL glibc rand() -- bit-exact indices -- plus structural invariants of the reference layer
(src/caffe/layers/video_sampled_shots_data_layer.cpp).  The reference has no test for this layer
(SURVEY.md section 4): these checks are all that pins it."""
import numpy as np
import pytest

from tests.pyref import PySampler
from videovector_amd.synth import SyntheticVideos


def _mk(oracle, ds, **kw):
    return oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)


@pytest.mark.parametrize("B,C,Nn,mb,swap,max_same", [
    (32, 5, 2, 100, 50, 0),      # BASELINE config 1 shape
    (16, 5, 10, 200, 50, 6),     # shipped prototxt values (quirk Q1 active)
    (8, 3, 4, 64, 99, 2),
    (8, 7, 3, 50, 0, 0),         # swap 0: buffer never changes
    (4, 5, 0, 0, 0, 0),          # no negatives at all
])
def test_c_oracle_equals_python_restatement_on_real_glibc(oracle, B, C, Nn, mb, swap, max_same):
    ds = SyntheticVideos(seed=7, n_videos=60, lo=2, span=20)      # includes videos shorter than C
    s = _mk(oracle, ds, batch_size=B, context_size=C, num_negative_samples=Nn,
            max_buffer_size=mb, negative_swap_percentage=swap, max_same_video_negs=max_same)
    p = PySampler(ds.video_id, ds.n_shots, ds.row_base, B, C, Nn, mb, swap, max_same)
    if Nn > 0:
        assert s.buffer_rows().tolist() == p.buf_row
    for _ in range(6):
        i1, l1, y1 = s.next()
        i2, l2, y2 = p.next()
        assert np.array_equal(i1, i2) and np.array_equal(l1, l2) and np.array_equal(y1, y2)
        assert s.rand_calls() == p.calls and s.cursor() == p.cursor
    if Nn > 0:
        assert s.buffer_rows().tolist() == p.buf_row
        assert s.buffer_ids().tolist() == p.buffer_ids


@pytest.mark.parametrize("ctype", ["PAST", "PAST_CONTINUOUS", "PAST_CONTINUOUS_FIXED"])
@pytest.mark.parametrize("B,C,Nn,mb,swap,max_same", [(16, 5, 10, 200, 50, 6), (8, 4, 4, 64, 99, 2), (8, 2, 3, 50, 0, 3),
                                                    (4, 6, 0, 0, 0, 0)])
def test_past_context_modes_equal_python_restatement_on_real_glibc(oracle, ctype, B, C, Nn, mb, swap, max_same):
    # video_sampled_shots_data_layer.cpp:510-757; even context sizes are legal here (only WINDOW needs an odd one)
    ds = SyntheticVideos(seed=9, n_videos=60, lo=2, span=24)
    s = _mk(oracle, ds, batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=mb,
            negative_swap_percentage=swap, max_same_video_negs=max_same, context_type=ctype)
    p = PySampler(ds.video_id, ds.n_shots, ds.row_base, B, C, Nn, mb, swap, max_same, context_type=ctype)
    for _ in range(6):
        i1, l1, y1 = s.next()
        i2, l2, y2 = p.next()
        assert np.array_equal(i1, i2) and np.array_equal(l1, l2) and np.array_equal(y1, y2)
        assert s.rand_calls() == p.calls and s.cursor() == p.cursor


@pytest.mark.parametrize("ctype", ["PAST", "PAST_CONTINUOUS", "PAST_CONTINUOUS_FIXED"])
def test_past_context_structure(oracle, ctype):
    ds = SyntheticVideos(seed=3, n_videos=80)
    C, Nn, B, max_same = 4, 6, 64, 3
    s = _mk(oracle, ds, batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=300,
            negative_swap_percentage=50, max_same_video_negs=max_same, context_type=ctype)
    row2vid = np.repeat(ds.video_id, ds.n_shots)
    vid_end = {int(v): int(ds.row_base[i] + ds.n_shots[i]) for i, v in enumerate(ds.video_id)}
    for _ in range(3):
        idx, last, label = s.next()
        for b in range(B):
            t, ctx = idx[b, 0], idx[b, 1:C]
            assert np.all(row2vid[idx[b, :C]] == label[b])
            assert np.all(np.diff(ctx) > 0) and ctx[-1] < t            # the context precedes the target, in time order
            if ctype != "PAST":
                steps = np.diff(np.append(ctx, t))
                assert np.all(steps == steps[0])                       # equally spaced frames
            if ctype == "PAST_CONTINUOUS_FIXED":
                # largest stride minus one, window ending at the video's last frame (:688-691)
                n = int(ds.n_shots[list(ds.video_id).index(label[b])])
                msl = (n - C) // (C - 1)
                assert steps[0] == (msl - 1 if msl >= 1 else 0) + 1 and t == vid_end[int(label[b])] - 1
            same = [c for c in range(C, C + Nn) if idx[b, c] != last[b, c]]   # Q1 slots = same-video negatives
            assert len(same) <= max_same
            for c in same:
                # PAST: frames before the SECOND chosen frame (rand_perm_ids[1], :570); continuous: before the window
                bound = np.append(ctx, t)[1] if ctype == "PAST" else ctx[0]
                assert row2vid[idx[b, c]] == label[b] and idx[b, c] < bound


def test_rand_skip_initial_cursor(oracle):
    # rand_skip (…data_layer.cpp:156-180): the cursor starts `skip` records in (wrapping), before the buffer is filled
    ds = SyntheticVideos(seed=7, n_videos=60, lo=2, span=20)
    for skip in (0, 7, 59, 60, 131):
        s = _mk(oracle, ds, batch_size=8, context_size=5, num_negative_samples=4, max_buffer_size=64,
                negative_swap_percentage=50, initial_cursor=skip)
        p = PySampler(ds.video_id, ds.n_shots, ds.row_base, 8, 5, 4, 64, 50, initial_cursor=skip)
        assert s.buffer_rows().tolist() == p.buf_row
        for _ in range(3):
            a, b = s.next(), p.next()
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
    a = _mk(oracle, ds, batch_size=8, context_size=5, num_negative_samples=4, max_buffer_size=64,
            negative_swap_percentage=50, initial_cursor=60).next()
    b = _mk(oracle, ds, batch_size=8, context_size=5, num_negative_samples=4, max_buffer_size=64,
            negative_swap_percentage=50).next()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))          # a full lap is no skip


def test_window_structure(oracle):
    ds = SyntheticVideos(seed=3, n_videos=80)
    C, Nn, B = 5, 6, 64
    s = _mk(oracle, ds, batch_size=B, context_size=C, num_negative_samples=Nn,
            max_buffer_size=300, negative_swap_percentage=50)
    row2vid = np.repeat(ds.video_id, ds.n_shots)
    for _ in range(4):
        buf_before = set(s.buffer_rows().tolist())
        idx, last, label = s.next()
        assert np.array_equal(idx, last)                       # max_same_video_negs == 0
        # target + context come from the labelled video, distinct frames, and the target is the
        # temporal middle: context channels are sorted and split evenly around it (:437-453)
        for b in range(B):
            assert np.all(row2vid[idx[b, :C]] == label[b])
            t, ctx = idx[b, 0], idx[b, 1:C]
            assert np.all(np.diff(ctx) > 0)
            assert (ctx < t).sum() == C // 2 and (ctx > t).sum() == C // 2
            assert len(set(idx[b, C:].tolist())) == Nn          # partial Fisher-Yates: distinct slots
        # item 0's negatives were drawn before any swap of this batch
        assert set(idx[0, C:].tolist()) <= buf_before


def test_short_videos_are_skipped_but_advance_cursor(oracle):
    # :387,:427,:848 -- records with fewer than C shots add nothing and draw nothing
    vid = np.arange(6, dtype=np.int32)
    ns = np.array([1, 4, 9, 2, 12, 3], np.int32)
    rb = np.concatenate([[0], np.cumsum(ns[:-1])]).astype(np.int64)
    s = oracle.Sampler(vid, ns, rb, batch_size=6, context_size=5, num_negative_samples=0,
                       max_buffer_size=0, negative_swap_percentage=0)
    _, _, label = s.next()
    assert label.tolist() == [2, 4, 2, 4, 2, 4]


def test_q1_same_video_negatives_keep_previous_last_feature(oracle):
    ds = SyntheticVideos(seed=11, n_videos=40)
    C, Nn, B = 5, 8, 8
    s = _mk(oracle, ds, batch_size=B, context_size=C, num_negative_samples=Nn,
            max_buffer_size=100, negative_swap_percentage=50, max_same_video_negs=6)
    idx0, last0, _ = s.next()
    row2vid = np.repeat(ds.video_id, ds.n_shots)
    # first batch: same-video negative slots were never written before -> last feature is zero (-1)
    same = row2vid[idx0[:, C:]] == row2vid[idx0[:, :1]]
    assert same[:, :6].all() and np.all(last0[:, C:C + 6] == -1)
    assert np.array_equal(idx0[:, C + 6:], last0[:, C + 6:])
    idx1, last1, _ = s.next()
    # second batch: those slots still hold nothing for the last feature (only Q1 copies hit them)
    assert np.all(last1[:, C:C + 6] == -1) and not np.array_equal(idx0, idx1)
    # the same-video negatives lie outside the target's immediate sampled neighbours (:489-490)
    for b in range(B):
        lo, hi = idx1[b, C // 2], idx1[b, C // 2 + 1]
        assert np.all((idx1[b, C:C + 6] < lo) | (idx1[b, C:C + 6] > hi))


def test_setup_check_fails_like_reference(oracle):
    # :344 CHECK_EQ(num_negatives_added, max_buffer_size): more slots than unique shots
    ds = SyntheticVideos(seed=1, n_videos=3, lo=2, span=1)     # 6 shots in total
    with pytest.raises(ValueError):
        _mk(oracle, ds, batch_size=2, context_size=3, num_negative_samples=2, max_buffer_size=7,
            negative_swap_percentage=10)
    with pytest.raises(ValueError):                             # :79-80 swap percentage range
        _mk(oracle, ds, batch_size=2, context_size=3, num_negative_samples=2, max_buffer_size=4,
            negative_swap_percentage=100)


@pytest.mark.parametrize("dist", [False, True])
@pytest.mark.parametrize("B,Nn,mb,swap", [(16, 10, 200, 50), (8, 0, 0, 0), (8, 3, 40, 99)])
def test_pairwise_equals_python_restatement_on_real_glibc(oracle, dist, B, Nn, mb, swap):
    # …data_layer.cpp:396-422, 200-201: two random frames per record, context_size forced to 2
    rng = np.random.default_rng(11)
    ns = rng.integers(1, 14, 60)
    vid = np.arange(100, 160)
    rb = np.concatenate([[0], np.cumsum(ns[:-1])])
    py = PySampler(vid, ns, rb, B, 7, Nn, mb, swap, context_type="PAIRWISE", output_shot_distance=dist,
                   max_shot_distance=4.5)
    oc = oracle.Sampler(vid, ns, rb, batch_size=B, context_size=7, num_negative_samples=Nn, max_buffer_size=mb,
                        negative_swap_percentage=swap, context_type="PAIRWISE", output_shot_distance=dist,
                        max_shot_distance=4.5)
    for _ in range(6):
        a, b = py.next(), oc.next()
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        idx, _, label = b
        assert idx.shape[1] == 2 + Nn
        assert np.all(idx[:, 0] != idx[:, 1])
        if dist:
            d = np.abs(idx[:, 0] - idx[:, 1])
            assert np.array_equal(label, np.where(d >= 4.5, 4, d))
    assert py.calls == oc.rand_calls()


def test_negative_dataset_fills_the_buffer_in_order(oracle):
    # …data_layer.cpp:253-286, 325-341: no rand(), the main cursor untouched, every shot of the first records
    ns = np.full(30, 8); vid = np.arange(30); rb = np.arange(30) * 8
    nns = np.array([5, 7, 4, 9]); nvid = np.arange(1000, 1004); nrb = 240 + np.concatenate([[0], np.cumsum(nns[:-1])])
    kw = dict(batch_size=4, context_size=3, num_negative_samples=2, max_buffer_size=16, negative_swap_percentage=50)
    oc = oracle.Sampler(vid, ns, rb, negatives=(nvid, nns, nrb), **kw)
    assert oc.rand_calls() == 0 and oc.cursor() == 0
    assert np.array_equal(oc.buffer_rows(), 240 + np.arange(16))
    py = PySampler(vid, ns, rb, 4, 3, 2, 16, 50, negatives=(nvid, nns, nrb))
    for _ in range(5):
        for x, y in zip(py.next(), oc.next()):
            assert np.array_equal(x, y)
    # a record that overshoots the buffer is the reference's out-of-bounds write (:325-343) -> refused
    with pytest.raises(ValueError):
        oracle.Sampler(vid, ns, rb, negatives=(nvid, nns, nrb), **dict(kw, max_buffer_size=15))
    # never full: CHECK_EQ fails (:348)
    with pytest.raises(ValueError):
        oracle.Sampler(vid, ns, rb, negatives=(nvid, nns, nrb), **dict(kw, max_buffer_size=26))
