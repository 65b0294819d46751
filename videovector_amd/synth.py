"""Synthetic videovec dataset: the stand-in for the reference's VideoShots LMDB
(`source:` of VIDEO_SAMPLED_SHOTS_DATA, projects/videovec_embedding/mednet_embedding_train.prototxt:11).

Everything is a pure function of (seed, index) built from integer hashing only, so the host
(numpy), the CPU oracle and the HIP kernel `vv_table_synth` regenerate bit-identical inputs:

  frames of video v      n_v = 16 + (mix64(seed, v) mod 49)                      in [16, 64]
  feature (row r, col j) x   = max(0, n0+n1+n2+n3 - 30) / 8, n_i = 4-bit nibbles of
                               mix64(seed, 2^40 + r*F + j)                        in {0, 1/8 .. 3.75}

About half of the values are zero (like post-ReLU fc7 activations) and every value is exactly
representable in bf16, f16 and f32, so "identical inputs" holds for every precision mode.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
FEATURE_KEY_OFFSET = 1 << 40


def mix64(seed, x):
    """splitmix64 finaliser of (seed * GOLD + x); vectorised over x (uint64)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * _GOLD + np.asarray(x, dtype=np.uint64)
        z = z + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def video_lengths(seed, n_videos, lo=16, span=49):
    return (lo + (mix64(seed, np.arange(n_videos, dtype=np.uint64)) % np.uint64(span))).astype(np.int32)


def feature_rows(seed, rows, F):
    """fp32 features of the given table rows: array [len(rows)][F]."""
    rows = np.asarray(rows, dtype=np.uint64).reshape(-1, 1)
    j = np.arange(F, dtype=np.uint64).reshape(1, -1)
    h = mix64(seed, np.uint64(FEATURE_KEY_OFFSET) + rows * np.uint64(F) + j)
    s = (h & np.uint64(15)) + ((h >> np.uint64(4)) & np.uint64(15)) + \
        ((h >> np.uint64(8)) & np.uint64(15)) + ((h >> np.uint64(12)) & np.uint64(15))
    return (np.maximum(s.astype(np.int32) - 30, 0).astype(np.float32)) * np.float32(0.125)


class SyntheticVideos:
    """V videos; record v has video_id v, shot_ids 0..n_v-1 and owns table rows
    [row_base[v], row_base[v] + n_v)."""

    def __init__(self, seed=1701, n_videos=2048, lo=16, span=49):
        self.seed, self.n_videos = seed, n_videos
        self.n_shots = video_lengths(seed, n_videos, lo, span)
        self.video_id = np.arange(n_videos, dtype=np.int32)
        self.row_base = np.concatenate([[0], np.cumsum(self.n_shots[:-1])]).astype(np.int64)
        self.n_rows = int(self.n_shots.sum())

    def table(self, F, rows=None):
        return feature_rows(self.seed, np.arange(self.n_rows) if rows is None else rows, F)


def init_weights(seed, D, F, std=1e-3):
    """fc7 fillers of the project prototxt (gaussian std 1e-3 weights, constant 0 bias;
    mednet_embedding_train.prototxt:199-208).  Values come from numpy's PCG64, not from the
    reference's boost RNG: weights are an INPUT that tests hand to both sides explicitly."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((D, F)) * std).astype(np.float32), np.zeros(D, np.float32)


def synthetic_windows(ds, windows=673, context=4, wseed=7):
    """TEST-phase records of the `synthetic-windows://` source (caffe_facade VideoDataset::Open): window w
    is `context` consecutive frames of video mix64(wseed, w) % V starting at
    mix64(wseed, 2^32 + w) % (n - context + 1).  Returns (rows [windows][context], video_ids [windows])."""
    w = np.arange(windows, dtype=np.uint64)
    v = (mix64(wseed, w) % np.uint64(ds.n_videos)).astype(np.int64)
    st = (mix64(wseed, np.uint64(1 << 32) + w) % (ds.n_shots[v] - context + 1).astype(np.uint64)).astype(np.int64)
    rows = (ds.row_base[v] + st)[:, None] + np.arange(context)[None, :]
    return rows.astype(np.int32), ds.video_id[v].astype(np.int32)
