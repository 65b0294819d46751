"""GPU tests of the TEST branch of the project net (BASELINE config 4): the
window-mean embedding (average_for_test -> fc7 -> ReLU -> test_norm) and the within-batch retrieval
statistics, against the oracle and the reference's known-answer test."""
import numpy as np
import pytest

from tests.test_gpu_parity import rel_rows
from videovector_amd.synth import SyntheticVideos, init_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vv():
    import videovector_amd
    return videovector_amd


def test_retrieval_stats_reference_known_answer(vv):
    # src/caffe/test/test_retrieval_stats_layer.cpp:34-39,82-84
    eng = vv.Engine(0, "f16")
    feat = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 0.06], [0.0, 1.0], [1.0, 0.1]], np.float32)
    m, h1, h5 = eng.retrieval_stats(feat, [2, 3, 4, 5, 6], {2: 1, 3: 2, 4: 1, 5: 2, 6: 2})
    assert abs(m - 0.7833333) <= 1e-3 and abs(h1 - 0.60) <= 1e-3 and abs(h5 - 0.32) <= 1e-3


@pytest.mark.parametrize("exclude", [True, False])
def test_retrieval_stats_matches_oracle(vv, oracle, exclude):
    rng = np.random.default_rng(1)
    n, dim = 300, 96
    feat = rng.standard_normal((n, dim)).astype(np.float32)
    feat /= np.linalg.norm(feat, axis=1, keepdims=True)
    vids = rng.integers(0, 60, n).astype(np.int32)
    cls = {int(v): int(v % 7) - (2 if v % 11 == 0 else 0) for v in range(60)}     # some negative classes
    got = vv.Engine(0, "f16").retrieval_stats(feat, vids, cls, exclude)
    ref = oracle.retrieval_stats(feat, vids, cls, exclude)
    assert np.allclose(got, ref, atol=2e-4), (got, ref)


def test_embed_mean_matches_oracle(vv, oracle):
    ds = SyntheticVideos(seed=21, n_videos=30)
    F, D = 512, 128
    table = ds.table(F)
    W, b = init_weights(21, D, F, std=0.01)
    eng = vv.Engine(0, "f16")
    eng.table_set(table); eng.params_set(W, b)
    rng = np.random.default_rng(2)
    rows = rng.integers(0, ds.n_rows, size=(37, 4)).astype(np.int32)
    coeff = np.array([0.25, 0.25, 0.25, 0.25], np.float32)
    got = eng.embed_mean(rows, coeff, relu=True, l2norm=True)
    mean_rows = (table[rows] * coeff[None, :, None]).sum(1).astype(np.float32)
    ref = oracle.embed(mean_rows, None, W, b, relu=True, l2norm=True)
    assert rel_rows(got, ref) <= 1e-3
    assert rel_rows(eng.embed_mean(rows, None, relu=True, l2norm=False),
                    oracle.embed(mean_rows, None, W, b, relu=True, l2norm=False)) <= 1e-3


def test_config4_retrieval_map_vs_cpu_reference(vv, oracle):
    # BASELINE config 4 shapes: fc 4096 -> 4096 on windows of 4 frames, TEST batch 673
    # (mednet_embedding_train.prototxt:30-45,133-177,344-352,673-688): the retrieval statistics computed
    # from the HIP embeddings must match those from the CPU-reference embeddings.
    F, D, n = 4096, 4096, 673
    ds = SyntheticVideos(seed=33, n_videos=200)
    W, b = init_weights(33, D, F, std=0.003)
    eng = vv.Engine(0, "f16")
    eng.table_synth(ds.seed, ds.n_rows, F)
    eng.params_set(W, b)
    rng = np.random.default_rng(3)
    vid = rng.integers(0, ds.n_videos, n).astype(np.int32)
    start = (rng.random(n) * (ds.n_shots[vid] - 4)).astype(np.int64)
    rows = (ds.row_base[vid] + start)[:, None] + np.arange(4)[None, :]
    got = eng.embed_mean(rows.astype(np.int32), None, relu=True, l2norm=True)
    uniq, inv = np.unique(rows.reshape(-1), return_inverse=True)
    t = ds.table(F, uniq)
    mean_rows = t[inv.reshape(n, 4)].mean(1).astype(np.float32)
    ref = oracle.embed(mean_rows, None, W, b, relu=True, l2norm=True)
    e = rel_rows(got, ref)
    cls = {int(v): int(v % 15) + 1 for v in range(ds.n_videos)}
    s_gpu = eng.retrieval_stats(got, vid, cls)
    s_ref = oracle.retrieval_stats(ref, vid, cls)
    print("CONFIG4 emb=%.3e  stats gpu=%s ref=%s" % (e, s_gpu, s_ref))
    assert e <= 1e-3
    assert np.allclose(s_gpu, s_ref, atol=5e-3)
