"""The committed fixtures of tests/golden/ (made by tests/golden/make_golden.py from the real glibc, the real
libstdc++, a libc-driven pure-Python restatement and the reference's own test data) against the oracle and the
product's host code.  No GPU needed."""
import json
import os

import numpy as np
import pytest

import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_rand_equals_recorded_glibc_streams(oracle):
    for seed, ref in json.load(open(os.path.join(G, "libc_rand.json"))).items():
        g = oracle.Rand(int(seed))
        assert [g.next() for _ in range(len(ref))] == ref, seed


def test_oracle_partial_shuffle_sort_shuffle_equal_recorded_libstdcxx(oracle):
    d = json.load(open(os.path.join(G, "stdlib_permutations.json")))
    g = oracle.Rand()
    for n, ref in zip(d["sizes"], d["permutations"]):
        take = min(n, d["take"])
        a = g.random_unique(np.arange(n, dtype=np.int32), take)
        a[:take] = np.sort(a[:take])
        a[take:] = g.random_shuffle(a[take:].copy())
        assert a.tolist() == ref, n


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_samplers_reproduce_recorded_batches(oracle, which):
    z = np.load(os.path.join(G, "sampler_batches.npz"))
    seed, nv, lo, span = [int(x) for x in z["dataset_seed_nvideos_lo_span"]]
    ds = SyntheticVideos(seed=seed, n_videos=nv, lo=lo, span=span)
    for case in json.load(open(os.path.join(G, "sampler_cases.json"))):
        kw = dict(batch_size=case["B"], context_size=case["C"], num_negative_samples=case["Nn"],
                  max_buffer_size=case["max_buffer"], negative_swap_percentage=case["swap"],
                  max_same_video_negs=case["max_same"], context_type=case["context_type"])
        s = (oracle.Sampler if which == "oracle" else vv.Sampler)(ds.video_id, ds.n_shots, ds.row_base, **kw)
        for it in range(3):
            got = s.next() if which == "oracle" else s.next(want_last=True, want_label=True)
            for k, a in zip(("idx", "last", "label"), got):
                assert np.array_equal(a, z["%s_%s_%d" % (case["name"], k, it)]), (case["name"], k, it)
        if which == "oracle":
            assert s.rand_calls() == case["rand_calls"] and s.cursor() == case["cursor"]


def test_retrieval_known_answer_of_the_reference(oracle):
    d = json.load(open(os.path.join(G, "retrieval_kat.json")))
    m, h1, h5 = oracle.retrieval_stats(np.array(d["features"], np.float32), d["video_ids"],
                                       {int(k): v for k, v in d["id_to_class"].items()})
    e = d["expected"]
    assert abs(m - e["mean_ap"]) <= e["tolerance"] and abs(h1 - e["hit_at_1"]) <= e["tolerance"] and abs(h5 - e["hit_at_5"]) <= e["tolerance"]
