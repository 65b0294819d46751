"""Pins the oracle's restatement of the third-party algorithms the reference sampler delegates to
-- glibc rand() (never seeded) and libstdc++ std::random_shuffle / std::sort -- against the REAL
libraries of this machine (oracle/pin/stdlib_pin).  Call sites in the reference:
src/caffe/layers/video_sampled_shots_data_layer.cpp:27,29,306,437,482, include/caffe/util/rng.hpp:43-54.
"""
import numpy as np


def test_rand_known_answer(oracle):
    # glibc 2.35 seed-1 stream (SURVEY.md App. B KAT)
    g = oracle.Rand()
    assert [g.next() for _ in range(5)] == [1804289383, 846930886, 1681692777, 1714636915,
                                            1957747793]


def test_rand_matches_real_glibc_100k(oracle):
    ref = np.array(oracle.stdlib_pin("rand", 100000), dtype=np.int64)
    g = oracle.Rand()
    mine = np.array([g.next() for _ in range(100000)], dtype=np.int64)
    assert np.array_equal(ref, mine)


def test_srand_other_seeds(oracle):
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 2, 1701, 0xFFFFFFFF, 0):
        libc.srand(ctypes.c_uint(seed))
        ref = [libc.rand() for _ in range(500)]
        g = oracle.Rand(seed)
        assert ref == [g.next() for _ in range(500)], seed


def test_random_unique_sort_shuffle_match_libstdcxx(oracle):
    sizes = [5, 6, 7, 16, 33, 64, 5, 9, 100, 2, 1, 3, 40]
    ref = oracle.stdlib_pin("script", ",".join(map(str, sizes)))
    g = oracle.Rand()
    for n, line in zip(sizes, ref):
        take = min(n, 5)
        a = g.random_unique(np.arange(n, dtype=np.int32), take)
        a[:take] = np.sort(a[:take])
        a[take:] = g.random_shuffle(a[take:].copy())
        assert a.tolist() == [int(x) for x in line.split()], n
