#!/usr/bin/env python3
"""Generates the fixtures in this directory.  Everything here comes from code that is NOT this repository's:
  * libc_rand.json          -- rand() of the real glibc of the build image (ctypes -> libc.so.6), several seeds
  * stdlib_permutations.json -- the real libstdc++'s std::sort / std::random_shuffle driven by the real rand()
                               (oracle/pin/stdlib_pin.cpp, the draw pattern of include/caffe/util/rng.hpp:43-54)
  * sampler_batches.npz     -- triplet index batches of tests/pyref.py:PySampler, a pure-Python restatement of
                               VideoSampledShotsDataLayer that consumes the REAL libc rand() stream
  * retrieval_kat.json      -- the known-answer vector of the reference's own test
                               (src/caffe/test/test_retrieval_stats_layer.cpp:40-84: inputs and expected mAP / hit@k)
The reference itself cannot be built or imported here (C++ with absent dependencies), so these
are the strongest pins available: third-party library behaviour and the reference's own test data.
Run from the repository root:  python tests/golden/make_golden.py
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    libc = ctypes.CDLL("libc.so.6")
    out = {}
    for seed in (1, 2, 1701, 0, 0xFFFFFFFF):
        libc.srand(ctypes.c_uint(seed))
        out[str(seed)] = [libc.rand() for _ in range(256)]
    json.dump(out, open(os.path.join(HERE, "libc_rand.json"), "w"))

    pin = os.path.join(ROOT, "oracle", "pin", "stdlib_pin")
    if not os.path.exists(pin):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    sizes = [5, 6, 7, 16, 33, 64, 5, 9, 100, 2, 1, 3, 40]
    lines = subprocess.check_output([pin, "script", ",".join(map(str, sizes))], text=True).splitlines()
    json.dump({"sizes": sizes, "take": 5, "permutations": [[int(x) for x in l.split()] for l in lines]},
              open(os.path.join(HERE, "stdlib_permutations.json"), "w"))

    from pyref import PySampler
    from videovector_amd.synth import SyntheticVideos
    ds = SyntheticVideos(seed=7, n_videos=60, lo=2, span=20)
    arrays, cases = {}, []
    for name, kw in [("window_q1", dict(B=16, C=5, Nn=10, max_buffer=200, swap=50, max_same=6, context_type="WINDOW")),
                     ("window_cfg1", dict(B=32, C=5, Nn=2, max_buffer=100, swap=50, max_same=0, context_type="WINDOW")),
                     ("past", dict(B=8, C=4, Nn=4, max_buffer=64, swap=99, max_same=2, context_type="PAST")),
                     ("past_continuous", dict(B=8, C=3, Nn=5, max_buffer=64, swap=50, max_same=3, context_type="PAST_CONTINUOUS")),
                     ("past_continuous_fixed", dict(B=8, C=5, Nn=3, max_buffer=50, swap=0, max_same=3,
                                                    context_type="PAST_CONTINUOUS_FIXED"))]:
        p = PySampler(ds.video_id, ds.n_shots, ds.row_base, kw["B"], kw["C"], kw["Nn"], kw["max_buffer"], kw["swap"],
                      kw["max_same"], context_type=kw["context_type"])
        for it in range(3):
            idx, last, label = p.next()
            arrays["%s_idx_%d" % (name, it)] = idx
            arrays["%s_last_%d" % (name, it)] = last
            arrays["%s_label_%d" % (name, it)] = label
        cases.append(dict(name=name, rand_calls=p.calls, cursor=p.cursor, **kw))
    arrays["dataset_seed_nvideos_lo_span"] = np.array([7, 60, 2, 20])
    np.savez_compressed(os.path.join(HERE, "sampler_batches.npz"), **arrays)
    json.dump(cases, open(os.path.join(HERE, "sampler_cases.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
