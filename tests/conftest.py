import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture
def fp32_ip2(monkeypatch):
    """For tests that hold two EXECUTIONS of one step (segment-wise against row-writing backward, de-duplicated against dense) to bit-equal or
    last-bit forward values: since round 6 the de-duplicated segment-wise path stores ip2 as f16 (option h16) and the other
    executions as fp32 -- a difference of storage, not of arithmetic.  Engines created under this fixture keep fp32 rows everywhere
    (VV_H16=0 is read when a context is created; subprocesses inherit it).  What the f16 rows cost is asserted on its own:
    tests/test_gpu_h16.py and the whole-batch oracle tests."""
    monkeypatch.setenv("VV_H16", "0")


@pytest.fixture
def fp32_slabs(monkeypatch):
    """For the FREE-RUNNING trajectory comparisons of small, chaotic cases (a 1e-6 relative change of W at step 0 moves W by 5e-3 after four
    steps: tests/test_gpu_parity.py::test_sgd_steps_match_oracle): their tight bounds hold only while the gradient's arithmetic is the fp32
    oracle's up to the operand rounding.  Since round 6 the split-K partial products of dW travel as f16 x a power of two per tile (option
    slab16: 2^-12 more per partial product) -- a perturbation these cases amplify 10-20x.  Engines created under this fixture keep fp32 slabs
    (VV_SLAB16=0, read when a context is created; subprocesses inherit it).  What slab16 costs a single step is asserted by
    tests/test_gpu_h16.py and the whole-batch oracle tests; that a long run under the defaults stays inside the operand rounding's envelope by
    tests/test_gpu_longrun.py; the same trajectories under the defaults, with the bounds a chaotic case allows, by the dedup = 1 variants."""
    monkeypatch.setenv("VV_SLAB16", "0")
