"""bench.py's roofline.traffic / roofline.step_traffic measured by the invocation itself: the full one-GPU line runs two
`rocprofv3 --pmc` passes of itself as child processes before it touches the GPU.  Where the profiler cannot run, the line must fall back to
the committed profile and say so -- never fail."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_measures_its_own_hbm_traffic():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    roof = d["roofline"]
    assert roof["kernel"] == "fwd_gemm" and roof["traffic"] is not None and roof["traffic_note"]
    print("TRAFFIC", roof["traffic"], roof.get("traffic_committed_profile"), roof["traffic_note"][:160])
    if "measured by this invocation" not in roof["traffic_note"]:
        assert "NOT measured by this run" in roof["traffic_note"] and "live passes:" in roof["traffic_note"]
        pytest.skip("live PMC passes unavailable here: " + roof["traffic_note"][-200:])
    # the forward GEMM of the de-duplicated benchmark step: 194.5 MB algorithmic; the committed passes say 230 MB
    assert 190e6 <= roof["traffic"] <= 280e6
    st = roof["step_traffic"]
    assert set(st["kernels"]) == {"fwd_gemm", "score_loss", "segsum", "wgrad_gemm", "reduce_sgd"}
    assert st["pmc_bytes_per_step"] == sum(st["kernels"].values()) and 1.0 <= st["ratio"] <= 1.5
    assert "measured by this invocation" in st["note"]
