#!/bin/bash
# Round-5 session 7: the update in the weight-gradient GEMM's epilogue (fixed: no static LDS beside the 160 KiB ring): full GPU suite, smoke,
# the shipped configuration with and without the hint, the default bench with the quota-aware CPU baseline.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --durations=6 > $O/r05_s7_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s7_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_s7_smoke.log 2>&1; echo "smoke exit $?" >> $O/r05_s7_smoke.log
for i in 1 2; do
  for nh in 0 1; do
    VV_BENCH_NO_HINT=$nh timeout 600 python bench.py --workload shipped --steps 200 --warmup 20 --no-cpu-baseline > $O/r05_s7_bench_shipped_nohint${nh}_$i.json 2>> $O/r05_s7_bench.err
  done
done
timeout 900 python bench.py > $O/r05_s7_bench.json 2>> $O/r05_s7_bench.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_s7_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 5), {k: round(v, 4) for k, v in d.get("kernels_ms", {}).items()}, "loss", d.get("final_loss"))
        if "cpu_baseline" in d: print("   cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "cores", "cpu_quota_cpus", "threads_tried_s_per_iteration", "fc7_gemm_gflops") if k in d["cpu_baseline"]})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -12 $O/r05_s7_pytest.log
tail -3 $O/r05_s7_smoke.log
tail -5 $O/r05_s7_bench.err
