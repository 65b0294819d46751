#!/bin/bash
# Round-5 session 6: the update in the weight-gradient GEMM's epilogue (vv_update_hint), the pipelined score kernel: tests and A/Bs; the
# host's CPU allowance.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(echo "nproc: $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"; grep Cpus_allowed_list /proc/self/status; uptime; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node" ) > $O/r05_s6_host.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_fused_update.py tests/test_gpu_fullsize.py tests/test_gpu_segbwd.py tests/test_gpu_dedup.py tests/test_gpu_parity.py tests/test_gpu_shipped.py tests/test_gpu_facade.py tests/test_gpu_fuzz.py -q -x --durations=5 > $O/r05_s6_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s6_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_s6_smoke.log 2>&1; echo "smoke exit $?" >> $O/r05_s6_smoke.log
for i in 1 2; do
  for pipe in 1 0; do
    VV_SCORE_PIPE=$pipe timeout 600 python bench.py --no-cpu-baseline --no-extra-legs > $O/r05_s6_bench_pipe${pipe}_$i.json 2>> $O/r05_s6_bench.err
  done
  for nh in 0 1; do
    VV_BENCH_NO_HINT=$nh timeout 600 python bench.py --workload shipped --steps 200 --warmup 20 --no-cpu-baseline > $O/r05_s6_bench_shipped_nohint${nh}_$i.json 2>> $O/r05_s6_bench.err
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_s6_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 5), {k: round(v, 4) for k, v in d.get("kernels_ms", {}).items()}, "loss", d.get("final_loss"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/r05_s6_host.txt
tail -12 $O/r05_s6_pytest.log
tail -3 $O/r05_s6_smoke.log
tail -5 $O/r05_s6_bench.err
