#!/bin/bash
# Round-5 session 8: weight-gradient kernel's time stamps; the (lab) pipelined score kernel against the product's; the update in the
# weight-gradient epilogue through LDS in row order (shipped configuration, hint on / off); N > 1 legs under torch.distributed.run; CPU baseline.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 tools/lab/wgrad_ts_lab 20650 40 2 0 > $O/r05_s8_wgrad_stamps_cold.txt 2>&1
timeout 600 tools/lab/wgrad_ts_lab 20650 40 1 1 > $O/r05_s8_wgrad_stamps_ic.txt 2>&1
VV_LIB=$PWD/videovector_amd/lib/libvideovec_lab.so timeout 600 python tools/lab/score_pipe_check.py > $O/r05_s8_score_pipe_check.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fused_update.py tests/test_gpu_shipped.py -q -x > $O/r05_s8_pytest_a.log 2>&1; echo "pytest exit $?" >> $O/r05_s8_pytest_a.log
for i in 1 2; do
  for nh in 0 1; do
    VV_BENCH_NO_HINT=$nh timeout 600 python bench.py --workload shipped --steps 200 --warmup 20 --no-cpu-baseline > $O/r05_s8_bench_shipped_nohint${nh}_$i.json 2>> $O/r05_s8_bench.err
  done
done
timeout 900 python -m pytest "tests/test_gpu_dist.py::test_bench_per_rank_samplers_is_the_default_for_two_ranks" "tests/test_gpu_dist.py::test_bench_bare_command_launches_its_own_ranks" -q -x --durations=4 > $O/r05_s8_pytest_b.log 2>&1; echo "pytest exit $?" >> $O/r05_s8_pytest_b.log
timeout 900 python bench.py --no-extra-legs > $O/r05_s8_bench.json 2>> $O/r05_s8_bench.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_s8_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 5), {k: round(v, 4) for k, v in d.get("kernels_ms", {}).items()}, "loss", d.get("final_loss"))
        if "cpu_baseline" in d: print("   cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "cores", "blas", "cpu_quota_cpus", "threads_tried_s_per_iteration", "fc7_gemm_gflops", "sample") if k in d["cpu_baseline"]})
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v "^check" $O/r05_s8_wgrad_stamps_cold.txt | tail -24 | cut -c1-420
grep -v "^check" $O/r05_s8_wgrad_stamps_ic.txt | tail -12 | cut -c1-420
cat $O/r05_s8_score_pipe_check.txt | tail -12
tail -4 $O/r05_s8_pytest_a.log; tail -8 $O/r05_s8_pytest_b.log
tail -3 $O/r05_s8_bench.err
