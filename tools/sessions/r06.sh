#!/bin/bash
# Round-6 GPU sessions: ONE script, named tasks.  `gpurun -- bash tools/sessions/r06.sh <tag> <task> [<task> ...]`; every task writes
# gpurun_out/r06_<tag>_<task>*.  What each session ran and what it showed: tools/sessions/r06_log.md.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
TAG=$1; shift
LAB=$PWD/videovector_amd/lib/libvideovec_lab.so

summ() {   # one line per bench JSON: ms per step, kernels, box
  python3 - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        b = d.get("box") or {}
        print(f.split("/")[-1], "ms %.5f" % d["ms_per_step"], "box %.0f TF %.0f MHz %.2f TB/s at_ref %.5f" % (b.get("gemm_tflops", 0), b.get("gemm_clock_mhz", 0), b.get("copy_tbs", 0), b.get("ms_per_step_at_ref", 0)),
              {k: round(v, 4) for k, v in d.get("kernels_ms", {}).items()})
    except Exception as e:
        print(f, "ERR", e)
PY
}

for task in "$@"; do
  echo "=== $task"
  case $task in
    lds)        # what an LDS fragment read costs beside the partner wave's MFMAs (tools/lab/lds_issue_lab.hip)
      timeout 300 tools/lab/lds_issue_lab > $O/r06_${TAG}_lds_issue.txt 2>&1; cat $O/r06_${TAG}_lds_issue.txt ;;
    driver)     # the driver's command line, twice
      for i in 1 2; do timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_${TAG}_bench_driver_args_$i.json 2>> $O/r06_${TAG}_bench.err; done
      summ $O/r06_${TAG}_bench_driver_args_*.json ;;
    default)    # the defaults (200 steps), all legs
      timeout 900 python bench.py > $O/r06_${TAG}_bench.json 2>> $O/r06_${TAG}_bench.err; summ $O/r06_${TAG}_bench.json ;;
    quick)      # the step alone, 400 steps, three times
      for i in 1 2 3; do timeout 600 python bench.py --no-extra-legs --no-cpu-baseline --steps 400 > $O/r06_${TAG}_quick_$i.json 2>> $O/r06_${TAG}_bench.err; done
      summ $O/r06_${TAG}_quick_*.json ;;
    mq31)       # lab build: 192-row forward tiles (216 workgroups) against 176-row tiles (236), alternating
      for i in 1 2 3; do
        for mq in 0 31; do
          VV_LIB=$LAB VV_PH_MQ=$mq timeout 600 python bench.py --no-extra-legs --no-cpu-baseline --steps 400 > $O/r06_${TAG}_mq${mq}_$i.json 2>> $O/r06_${TAG}_bench.err
        done
      done
      summ $O/r06_${TAG}_mq*.json ;;
    ablab)      # the same A/B on the LAB library (ablated / timing-only instantiations: VV_LAB_* switches)
      IFS='|' read -ra ARMS <<< "${AB_ENVS:-|}"
      for i in 1 2 3; do
        n=0
        for arm in "${ARMS[@]}"; do
          env VV_LIB=$LAB $arm timeout 600 python bench.py --no-extra-legs --no-cpu-baseline --steps 400 $AB_ARGS > $O/r06_${TAG}_ablab${n}_$i.json 2>> $O/r06_${TAG}_bench.err
          n=$((n + 1))
        done
      done
      echo "arms (lab library): ${AB_ENVS}"; summ $O/r06_${TAG}_ablab*.json ;;
    ab)         # A/B of environment settings, alternating: AB_ENVS="A=1 B=2|A=2" (| separates the arms), AB_ARGS extra bench.py arguments
      IFS='|' read -ra ARMS <<< "${AB_ENVS:-|}"
      for i in 1 2 3; do
        n=0
        for arm in "${ARMS[@]}"; do
          env $arm timeout 600 python bench.py --no-extra-legs --no-cpu-baseline --steps 400 $AB_ARGS > $O/r06_${TAG}_ab${n}_$i.json 2>> $O/r06_${TAG}_bench.err
          n=$((n + 1))
        done
      done
      echo "arms: ${AB_ENVS}"; summ $O/r06_${TAG}_ab*.json ;;
    gputests)   # the GPU parity suite (TESTS = a -k expression or file list; default: everything marked gpu)
      timeout 3000 python -m pytest ${TESTS:-tests} ${NOX:--x} -q -m gpu > $O/r06_${TAG}_gputests.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/r06_${TAG}_gputests.txt | tail -40 ;;
    tests_s)    # selected GPU tests with their printed figures: TESTS = files, KEXPR = a -k expression
      timeout 3000 python -m pytest $TESTS -x -q -s -m gpu -k "${KEXPR:-test}" > $O/r06_${TAG}_tests_s.txt 2>&1; grep -v '^$' $O/r06_${TAG}_tests_s.txt | grep -i 'FULLBATCH\|CFG5\|passed\|failed\|error' | tail -40 ;;
    ab5)        # the same A/B on the per-GPU work of BASELINE configs[4] (batch 4096, 200 negatives, 4096 -> 1024)
      IFS='|' read -ra ARMS <<< "${AB_ENVS:-|}"
      for i in 1 2; do
        n=0
        for arm in "${ARMS[@]}"; do
          env $arm timeout 900 python bench.py --workload cfg5 --no-extra-legs --no-cpu-baseline --steps 30 --warmup 5 > $O/r06_${TAG}_ab5_${n}_$i.json 2>> $O/r06_${TAG}_bench.err
          n=$((n + 1))
        done
      done
      echo "arms: ${AB_ENVS}"; summ $O/r06_${TAG}_ab5_*.json ;;
    diag)       # the driver's 20-step region, step by step (VV_BENCH_DIAG: one HIP event per step + the host time of every call)
      VV_BENCH_DIAG=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/r06_${TAG}_diag.json 2> $O/r06_${TAG}_diag.err; grep -E "main-leg|host ms" $O/r06_${TAG}_diag.err | cut -c1-600; summ $O/r06_${TAG}_diag.json ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_${TAG}_smoke.txt 2>&1; tail -5 $O/r06_${TAG}_smoke.txt ;;
    cfg5)
      timeout 900 python bench.py --workload cfg5 --steps 30 --warmup 5 --no-cpu-baseline > $O/r06_${TAG}_bench_cfg5.json 2>> $O/r06_${TAG}_bench.err; summ $O/r06_${TAG}_bench_cfg5.json ;;
    shipped)
      timeout 900 python bench.py --workload shipped --steps 200 --warmup 20 --no-cpu-baseline > $O/r06_${TAG}_bench_shipped.json 2>> $O/r06_${TAG}_bench.err; summ $O/r06_${TAG}_bench_shipped.json ;;
    final)      # the closing session: bench legs, rocprofv3 kernel trace + HBM PMC passes (dedup on / off), SQ counters, sampler rates, facade rate
      P0=$O/r06_${TAG}
      timeout 1200 python bench.py > ${P0}_bench.json 2> ${P0}_bench.err; echo "bench exit $?" >> ${P0}_bench.err
      timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > ${P0}_bench_driver_args.json 2>> ${P0}_bench.err
      timeout 600 python bench.py --dedup off --no-cpu-baseline --no-extra-legs > ${P0}_bench_dense.json 2>> ${P0}_bench.err
      timeout 600 python bench.py --prec bf16 --no-cpu-baseline --no-extra-legs > ${P0}_bench_bf16.json 2>> ${P0}_bench.err
      timeout 600 python bench.py --workload cfg5 --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 > ${P0}_bench_cfg5.json 2>> ${P0}_bench.err
      timeout 600 python bench.py --workload shipped --steps 200 --warmup 20 --no-cpu-baseline > ${P0}_bench_shipped.json 2>> ${P0}_bench.err
      rm -rf $O/prof_r06_${TAG} && mkdir -p $O/prof_r06_${TAG}
      export VV_BENCH_NO_BOX=1       # (the profiled runs: without the box probe, whose GEMM launches carry the product kernel's name)
      for mode in on off; do
        P=$O/prof_r06_${TAG}/dedup_${mode}
        timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o kt -- python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-extra-legs --dedup $mode > ${P0}_prof_trace_${mode}.log 2>&1
        timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pmc_fetch -o pf -- python3 bench.py --steps 6 --warmup 2 --settle-ms 0 --no-cpu-baseline --no-extra-legs --dedup $mode > ${P0}_prof_fetch_${mode}.log 2>&1
        timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pmc_write -o pw -- python3 bench.py --steps 6 --warmup 2 --settle-ms 0 --no-cpu-baseline --no-extra-legs --dedup $mode > ${P0}_prof_write_${mode}.log 2>&1
        python3 tools/summarize_prof.py $P > ${P0}_kernel_trace_and_pmc_summary_dedup_${mode}.txt 2>&1
        cp $(find $P/trace -name "*kernel_stats.csv" | head -1) ${P0}_kernel_stats_dedup_${mode}.csv
      done
      python3 tools/make_pmc_json.py $O/prof_r06_${TAG} > ${P0}_pmc.json 2> ${P0}_pmc.err
      i=0
      for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
                 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
                 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
        i=$((i+1))
        timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/prof_r06_${TAG}/sq_on_$i -o sq -- python3 bench.py --steps 6 --warmup 2 --settle-ms 0 --no-cpu-baseline --no-extra-legs > ${P0}_sq_on_$i.log 2>&1
      done
      PROFDIR=$O/prof_r06_${TAG} python3 - > ${P0}_sq_counters_summary.txt <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.environ["PROFDIR"] + "/sq_on_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        for k in ("k_fwd_gemm", "k_wgrad_gemm", "k_score_fwd", "k_seg_bwd", "k_reduce_sgd"):
            if k in n:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== dedup on (the default step)")
for k, cs in acc.items():
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
        mf = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
        bz = sum(cs["SQ_BUSY_CYCLES"]) / len(cs["SQ_BUSY_CYCLES"])
        print("   kernel cycles %.0f, matrix pipe busy per SIMD = %.3f" % (bz / 32, mf / (bz / 32 * 1024)))
PY
      unset VV_BENCH_NO_BOX
      python3 tools/samp_rates.py 10 ${P0}_sampler_rates.json > ${P0}_sampler_rates.txt 2>&1
      bash tools/facade_rate.sh > ${P0}_facade_rate.txt 2>&1
      (lscpu | head -25; nproc; echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; uptime) > ${P0}_host_cpu.txt 2>&1
      find $O/prof_r06_${TAG} -name "*kernel_trace.csv" -size +1M -delete
      find $O/prof_r06_${TAG} -name "*counter_collection.csv" -size +1M -delete
      summ ${P0}_bench.json ${P0}_bench_driver_args.json ${P0}_bench_dense.json ${P0}_bench_bf16.json ${P0}_bench_cfg5.json ${P0}_bench_shipped.json
      grep -E "k_fwd|k_wgrad|k_score|k_reduce|k_sgd|k_seg|k_dd" ${P0}_kernel_trace_and_pmc_summary_dedup_on.txt | head -24
      cut -c1-300 ${P0}_sq_counters_summary.txt ;;
    *) echo "unknown task $task" ;;
  esac
done
