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
      timeout 3000 python -m pytest ${TESTS:-tests} -x -q -m gpu > $O/r06_${TAG}_gputests.txt 2>&1; tail -15 $O/r06_${TAG}_gputests.txt ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_${TAG}_smoke.txt 2>&1; tail -5 $O/r06_${TAG}_smoke.txt ;;
    cfg5)
      timeout 900 python bench.py --workload cfg5 --steps 30 --warmup 5 --no-cpu-baseline > $O/r06_${TAG}_bench_cfg5.json 2>> $O/r06_${TAG}_bench.err; summ $O/r06_${TAG}_bench_cfg5.json ;;
    shipped)
      timeout 900 python bench.py --workload shipped --steps 200 --warmup 20 --no-cpu-baseline > $O/r06_${TAG}_bench_shipped.json 2>> $O/r06_${TAG}_bench.err; summ $O/r06_${TAG}_bench_shipped.json ;;
    prof)       # rocprofv3 kernel trace + stats of the default step (cd /tmp first: the profiler writes beside the cwd)
      R=$PWD; ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/r06_${TAG}_prof -o trace -- python3 $R/bench.py --no-extra-legs --no-cpu-baseline --steps 600 > $R/$O/r06_${TAG}_prof_bench.json 2> $R/$O/r06_${TAG}_prof.err )
      find $O/r06_${TAG}_prof -name '*kernel_stats.csv' | head -1 | xargs -r head -12 ;;
    pmc)        # HBM traffic per kernel: separate passes (FETCH_SIZE / WRITE_SIZE do not fit one)
      R=$PWD
      for ctr in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/r06_${TAG}_pmc_$ctr -o pmc -- python3 $R/bench.py --no-extra-legs --no-cpu-baseline --steps 60 --warmup 5 --settle-ms 0 > /dev/null 2> $R/$O/r06_${TAG}_pmc_$ctr.err )
      done
      python3 tools/make_pmc_json.py $O/r06_${TAG}_pmc_FETCH_SIZE $O/r06_${TAG}_pmc_WRITE_SIZE > $O/r06_${TAG}_pmc.json 2> $O/r06_${TAG}_pmc_json.err; head -c 1500 $O/r06_${TAG}_pmc.json ;;
    *) echo "unknown task $task" ;;
  esac
done
