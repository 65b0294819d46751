#!/bin/bash
# quick per-kernel trace of bench.py (dedup on, no dense leg): prints the stats table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/trace_q && mkdir -p gpurun_out/trace_q
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_q -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense-leg $@ > gpurun_out/trace_q.log 2>&1
f=$(find gpurun_out/trace_q -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-60s calls %5s avg %9.1f ns  total%% %s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
find gpurun_out/trace_q -name "*kernel_trace.csv" -size +2M -delete
