#!/bin/bash
# idle gaps between consecutive kernels of the step's stream: end-to-end (ring) against resident indices
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for src in ring resident; do
  rm -rf gpurun_out/gaps_$src
  VV_BENCH_SOURCE=$src timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps_$src -o kt -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/gaps_$src.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for src in ("ring", "resident"):
    f = glob.glob("gpurun_out/gaps_%s/**/*kernel_trace.csv" % src, recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
    rows.sort()
    # main-leg steps: from a k_fwd_gemm to the next k_fwd_gemm; use the last 40 steps of the first leg (70 steps)
    fw = [i for i, r in enumerate(rows) if "k_fwd_gemm" in r[2]]
    fw = fw[25:65]
    per = []
    gaps = collections.defaultdict(list)
    for a, b in zip(fw[:-1], fw[1:]):
        seg = [r for r in rows[a:b] if "k_dd_" not in r[2]]
        per.append((rows[b][0] - rows[a][0]) / 1e3)
        for x, y in zip(seg[:-1], seg[1:]):
            gaps[x[2][-28:] + " -> " + y[2][-28:]].append((y[0] - x[1]) / 1e3)
        gaps[seg[-1][2][-28:] + " -> next fwd"].append((rows[b][0] - seg[-1][1]) / 1e3)
    print("==", src, "step (fwd to fwd) mean %.1f us" % (sum(per) / len(per)))
    for k, v in gaps.items():
        print("   %-62s gap mean %6.2f us" % (k, sum(v) / len(v)))
PY
find gpurun_out/gaps_* -name "*.csv" -size +1M -delete
