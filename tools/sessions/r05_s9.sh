#!/bin/bash
# Round-5 session 9: k_score_fwd's dependence on bytes / footprint (lab hack), clean weight-gradient stamps, shipped configuration again.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
bash tools/lab/score_bytes_probe.sh > $O/r05_s9_score_bytes.txt 2>&1
timeout 600 tools/lab/wgrad_ts_lab 20650 40 2 0 > $O/r05_s9_wgrad_stamps_cold.txt 2>&1
timeout 600 tools/lab/wgrad_ts_lab 20650 40 1 1 > $O/r05_s9_wgrad_stamps_ic.txt 2>&1
for i in 1 2 3; do
  for nh in 0 1; do
    VV_BENCH_NO_HINT=$nh timeout 600 python bench.py --workload shipped --steps 200 --warmup 20 --no-cpu-baseline > $O/r05_s9_bench_shipped_nohint${nh}_$i.json 2>> $O/r05_s9_bench.err
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_s9_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 5), {k: round(v, 4) for k, v in d.get("kernels_ms", {}).items()})
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/r05_s9_score_bytes.txt
grep -v "^check" $O/r05_s9_wgrad_stamps_cold.txt | cut -c1-420
grep -v "^check" $O/r05_s9_wgrad_stamps_ic.txt | cut -c1-420
