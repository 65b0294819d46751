#!/bin/bash
# which stamp releases the host to queue the next grouping: 0 = the forward GEMM's (at its start), 1 = the score kernel's
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3 4; do
for g in 0 1; do
  VV_DEDUP_GATE=$g timeout 300 python3 bench.py --steps 4000 --warmup 300 --no-cpu-baseline --no-extra-legs > gpurun_out/gate_$g.json 2> gpurun_out/gate_$g.err
  python3 - $g <<'PY'
import json, sys
g = sys.argv[1]
d = json.loads(open("gpurun_out/gate_%s.json" % g).read().strip().splitlines()[-1])
print("gate", g, "ms_per_step %.4f" % d["ms_per_step"], d.get("kernels_ms"), "frac %.3f" % d["roofline"]["frac"])
PY
done
done
