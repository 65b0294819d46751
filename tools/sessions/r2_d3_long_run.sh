#!/bin/bash
# per-step times of a long run (does the step time drift?)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VV_BENCH_DIAG=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 6000 --warmup 20 > gpurun_out/long.json 2> gpurun_out/long.err
python3 - <<'PY'
import re, numpy as np
t = open('gpurun_out/long.err').read()
m = re.search(r"main-leg step ms: (.*)", t)
a = np.array([float(x) for x in m.group(1).split()])
print("n", len(a), "mean", a.mean())
for i in range(0, len(a), 500):
    w = a[i:i+500]; print(i, "median %.4f mean %.4f p95 %.4f max %.3f" % (np.median(w), w.mean(), np.percentile(w, 95), w.max()))
PY
rocm-smi --showclocks --showpower 2>/dev/null | head -20
