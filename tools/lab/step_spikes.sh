#!/bin/bash
# Where the occasional slow steps of a long run come from: 20 000 end-to-end steps with per-step HIP events (VV_BENCH_DIAG=1) under a few
# host arrangements; prints median / mean and the number and total excess of steps above 0.4 ms.
cd $GRAFT_REPO_ROOT
run() {
  tag="$1"; shift
  env "$@" VV_BENCH_DIAG=1 timeout 600 python3 bench.py --no-cpu-baseline --no-extra-legs --steps 20000 --warmup 20 $EXTRA > /tmp/sp.json 2> /tmp/sp.err
  python3 - "$tag" <<'PY'
import sys, json, numpy as np
tag = sys.argv[1]
d = json.loads(open("/tmp/sp.json").read().strip().splitlines()[-1])
x = np.array([float(t) for t in [l for l in open("/tmp/sp.err") if l.startswith("main-leg step ms:")][0].split(":")[1].split()])
big = x > 0.4
print("%-34s ms_per_step %.4f  median %.4f mean %.4f  steps > 0.4 ms: %4d (excess %.1f ms of %.0f)  max %.2f" % (tag, d["ms_per_step"], np.median(x), x.mean(), big.sum(), (x[big] - np.median(x)).sum(), x.sum(), x.max()))
PY
}
for rep in 1 2; do
EXTRA="" run "default (4 stage threads, pinned)" A=1
EXTRA="" run "VV_SAMPLER_PIN=0" VV_SAMPLER_PIN=0
EXTRA="--sampler-threads 2" run "2 stage threads" A=1
EXTRA="--cpu-bind off" run "--cpu-bind off" A=1
EXTRA="--prefetch-depth 512" run "prefetch depth 512" A=1
done
