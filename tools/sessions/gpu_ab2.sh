#!/bin/bash
# A/B over an env var: usage gpu_ab2.sh VAR "v1 v2 ..."
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do for v in $2; do
  env $1=$v timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dense-leg > gpurun_out/ab_bench.log 2>&1
  echo "$1=$v: $(python3 -c "
import json
l=[x for x in open('gpurun_out/ab_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms'], 'loss', d['final_loss'])")"
done; done
