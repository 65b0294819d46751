#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 0 1; do for ab in 0 1 2 3 6 7; do
  VV_GEMM_VARIANT=$v VV_ABLATE=$ab timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/abl.log 2>&1
  echo "variant $v ablate $ab: $(python3 -c "
import json
l=[x for x in open('gpurun_out/abl.log') if x.startswith('{')]
d=json.loads(l[-1]); print(d['kernels_ms']['fwd_gemm'], d['kernels_ms']['wgrad_gemm'])")"
done; done 2>&1 | tee gpurun_out/s5_ablate.txt
