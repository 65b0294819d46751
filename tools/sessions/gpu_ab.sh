#!/bin/bash
# A/B of GEMM variants: usage gpu_ab.sh "v1 v2 ..." (VV_GEMM_VARIANT values), interleaved twice
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
VV_GEMM_VARIANT=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/ab_pytest.log 2>&1
tail -2 gpurun_out/ab_pytest.log
for rep in 1 2; do for v in $1; do
  VV_GEMM_VARIANT=$v timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/ab_bench.log 2>&1
  echo "variant $v: $(python3 -c "
import json
l=[x for x in open('gpurun_out/ab_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms'])")"
done; done
