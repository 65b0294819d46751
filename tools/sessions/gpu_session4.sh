#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s -x > gpurun_out/s4_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/s4_pytest.log
for v in 1 0 1 0; do
  VV_GEMM_VARIANT=$v timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/s4_bench_v$v.log 2>&1
  echo "variant $v: $(python3 -c "
import json,sys
l=[x for x in open('gpurun_out/s4_bench_v$v.log') if x.startswith('{')]
d=json.loads(l[-1]); print(d['ms_per_step'], d['kernels_ms'], d['final_loss'])")"
done
grep -E "PARITY|FULLSIZE|SGD |passed|failed" gpurun_out/s4_pytest.log | cut -c1-200
