#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/s7_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/s7_pytest.log
grep -E "passed|failed|FAILED" gpurun_out/s7_pytest.log | tail -8
for cfg in "0 8" "1 8" "1 0" "0 0" "1 7" "1 0"; do
  set -- $cfg
  VV_SCORE_REG=$1 VV_FWD_MI=$2 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/s7_bench.log 2>&1
  echo "score_reg $1 fwd_mi $2: $(python3 -c "
import json
l=[x for x in open('gpurun_out/s7_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms'], round(d['value']/1e6,2))")"
done
