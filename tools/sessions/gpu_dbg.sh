#!/bin/bash
# debug helper: run each GPU test file separately, keep full logs under gpurun_out/dbg_*.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in tests/test_gpu_*.py; do
  n=$(basename $f .py)
  timeout 600 python -m pytest $f -q -v -x > gpurun_out/dbg_$n.log 2>&1
  echo "$n exit $?"
done
