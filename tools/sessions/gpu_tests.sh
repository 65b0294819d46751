#!/bin/bash
# usage: gpu_tests.sh [pytest -k expression]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
timeout 1500 python -m pytest tests -m gpu -q -s "${K[@]}" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "PARITY|FULLSIZE|SGD |passed|failed|FAILED|Error|^E  " gpurun_out/pytest_gpu.log | cut -c1-250 | tail -60
