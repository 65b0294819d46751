#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_facade.py tests/test_gpu_lmdb.py tests/test_gpu_testbranch.py -x -q -s --durations=6 > gpurun_out/r2_c3.txt 2>&1; grep -E "FACADE-DP|passed|failed|Error|^E |s call|s setup" gpurun_out/r2_c3.txt | cut -c1-300 | tail -20
