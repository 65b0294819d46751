#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_comm.py -x -q -s > gpurun_out/r2_c1.txt 2>&1; grep -E "COMM|passed|failed|Error|^E " gpurun_out/r2_c1.txt | cut -c1-300 | tail -20
