#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_dedup.py tests/test_gpu_parity.py tests/test_gpu_cfg5.py -x -q -s > gpurun_out/r2_o3.txt 2>&1; grep -E "DEDUP heavy|passed|failed|Error|^E " gpurun_out/r2_o3.txt | cut -c1-300 | tail
