#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_facade.py -x -q -s -k "test_net" > gpurun_out/r2_o2.txt 2>&1; grep -E "SEQUENTIAL|passed|failed|Error|^E |Check failed" gpurun_out/r2_o2.txt | cut -c1-400 | tail -20
