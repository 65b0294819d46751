#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r2_o1.txt 2>&1; tail -25 gpurun_out/r2_o1.txt | cut -c1-250
