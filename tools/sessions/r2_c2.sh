#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r2_c2.txt 2>&1; tail -15 gpurun_out/r2_c2.txt
