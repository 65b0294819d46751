#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 500 python3 tools/lab/samp_tune.py > gpurun_out/r05_s19_tune.txt 2>&1
cat gpurun_out/r05_s19_tune.txt
