#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 500 python3 tools/lab/samp_rates_pinned.py > gpurun_out/r05_s29_pinned.txt 2>&1
cat gpurun_out/r05_s29_pinned.txt
