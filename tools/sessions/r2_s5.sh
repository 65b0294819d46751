#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q -s > gpurun_out/r2_s5_gputests.txt 2>&1
grep -E "CFG3|passed|failed|FAILED|Error|^E  " gpurun_out/r2_s5_gputests.txt | cut -c1-250 | tail -30
bash tools/facade_rate.sh > gpurun_out/r2_s5_facade_rate.txt 2>&1; cat gpurun_out/r2_s5_facade_rate.txt
