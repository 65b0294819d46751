#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/lab/samp_handover.sh > gpurun_out/r05_s22_handover.txt 2>&1
cat gpurun_out/r05_s22_handover.txt
