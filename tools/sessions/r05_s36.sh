#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 120 tools/lab/slab_pitch_lab; echo "-- (a new process: new physical placement)"; done > gpurun_out/r05_s36_slab_pitch.txt 2>&1
cat gpurun_out/r05_s36_slab_pitch.txt
