#!/bin/bash
cd $GRAFT_REPO_ROOT
{ timeout 300 tools/lab/gather_burst_lab 1; timeout 300 tools/lab/gather_burst_lab 0; } > gpurun_out/r05_s24_gather_burst.txt 2>&1
cat gpurun_out/r05_s24_gather_burst.txt
