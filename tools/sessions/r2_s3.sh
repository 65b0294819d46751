#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
VV_TRACE_HOST=0.5 VV_BENCH_DIAG=1 python3 bench.py --no-cpu-baseline --no-extra-legs --steps 100 --warmup 5 > gpurun_out/r2_s3_a.json 2> gpurun_out/r2_s3_a.err
grep "vv host" gpurun_out/r2_s3_a.err | head -40
