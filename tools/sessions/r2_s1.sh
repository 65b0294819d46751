#!/bin/bash
# round 2, session 1: state of the tree at the start of the round + host sampler rates on the GPU box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket|MHz" > gpurun_out/r2_host.txt
python3 tools/samp_time.py gpurun_out/r2_s1_samp.json > gpurun_out/r2_s1_samp.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2_s1_gputests.txt 2>&1
tail -3 gpurun_out/r2_s1_gputests.txt
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r2_s1_bench.json 2> gpurun_out/r2_s1_bench.err
cat gpurun_out/r2_s1_samp.txt
