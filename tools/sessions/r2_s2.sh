#!/bin/bash
# round 2, session 2: new bench line (end-to-end with the prefetch ring), sampler rates with / without pinning
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 tools/samp_time.py gpurun_out/r2_s2_samp_pin.json > gpurun_out/r2_s2_samp_pin.txt 2>&1
VV_SAMPLER_PIN=0 python3 tools/samp_time.py gpurun_out/r2_s2_samp_nopin.json > gpurun_out/r2_s2_samp_nopin.txt 2>&1
python3 bench.py --no-cpu-baseline > gpurun_out/r2_s2_bench.json 2> gpurun_out/r2_s2_bench.err
tail -5 gpurun_out/r2_s2_bench.err
timeout 900 python3 -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r2_s2_dist.txt 2>&1; tail -15 gpurun_out/r2_s2_dist.txt
echo PIN; cat gpurun_out/r2_s2_samp_pin.txt; echo NOPIN; cat gpurun_out/r2_s2_samp_nopin.txt
