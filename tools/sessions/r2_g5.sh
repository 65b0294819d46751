#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2_g5_gputests.txt 2>&1; tail -4 gpurun_out/r2_g5_gputests.txt
python3 bench.py --no-cpu-baseline > gpurun_out/r2_g5_bench.json 2> gpurun_out/r2_g5_bench.err
python3 -c "
import json;d=json.load(open('gpurun_out/r2_g5_bench.json'));print(d['value'], d['ms_per_step'], d['gpu_path_only']['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], 'dense', d['dense_execution']['ms_per_step'], d['dense_execution']['kernels_ms'], 'bf16', d['bf16_execution']['ms_per_step'])"
python3 bench.py --no-cpu-baseline --workload cfg5 --prec bf16 --no-extra-legs --steps 30 --warmup 5 > gpurun_out/r2_g5_cfg5.json 2> gpurun_out/r2_g5_cfg5.err
python3 -c "
import json;d=json.load(open('gpurun_out/r2_g5_cfg5.json'));print('cfg5', d['value'], d['ms_per_step'], d['kernels_ms'])"
