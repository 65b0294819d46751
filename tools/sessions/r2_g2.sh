#!/bin/bash
# phase-staggered kernels: parity tests + timing A/B (5 = both new, 6 = new wgrad only, 7 = new fwd only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
VV_GEMM_VARIANT=5 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dedup.py tests/test_gpu_fullsize.py tests/test_gpu_cfg5.py -x -q > gpurun_out/r2_g2_tests.txt 2>&1; tail -5 gpurun_out/r2_g2_tests.txt
for v in 0 5; do
  VV_GEMM_VARIANT=$v python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r2_g2_v$v.json 2> gpurun_out/r2_g2_v$v.err
  python3 -c "
import json;d=json.load(open('gpurun_out/r2_g2_v$v.json'));print('variant $v', d['ms_per_step'], d['gpu_path_only']['ms_per_step'], d['kernels_ms'], 'dense', d['dense_execution']['ms_per_step'], d['dense_execution']['kernels_ms'])"
done
