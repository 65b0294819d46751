#!/bin/bash
# phase-staggered forward kernel: parity tests + timing A/B (variant 5 = new forward, old wgrad)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
VV_GEMM_VARIANT=5 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dedup.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r2_g1_tests.txt 2>&1; tail -5 gpurun_out/r2_g1_tests.txt
for v in 0 5; do
  VV_GEMM_VARIANT=$v python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r2_g1_v$v.json 2> gpurun_out/r2_g1_v$v.err
  python3 -c "
import json;d=json.load(open('gpurun_out/r2_g1_v$v.json'));print('variant $v', d['ms_per_step'], d['kernels_ms'], 'dense', d['dense_execution']['kernels_ms'])"
done
for q in 2 3 4; do
  VV_PH_MQ=$q VV_GEMM_VARIANT=5 python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r2_g1_q$q.json 2> gpurun_out/r2_g1_q$q.err
  python3 -c "
import json;d=json.load(open('gpurun_out/r2_g1_q$q.json'));print('MQ $q', d['ms_per_step'], d['kernels_ms']['fwd_gemm'], 'dense', d['dense_execution']['kernels_ms']['fwd_gemm'])"
done
