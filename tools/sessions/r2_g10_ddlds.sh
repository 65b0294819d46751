#!/bin/bash
# grouping kernels asking for N KB of (unused) LDS so that they cannot share a CU with a forward-GEMM workgroup (128 of 160 KB)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
for k in 0 33 40 60; do
  VV_DEDUP_LDS_KB=$k timeout 300 python3 bench.py --steps 4000 --warmup 300 --no-cpu-baseline --no-extra-legs | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lds $k KB: ms_per_step %.4f' % d['ms_per_step'], d['kernels_ms'], 'frac %.3f' % d['roofline']['frac'])"
done
done
