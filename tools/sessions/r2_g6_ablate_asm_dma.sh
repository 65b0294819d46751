#!/bin/bash
# ablation of the phase-staggered kernels after the LDS-DMA went to inline assembly (dense sizes, f16)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for ab in 0 1 2 4 8 3 6 7; do
  VV_GEMM_VARIANT=5 VV_ABLATE=$ab timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/abl.log 2>&1
  echo "ablate $ab: $(python3 -c "
import json
l=[x for x in open('gpurun_out/abl.log') if x.startswith('{')]
d=json.loads(l[-1]); print('fwd', d['kernels_ms']['fwd_gemm'], 'wgrad', d['kernels_ms']['wgrad_gemm'])")"
done 2>&1 | tee gpurun_out/r2_ablate_asm_dma.txt
