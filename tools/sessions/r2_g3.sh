#!/bin/bash
# ablation of the phase-staggered kernels (dense execution; results are wrong by construction when VV_ABLATE != 0)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for ab in 0 1 2 4 8 9 3 6 7; do
  VV_ABLATE=$ab VV_GEMM_VARIANT=5 python3 bench.py --no-cpu-baseline --no-extra-legs --dedup off --steps 40 --warmup 5 > gpurun_out/r2_g3_$ab.json 2> gpurun_out/r2_g3_$ab.err
  python3 -c "
import json;d=json.load(open('gpurun_out/r2_g3_$ab.json'));k=d['kernels_ms'];print('ablate $ab: fwd %.4f wgrad %.4f' % (k['fwd_gemm'], k['wgrad_gemm']))"
done
