#!/bin/bash
# (lab) what k_score_fwd's time depends on: the same kernel with rows 1 KiB apart (half the L2 footprint, the same requests) and with only the
# first half of every row loaded (half the bytes and half the requests).  Results of the hacked runs are wrong by design.
cd $GRAFT_REPO_ROOT
export VV_LIB=$PWD/videovector_amd/lib/libvideovec_lab.so
for r in 1 2; do
  for h in 0 1 2; do
    VV_LAB_SCORE_HACK=$h timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs > gpurun_out/sbp.log 2>/dev/null
    echo "hack $h: $(python3 -c "
import json
l=[x for x in open('gpurun_out/sbp.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), {k: d['kernels_ms'][k] for k in ('score_loss','segsum')})")"
  done
done
