#!/bin/bash
# Round-5 session 11: the update in the weight-gradient epilogue is bimodal between processes (wgrad 101-109 us in most runs, 182-188 us in
# 2 of 7): does the slow mode go with where the parameter buffers were placed?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for i in 1 2 3 4 5 6 7 8 9 10; do
  VV_DEBUG_PTRS=1 timeout 300 python bench.py --workload shipped --steps 120 --warmup 10 --no-cpu-baseline > $O/s11.json 2> $O/s11.err
  echo "run $i: $(python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/s11.json') if x.startswith('{')][-1]); print(round(d['ms_per_step'],4), round(d['kernels_ms']['wgrad_gemm'],4), round(d['kernels_ms']['reduce_sgd'],4))") $(grep 'vv ptrs' $O/s11.err | head -1)"
done
