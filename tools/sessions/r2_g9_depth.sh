#!/bin/bash
# prefetch ring depth against the sampler's batch-to-batch variance: 20 000 end-to-end steps each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for d in 8 32 128; do
  timeout 300 python3 bench.py --steps 20000 --warmup 20 --no-cpu-baseline --no-extra-legs --prefetch-depth $d | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $d: ms_per_step %.4f value %.1f M' % (d['ms_per_step'], d['value'] / 1e6))"
done
done
