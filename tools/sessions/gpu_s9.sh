#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "pipelined or sgd" > gpurun_out/s9_pytest.log 2>&1
grep -E "PIPELINED|passed|failed|Error|^E " gpurun_out/s9_pytest.log | head
for m in auto overlap; do
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --allreduce $m > gpurun_out/s9_bench.log 2>&1
  echo "allreduce=$m: $(python3 -c "
import json
l=[x for x in open('gpurun_out/s9_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms'], d['final_loss'], d['roofline']['traffic'], d['config']['allreduce'])" 2>&1 | tail -3)"
done
# torchrun single-rank smoke of the distributed code path (nccl with world 1)
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/s9_torchrun.log 2>&1
echo "torchrun exit $?"; grep '^{' gpurun_out/s9_torchrun.log | cut -c1-160
