#!/bin/bash
# The direct peer exchange (VV_COMM_PEER) between N processes that share ONE device, against the host-staged shared-memory
# transport and against one process: what the exchange's fixed costs (meeting kernels through host-coherent flags, the two
# one-shot kernels, the stream joins) add when the wire is free.  NOT a scaling measurement: the N ranks time-slice one GPU.
# Run on a GPU box: bash tools/lab/peer_one_device.sh > gpurun_out/peer_one_device.txt
export VV_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
port=29600
one() {  # N comm allreduce
  port=$((port + 2))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus $1 \
    --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --sampler node --comm $2 --allreduce $3 2>/dev/null | grep '^{' | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=%d comm=%-4s allreduce=%-8s  %.4f ms/step  (%.4f ms per rank-step on the shared device)' % ($1, '$2', '$3', d['ms_per_step'], d['ms_per_step'] / $1))"
}
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=1 one process                      %.4f ms/step' % d['ms_per_step'])"
for n in 2 4 8; do
  for ar in sync sharded overlap; do one $n peer $ar; done
  one $n lib sharded
done
