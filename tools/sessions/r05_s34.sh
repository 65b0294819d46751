#!/bin/bash
# Round-5 session 34: the update in the weight-gradient epilogue with its row groups rotated per tile: twenty processes (is the slow mode gone?), parity.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_fused_update.py -q -x > $O/r05_s34_pytest.log 2>&1; tail -2 $O/r05_s34_pytest.log
{
for i in $(seq 1 20); do
VV_WGRAD_UPDATE=1 timeout 300 python bench.py --workload shipped --steps 150 --warmup 20 --no-cpu-baseline --no-extra-legs 2> /dev/null | python3 -c "
import sys, json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); k=d['kernels_ms']; print('run %2d: step %.4f  wgrad %.4f  reduce_sgd %.4f' % ($i, d['ms_per_step'], k['wgrad_gemm'], k['reduce_sgd']))"
done
} > $O/r05_s34_rotated.txt 2>&1
cat $O/r05_s34_rotated.txt
