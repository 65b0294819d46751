#!/bin/bash
# Round-5 session 35: wgrad_update ON by default: full GPU suite, twenty more processes of the shipped workload, the shipped line with its extra leg.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/r05_s35_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s35_pytest.log; tail -3 $O/r05_s35_pytest.log
{
for i in $(seq 21 40); do
timeout 300 python bench.py --workload shipped --steps 150 --warmup 20 --no-cpu-baseline --no-extra-legs 2> /dev/null | python3 -c "
import sys, json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); k=d['kernels_ms']; print('run %2d: step %.4f  wgrad %.4f  reduce_sgd %.4f' % ($i, d['ms_per_step'], k['wgrad_gemm'], k['reduce_sgd']))"
done
} > $O/r05_s35_rotated.txt 2>&1
cat $O/r05_s35_rotated.txt
timeout 600 python bench.py --workload shipped --steps 200 --warmup 20 > $O/r05_s35_bench_shipped.json 2> $O/r05_s35.err
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/r05_s35_bench_shipped.json') if x.startswith('{')][-1]); print('shipped line:', round(d['ms_per_step'],4), d['kernels_ms']); e=d['update_as_own_launch_execution']; print('own launch:', round(e['ms_per_step'],4), e['kernels_ms'], e['final_loss'], d['final_loss'])"
