#!/bin/bash
# Round-5 session 37: split-K slabs 4 KiB askew: tests, bench (reduce_sgd / wgrad before: 21.8-22.4 / 66.4-67.8 us).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fused_update.py tests/test_gpu_dedup.py tests/test_gpu_segbwd.py tests/test_gpu_cfg5.py tests/test_gpu_ops.py tests/test_gpu_comm.py -q -x > $O/r05_s37_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s37_pytest.log; tail -3 $O/r05_s37_pytest.log
for i in 1 2 3; do
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra-legs 2> /dev/null | python3 -c "
import sys, json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('cfg2:', round(d['ms_per_step'],4), d['kernels_ms'])"
done
timeout 600 python bench.py --workload cfg5 --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs 2> /dev/null | python3 -c "
import sys, json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('cfg5:', round(d['ms_per_step'],4), d['kernels_ms'])"
timeout 600 python bench.py --dedup off --no-cpu-baseline --no-extra-legs 2> /dev/null | python3 -c "
import sys, json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('dense:', round(d['ms_per_step'],4), d['kernels_ms'])"
