#!/bin/bash
# Round-5 session 23: forward GEMM with two phases per barrier pair (VV_FWD_MERGE=1): parity, then A/B in the step.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
VV_FWD_MERGE=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dedup.py tests/test_gpu_ops.py -q -x > $O/r05_s23_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s23_pytest.log
tail -3 $O/r05_s23_pytest.log
for i in 1 2 3; do
for m in 0 1; do
VV_FWD_MERGE=$m timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra-legs > $O/r05_s23_bench_m$m.json 2> $O/r05_s23_bench.err
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/r05_s23_bench_m$m.json') if x.startswith('{')][-1]); print('merge $m:', round(d['ms_per_step'],4), d.get('kernels_ms'), d['final_loss'])"
done
done
for m in 0 1; do
VV_FWD_MERGE=$m timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --dedup off > $O/r05_s23_bench_dense_m$m.json 2>> $O/r05_s23_bench.err
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/r05_s23_bench_dense_m$m.json') if x.startswith('{')][-1]); print('dense merge $m:', round(d['ms_per_step'],4), d.get('kernels_ms'), d['final_loss'])"
done
