#!/bin/bash
# Round-5 session 26: the merged-phase forward GEMM on the shipped configuration (128-row tiles, dropout 0.9) and on cfg 5.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
VV_FWD_MERGE=1 timeout 900 python -m pytest tests/test_gpu_shipped.py tests/test_gpu_cfg5.py tests/test_gpu_fuzz.py -q -x > $O/r05_s26_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s26_pytest.log
tail -3 $O/r05_s26_pytest.log
for i in 1 2 3; do
for m in 0 1; do
VV_FWD_MERGE=$m timeout 600 python bench.py --workload shipped --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs > $O/r05_s26_shipped_m$m.json 2> $O/r05_s26_bench.err
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/r05_s26_shipped_m$m.json') if x.startswith('{')][-1]); print('shipped merge $m:', round(d['ms_per_step'],4), d.get('kernels_ms'), d['final_loss'])"
done
done
for m in 0 1; do
VV_FWD_MERGE=$m timeout 600 python bench.py --workload cfg5 --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/r05_s26_cfg5_m$m.json 2>> $O/r05_s26_bench.err
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/r05_s26_cfg5_m$m.json') if x.startswith('{')][-1]); print('cfg5 merge $m:', round(d['ms_per_step'],4), d.get('kernels_ms'), d['final_loss'])"
done
