#!/bin/bash
# Round-5 session 12: wgrad_update opt-in: tests, the shipped line with its extra leg.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_update.py tests/test_gpu_shipped.py tests/test_gpu_facade.py -q -x > $O/r05_s12_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s12_pytest.log
timeout 600 python bench.py --workload shipped --steps 200 --warmup 20 > $O/r05_bench_shipped.json 2> $O/r05_s12_bench.err
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/r05_bench_shipped.json') if x.startswith('{')][-1]); print(round(d['ms_per_step'],4), d['kernels_ms']); print(d['update_in_wgrad_execution'])"
tail -4 $O/r05_s12_pytest.log
