#!/bin/bash
# Round-5 session 28: full GPU test suite + smoke + the bench line on the round's final code.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/r05_s28_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s28_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_s28_smoke.log 2>&1; echo "smoke exit $?" >> $O/r05_s28_smoke.log
timeout 900 python bench.py > $O/r05_s28_bench.json 2> $O/r05_s28_bench.err; echo "bench exit $?" >> $O/r05_s28_bench.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_s28_bench_driver_args.json 2>> $O/r05_s28_bench.err
tail -4 $O/r05_s28_pytest.log; tail -2 $O/r05_s28_smoke.log
python3 -c "
import json
for f in ('r05_s28_bench.json','r05_s28_bench_driver_args.json'):
    d=json.loads([x for x in open('gpurun_out/'+f) if x.startswith('{')][-1]); print(f, round(d['ms_per_step'],4), round(d['value']/1e6,1), d['roofline']['frac'], d.get('kernels_ms'), d['cpu_baseline']['value'])"
