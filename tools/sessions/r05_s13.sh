#!/bin/bash
# Round-5 session 13: scalar wave index in the score kernels (row-index loads become s_load: no vmcnt(0) between the row requests),
# context rows requested first.  Parity, bench, per-kernel times, stamps.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dedup.py -q -x > $O/r05_s13_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s13_pytest.log
tail -3 $O/r05_s13_pytest.log
timeout 600 python bench.py --steps 300 --warmup 30 > $O/r05_s13_bench.json 2> $O/r05_s13_bench.err
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/r05_s13_bench.json') if x.startswith('{')][-1]); print(round(d['ms_per_step'],4), d.get('kernels_ms'))"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r05_s13_prof -o s13 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1)
f=$(find $O/r05_s13_prof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-150
VV_LIB=$PWD/videovector_amd/lib/libvideovec_lab.so timeout 300 python tools/lab/score_ts.py 2>&1 | tail -15
