#!/bin/bash
# first GPU session: parity tests, smoke, short bench, kernel-trace profile
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4 > gpurun_out/s1_info.log 2>&1
nproc >> gpurun_out/s1_info.log
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/s1_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/s1_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s1_smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/s1_smoke.log
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/s1_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/s1_bench.log
tail -5 gpurun_out/s1_pytest.log; cat gpurun_out/s1_smoke.log | tail -3; tail -3 gpurun_out/s1_bench.log
