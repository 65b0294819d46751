#!/bin/bash
# round-end style run: full GPU test suite, smoke, bench, rocprofv3 kernel trace + PMC passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/${TAG}_smoke.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/${TAG}_bench.log
VV_PREC=bf16 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16.log 2>&1
rm -rf gpurun_out/prof_${TAG} && mkdir -p gpurun_out/prof_${TAG}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}/trace -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_prof_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_${TAG}/pmc_fetch -o pf -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_${TAG}/pmc_write -o pw -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_prof_write.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof_${TAG} > gpurun_out/${TAG}_prof_summary.txt 2>&1
find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" -size +2M -delete
find gpurun_out/prof_${TAG} -name "*counter_collection.csv" -size +2M -delete
tail -3 gpurun_out/${TAG}_pytest.log; tail -2 gpurun_out/${TAG}_smoke.log; grep '^{' gpurun_out/${TAG}_bench.log | cut -c1-400; grep '^{' gpurun_out/${TAG}_bench_bf16.log | cut -c1-200; grep -E "k_fwd|k_wgrad|k_score|k_reduce|k_sgd" gpurun_out/${TAG}_prof_summary.txt | head -24
