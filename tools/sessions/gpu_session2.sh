#!/bin/bash
# GPU session: full parity suite, bench, rocprofv3 kernel trace + PMC passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/s2_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/s2_pytest.log
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/s2_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/s2_bench.log
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s2_prof_trace.log 2>&1
echo "trace exit $?" >> gpurun_out/s2_prof_trace.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_fetch -o pf -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/s2_prof_fetch.log 2>&1
echo "fetch exit $?" >> gpurun_out/s2_prof_fetch.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_write -o pw -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/s2_prof_write.log 2>&1
echo "write exit $?" >> gpurun_out/s2_prof_write.log
find gpurun_out/prof -name "*.csv" | head -20
# keep only the small summaries (stats + per-kernel counter sums)
python3 tools/summarize_prof.py gpurun_out/prof > gpurun_out/s2_prof_summary.txt 2>&1
find gpurun_out/prof -name "*kernel_trace.csv" -size +3M -delete
find gpurun_out/prof -name "*counter_collection.csv" -size +3M -delete
tail -4 gpurun_out/s2_pytest.log; tail -2 gpurun_out/s2_bench.log | cut -c1-600; cat gpurun_out/s2_prof_summary.txt | head -60
