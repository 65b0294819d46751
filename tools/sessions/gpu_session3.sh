#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/s3_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/s3_pytest.log
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/s3_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/s3_bench.log
rm -rf gpurun_out/prof3 && mkdir -p gpurun_out/prof3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3/trace -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s3_prof_trace.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof3 > gpurun_out/s3_prof_summary.txt 2>&1
find gpurun_out/prof3 -name "*kernel_trace.csv" -size +3M -delete
grep -E "PARITY|FULLSIZE|SGD |passed|failed" gpurun_out/s3_pytest.log | cut -c1-260; tail -2 gpurun_out/s3_bench.log | cut -c1-300; head -12 gpurun_out/s3_prof_summary.txt
