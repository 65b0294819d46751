#!/bin/bash
# round-end style run: full GPU test suite, smoke, bench, rocprofv3 kernel trace + PMC passes (both execution modes)
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
timeout 600 python bench.py --dedup off --no-cpu-baseline > gpurun_out/${TAG}_bench_dense.log 2>&1
VV_PREC=bf16 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16.log 2>&1
rm -rf gpurun_out/prof_${TAG} && mkdir -p gpurun_out/prof_${TAG}
for mode in on off; do
  P=gpurun_out/prof_${TAG}/dedup_${mode}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense-leg --dedup $mode > gpurun_out/${TAG}_prof_trace_${mode}.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pmc_fetch -o pf -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dense-leg --dedup $mode > gpurun_out/${TAG}_prof_fetch_${mode}.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pmc_write -o pw -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dense-leg --dedup $mode > gpurun_out/${TAG}_prof_write_${mode}.log 2>&1
  python3 tools/summarize_prof.py $P > gpurun_out/${TAG}_prof_summary_dedup_${mode}.txt 2>&1
done
python3 tools/make_pmc_json.py gpurun_out/prof_${TAG} > gpurun_out/${TAG}_pmc.json 2> gpurun_out/${TAG}_pmc.err
for mode in on off; do cp $(find gpurun_out/prof_${TAG}/dedup_${mode}/trace -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats_dedup_${mode}.csv; done
find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" -size +2M -delete
find gpurun_out/prof_${TAG} -name "*counter_collection.csv" -size +2M -delete
tail -3 gpurun_out/${TAG}_pytest.log; tail -2 gpurun_out/${TAG}_smoke.log; grep '^{' gpurun_out/${TAG}_bench.log | cut -c1-300; grep '^{' gpurun_out/${TAG}_bench_dense.log | cut -c1-200; grep '^{' gpurun_out/${TAG}_bench_bf16.log | cut -c1-200
grep -E "k_fwd|k_wgrad|k_score|k_reduce|k_sgd|k_seg|k_dd" gpurun_out/${TAG}_prof_summary_dedup_on.txt | head -30
