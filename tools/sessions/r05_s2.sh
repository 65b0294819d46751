#!/bin/bash
# Round-5 session 2: the forward GEMM's epilogue with the bias loaded once (was: 24 serial load + store round trips): parity tests,
# bench, and the stamps again.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=tools/lab/fwd_dr_lab
timeout 600 $L 20650 40 3 "ph_lead,ph_plain,ph_ts_lead,ph_ts_plain,ph_ts_lead_hotA" 0 > $O/r05_s2_stamps_cold.txt 2>&1
timeout 600 $L 20650 40 2 "ph_lead,ph_plain,ph_ts_lead" 1 > $O/r05_s2_stamps_ic.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dedup.py tests/test_gpu_shipped.py tests/test_gpu_cfg5.py tests/test_gpu_fused_update.py -q -x > $O/r05_s2_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s2_pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/r05_s2_bench.json 2> $O/r05_s2_bench.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/r05_s2_bench_driver.json 2>> $O/r05_s2_bench.err
timeout 600 python bench.py --workload shipped --steps 200 --warmup 20 --no-cpu-baseline > $O/r05_s2_bench_shipped.json 2>> $O/r05_s2_bench.err
grep -v "^check" $O/r05_s2_stamps_cold.txt | tail -24
grep -v "^check" $O/r05_s2_stamps_ic.txt | tail -12
tail -5 $O/r05_s2_pytest.log
cut -c1-1500 $O/r05_s2_bench.json; cut -c1-600 $O/r05_s2_bench_driver.json; cut -c1-900 $O/r05_s2_bench_shipped.json; tail -3 $O/r05_s2_bench.err
