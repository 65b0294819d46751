#!/bin/bash
# Round-5 closing session: full GPU test suite, smoke, bench legs (default, driver's command line, dense, bf16, cfg5, the
# shipped configuration), rocprofv3 kernel trace + HBM PMC passes (dedup on / off), SQ counters of the GEMMs, sampler rates
# with spread and per-stage waits, the cost of the data-parallel schedules on one rank over real RCCL.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> $O/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke exit $?" >> $O/${TAG}_smoke.log
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench exit $?" >> $O/${TAG}_bench.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_driver_args.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --dedup off --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_dense.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --prec bf16 --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_bf16.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --workload cfg5 --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 > $O/${TAG}_bench_cfg5.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --workload shipped --steps 200 --warmup 20 > $O/${TAG}_bench_shipped.json 2>> $O/${TAG}_bench.err
rm -rf $O/prof_${TAG} && mkdir -p $O/prof_${TAG}
for mode in on off; do
  P=$O/prof_${TAG}/dedup_${mode}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o kt -- python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-extra-legs --dedup $mode > $O/${TAG}_prof_trace_${mode}.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pmc_fetch -o pf -- python3 bench.py --steps 6 --warmup 2 --settle-ms 0 --no-cpu-baseline --no-extra-legs --dedup $mode > $O/${TAG}_prof_fetch_${mode}.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pmc_write -o pw -- python3 bench.py --steps 6 --warmup 2 --settle-ms 0 --no-cpu-baseline --no-extra-legs --dedup $mode > $O/${TAG}_prof_write_${mode}.log 2>&1
  python3 tools/summarize_prof.py $P > $O/${TAG}_kernel_trace_and_pmc_summary_dedup_${mode}.txt 2>&1
  cp $(find $P/trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_dedup_${mode}.csv
done
python3 tools/make_pmc_json.py $O/prof_${TAG} > $O/${TAG}_pmc.json 2> $O/${TAG}_pmc.err
# SQ counters of the two GEMMs and the score / segment kernels
for mode in on off; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    P=$O/prof_${TAG}/sq_${mode}_$i
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $P -o sq -- python3 bench.py --steps 6 --warmup 2 --settle-ms 0 --no-cpu-baseline --no-extra-legs --dedup $mode > $O/${TAG}_sq_${mode}_$i.log 2>&1
  done
done
TAGX=$TAG python3 - > $O/${TAG}_sq_counters_summary.txt <<'PY'
import csv, glob, collections, os
tag = os.environ.get("TAGX", "r05")
for mode in ("on", "off"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/prof_%s/sq_%s_*/**/*counter_collection.csv" % (tag, mode), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0]
            for k in ("k_fwd_gemm", "k_wgrad_gemm", "k_score_fwd", "k_score_loss", "k_seg_bwd"):
                if k in n:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== dedup", mode)
    for k, cs in acc.items():
        print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
            # SQ_BUSY_CYCLES sums the 32 shader engines' busy cycles (8 XCDs x 4), SQ_VALU_MFMA_BUSY_CYCLES the 1024 SIMDs'
            mf = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
            bz = sum(cs["SQ_BUSY_CYCLES"]) / len(cs["SQ_BUSY_CYCLES"])
            print("   kernel cycles %.0f, matrix pipe busy per SIMD = %.3f" % (bz / 32, mf / (bz / 32 * 1024)))
PY
python3 tools/samp_rates.py 10 $O/${TAG}_sampler_rates.json > $O/${TAG}_sampler_rates.txt 2>&1
python3 tools/lab/samp_stages.py >> $O/${TAG}_sampler_rates.txt 2>&1
bash tools/facade_rate.sh > $O/${TAG}_facade_rate.txt 2>&1
(lscpu | head -25; nproc; echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; uptime) > $O/${TAG}_host_cpu.txt 2>&1
find $O/prof_${TAG} -name "*kernel_trace.csv" -size +1M -delete
find $O/prof_${TAG} -name "*counter_collection.csv" -size +1M -delete
tail -3 $O/${TAG}_pytest.log; tail -2 $O/${TAG}_smoke.log; cut -c1-400 $O/${TAG}_bench.json; tail -3 $O/${TAG}_bench.err
grep -E "k_fwd|k_wgrad|k_score|k_reduce|k_sgd|k_seg|k_dd" $O/${TAG}_kernel_trace_and_pmc_summary_dedup_on.txt | head -24
cat $O/${TAG}_sq_counters_summary.txt | cut -c1-400
