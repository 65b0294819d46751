#!/bin/bash
# Round-5 session 3: weight-gradient prologue (row ids requested together), sibling lead on / off in the step, the full GPU test suite,
# TCC counters of the duplicate-free stream ablations.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for i in 1 2; do
  for lead in 1 0; do
    VV_FWD_LEAD=$lead timeout 600 python bench.py --no-cpu-baseline --no-extra-legs > $O/r05_s3_bench_lead${lead}_$i.json 2>> $O/r05_s3_bench.err
  done
done
timeout 1800 python -m pytest tests -m gpu -q -x > $O/r05_s3_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s3_pytest.log
L=tools/lab/fwd_dr_lab
for v in dr12_onlyA dr12_even_onlyA dr12_split_onlyA; do
  P=$O/r05_s3_pmc_$v
  rm -rf $P
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $P -o p -- $L 20650 6 1 "$v" 0 > $O/r05_s3_pmc_$v.log 2>&1
done
python3 - > $O/r05_s3_pmc_summary.txt <<'PY'
import csv, glob, collections
for v in ("dr12_onlyA", "dr12_even_onlyA", "dr12_split_onlyA"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r05_s3_pmc_%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "k_fwd_gemm_dr" in n:
                acc[n[:120]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, cs in acc.items():
        print(v, n, " ".join("%s=%.4g (n=%d)" % (c, sum(x[-6:]) / len(x[-6:]), len(x)) for c, x in sorted(cs.items())))
PY
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_s3_bench_lead*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 5), {k: round(v, 4) for k, v in d.get("kernels_ms", {}).items()}, d.get("step_ms_stats", {}).get("median"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 $O/r05_s3_pytest.log
cat $O/r05_s3_pmc_summary.txt | cut -c1-400
