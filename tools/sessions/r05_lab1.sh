#!/bin/bash
# Round-5 lab session 1 (VERDICT r4 item 1 a/b): what bounds k_fwd_gemm_ph.
#  a) the gathered stream alone with NO duplicate requests (only the even column tile asks / every sibling only its half of the rows),
#     with TCC counters; b) per-wave shader-clock stamps of the product kernel (cold rows, Infinity-Cache-resident rows, ablations).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=tools/lab/fwd_dr_lab
V1="ph_lead,ph_plain,dr12_onlyA,dr12_onlyB,dr12_onlyAB,dr12_even_onlyA,dr12_even_onlyAB,dr12_split_onlyA,dr12_split_onlyAB"
V2="ph_ts_lead,ph_ts_plain,ph_ts_lead_hotA,ph_ts_lead_nostream,ph_ts_lead_nomm,ph_ts_lead_nostore"
timeout 600 $L 20650 40 3 "$V1" 0 > $O/r05_lab1_streams_cold.txt 2>&1
timeout 600 $L 20650 40 2 "$V1" 1 > $O/r05_lab1_streams_ic.txt 2>&1
timeout 600 $L 20650 40 2 "$V2" 0 > $O/r05_lab1_stamps_cold.txt 2>&1
timeout 600 $L 20650 40 2 "$V2" 1 > $O/r05_lab1_stamps_ic.txt 2>&1
for v in dr12_onlyA dr12_even_onlyA dr12_split_onlyA dr12_onlyAB dr12_split_onlyAB; do
  P=$O/r05_lab1_pmc_$v
  rm -rf $P
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $P -o p -- $L 20650 6 1 "$v" 0 > $O/r05_lab1_pmc_$v.log 2>&1
done
python3 - > $O/r05_lab1_pmc_summary.txt <<'PY'
import csv, glob, collections
for v in ("dr12_onlyA", "dr12_even_onlyA", "dr12_split_onlyA", "dr12_onlyAB", "dr12_split_onlyAB"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r05_lab1_pmc_%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "k_fwd_gemm_dr" in n and ("Li22E" in n or "Li6E" in n):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, " ".join("%s=%.4g (n=%d)" % (c, sum(x[-6:]) / len(x[-6:]), len(x)) for c, x in sorted(acc.items())))
PY
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
cat $O/r05_lab1_streams_cold.txt | grep -v "^check" | tail -30
cat $O/r05_lab1_stamps_cold.txt | grep -v "^check" | tail -30
cat $O/r05_lab1_pmc_summary.txt
