#!/bin/bash
# SQ counters of the phase-staggered kernels (variant 5), dedup on and off
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export VV_GEMM_VARIANT=5
mkdir -p gpurun_out
for mode in on off; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    P=gpurun_out/sq5_${mode}_$i
    rm -rf $P
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $P -o sq -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --dedup $mode > gpurun_out/sq5_${mode}_$i.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for mode in ("on", "off"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/sq5_%s_*/**/*counter_collection.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0]
            for k in ("k_fwd_gemm", "k_wgrad_gemm", "k_score_loss"):
                if k in n:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== dedup", mode)
    for k, cs in acc.items():
        print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
