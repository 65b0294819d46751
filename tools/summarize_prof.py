#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel launch count / average duration from kernel traces and
per-kernel counter averages from counter-collection files."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(n):
    n = n.split("(")[0]
    for k in ("k_fwd_gemm", "k_wgrad_gemm", "k_score_loss", "k_score_fwd", "k_seg_bwd", "k_reduce_sgd", "k_reduce", "k_sgd", "k_map_rows",
              "k_final_loss", "k_scale_update", "k_segsum", "k_dd_claim", "k_dd_leaders", "k_dd_map", "k_dd_segstart", "k_dd_pos"):
        if k in n:
            return k
    return n[:60]


for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
    d = defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("== kernel trace", f)
    tot = sum(sum(v) for v in d.values())
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print("%-28s calls %5d  avg %9.1f us  med %9.1f us  min %9.1f us  share %5.1f%%" %
              (k, len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3, v2[0] / 1e3, 100.0 * sum(v) / tot))
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    d = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== counters", f)
    for k, cs in d.items():
        for c, v in cs.items():
            print("%-28s %-14s n %5d  avg %.6g" % (k, c, len(v), sum(v) / len(v)))
