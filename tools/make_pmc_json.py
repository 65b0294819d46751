#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/sessions/*_final.sh) -> profiles/pmc_latest.json.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: /opt/skills/guides/MI355X_MICROARCH.md says FETCH_SIZE
reports half of the bytes of wide coalesced reads on gfx950 (64-B tally of 128-B requests); unit KiB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
NAMES = {"k_fwd_gemm": "fwd_gemm", "k_wgrad_gemm": "wgrad_gemm", "k_score_loss": "score_loss", "k_score_fwd": "score_loss",
         "k_reduce_sgd": "reduce_sgd", "k_reduce": "reduce", "k_sgd": "sgd", "k_segsum": "segsum", "k_seg_bwd": "segsum", "k_dd_claim": "dd_claim", "k_dd_leaders": "dd_leaders",
         "k_dd_map": "dd_map", "k_dd_segstart": "dd_segstart", "k_dd_pos": "dd_pos"}


def short(n):
    n = n.split("(")[0]
    for k, v in NAMES.items():
        if k in n:
            return v
    return None


out = {"_source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py (tools/sessions/*_final.sh); "
                  "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled as "
                  "/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950"}
for mode in ("on", "off"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "dedup_" + mode, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    d = {}
    for k, cs in acc.items():
        fe = sum(cs["FETCH_SIZE"]) / max(len(cs["FETCH_SIZE"]), 1) if "FETCH_SIZE" in cs else None
        wr = sum(cs["WRITE_SIZE"]) / max(len(cs["WRITE_SIZE"]), 1) if "WRITE_SIZE" in cs else None
        if fe is None or wr is None:
            continue
        d[k] = {"fetch_kib": round(fe, 1), "write_kib": round(wr, 1), "hbm_bytes_per_launch": int((2 * fe + wr) * 1024)}
    out["dedup_" + mode] = d
print(json.dumps(out, indent=1))
