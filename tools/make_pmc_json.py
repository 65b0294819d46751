#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/sessions/r06.sh final; bench.py's own live passes) -> HBM bytes per launch.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: /opt/skills/guides/MI355X_MICROARCH.md says FETCH_SIZE
reports half of the bytes of wide coalesced reads on gfx950 (64-B tally of 128-B requests); unit KiB.
As a script: `make_pmc_json.py <dir with dedup_on/ dedup_off/>` prints profiles/pmc_latest.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

NAMES = {"k_fwd_gemm": "fwd_gemm", "k_wgrad_gemm": "wgrad_gemm", "k_score_loss": "score_loss", "k_score_fwd": "score_loss", "k_score_stream": "score_loss",
         "k_reduce_sgd": "reduce_sgd", "k_reduce": "reduce", "k_sgd": "sgd", "k_segsum": "segsum", "k_seg_bwd": "segsum", "k_dd_claim": "dd_claim", "k_dd_leaders": "dd_leaders",
         "k_dd_map": "dd_map", "k_dd_segstart": "dd_segstart", "k_dd_pos": "dd_pos"}
SOURCE = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB, "
          "FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950")


def short(n):
    n = n.split("(")[0]
    for k, v in NAMES.items():
        if k in n:
            return v
    return None


def fold(root):
    """Per kernel (bench.py's names): the average FETCH_SIZE / WRITE_SIZE over all launches found in the counter_collection.csv files below
    `root`, and the HBM bytes per launch they give.  Kernels seen in only one of the two passes are left out."""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    d = {}
    for k, cs in acc.items():
        if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
            continue
        fe = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
        wr = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        d[k] = {"fetch_kib": round(fe, 1), "write_kib": round(wr, 1), "hbm_bytes_per_launch": int((2 * fe + wr) * 1024),
                "launches": min(len(cs["FETCH_SIZE"]), len(cs["WRITE_SIZE"]))}
    return d


if __name__ == "__main__":
    out = {"_source": SOURCE}
    for mode in ("on", "off"):
        out["dedup_" + mode] = fold(os.path.join(sys.argv[1], "dedup_" + mode))
    print(json.dumps(out, indent=1))
