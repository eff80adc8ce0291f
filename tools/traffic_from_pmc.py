#!/usr/bin/env python3
"""Per-launch HBM traffic of each kernel class from two rocprofv3 PMC passes of bench.py
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each with --kernel-trace, eager launches).

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of wide coalesced
reads (MI355X_MICROARCH.md §HBM), so reads are doubled -- that correction is calibrated for 16-B-per-lane streaming reads
only; the guide calls other access widths uncalibrated, so read the absolute number as an upper estimate and use it
for comparisons between builds.  Writes `profiles/rNN_traffic.json`:
{class: {"bytes_per_launch": ..., "launches": ..., "read_bytes": ..., "write_bytes": ...}, "git": ..., "command": ...}.
"""
import csv
import glob
import json
import os
import sys

import re

# kernel names: gemm_kernel<BM, BN, MODE, EPI, DBG> (MODE 0 plain / 1 conv3x3), attn_kernel<NW, KT, ...>
CLASSES = {"gemm": re.compile(r"gemm_kernel<(64|128|160), (128|160), 0, |ff_fused"),
           "conv": re.compile(r"gemm_kernel<(64|128|160), (128|160), [123], |conv_win_kernel"),
           "attention": re.compile(r"attn_kernel<4, 64|attn2_kernel|attn16_kernel")}


def load(d, counter):
    tot = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for cls, pat in CLASSES.items():
                if pat.search(r["Kernel_Name"]):
                    e = tot.setdefault(cls, [0.0, 0])
                    e[0] += float(r["Counter_Value"]) * 1024.0
                    e[1] += 1
    return tot


if __name__ == "__main__":
    fetch_dir, write_dir, out = sys.argv[1:4]
    rd, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    res = {}
    for cls in CLASSES:
        if cls in rd and cls in wr:
            n = rd[cls][1]
            read_b, write_b = 2.0 * rd[cls][0], wr[cls][0]
            res[cls] = {"bytes_per_launch": (read_b + write_b) / n, "launches": n,
                        "read_bytes": read_b, "write_bytes": write_b,
                        "note": "FETCH_SIZE*2 (gfx950 correction) + WRITE_SIZE, summed over all launches of the class in the profiled run (every eager step of it, warm-up included); bytes_per_launch is the figure to read"}
    import subprocess
    try:
        res["git"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("SEVA_GIT_REV")
    except Exception:
        res["git"] = os.environ.get("SEVA_GIT_REV")
    res["command"] = os.environ.get("SEVA_TRAFFIC_COMMAND", "SEVA_HIPGRAPH=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-other-configs")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
