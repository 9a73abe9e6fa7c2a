#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, --output-format csv) into one
small JSON for profiles/: per-kernel average KiB per dispatch and the corrected HBM traffic of the rollout kernels.

Usage: pmc_summary.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out.json> [note ...]
Correction (MI355X_MICROARCH.md, HBM section): on gfx950 rocprofv3's FETCH_SIZE tallies 128-byte requests at 64 bytes —
doubled here; WRITE_SIZE is taken as reported (for the rollout kernels it equals costs + partial records exactly)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("mppi::kernels::", "").replace("mppi::sampling_distributions::", "")
    return name


def collect(d, counter):
    acc = defaultdict(list)
    # gpurun merges every call's output into the same directory: only the newest run counts
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: {"dispatches": len(v), "avg_kb": sum(v) / len(v), "min_kb": min(v), "max_kb": max(v)} for k, v in acc.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    note = " ".join(sys.argv[4:])
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        e = {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)}
        if k.startswith("rollout") and fetch.get(k) and write.get(k):
            e["hbm_traffic_bytes_per_launch"] = (2.0 * fetch[k]["avg_kb"] + write[k]["avg_kb"]) * 1024.0
        kernels[k] = e
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --steps 100 "
                          "--warmup 20 --no-cpu-baseline ; the same with --pmc WRITE_SIZE (separate passes)",
               "units": "rocprofv3 FETCH_SIZE / WRITE_SIZE are KiB per dispatch",
               "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled; WRITE_SIZE as reported",
               "note": note, "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
