#!/usr/bin/env python3
"""Summarise a rocprofv3 PMC pass over the SQ matrix / vector counters into profiles/: per rollout kernel, averages per
dispatch and the derived utilisations (formulas of rocprofv3's gfx94x derived counters, which gfx950 falls back to):
  MfmaUtil  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD * SIMD_NUM)        matrix pipe busy, chip-wide
  VALUBusy  = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE per XCD * SIMD_NUM)         vector ALU busy (quad-cycle counter)
rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs of an MI355X; SIMD_NUM = 256 CUs * 4.

Usage: pmc_mfma_summary.py <rocprof output dir> <out.json> "<profiled command>" """
import csv
import glob
import json
import re
import sys
from collections import defaultdict

XCDS, SIMDS = 8, 1024


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void\s+", "", name)
    return name.replace("mppi::kernels::", "").replace("mppi::sampling_distributions::", "")


def main():
    src, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    f = sorted(glob.glob(src + "/*/*counter_collection.csv"))[-1]
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "rollout" in r["Kernel_Name"]:
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"command": cmd, "note": "averages per dispatch; GRBM_GUI_ACTIVE is the sum over %d XCDs" % XCDS, "kernels": {}}
    for k, v in acc.items():
        e = {c: sum(x) / len(x) for c, x in v.items()}
        e["dispatches"] = len(next(iter(v.values())))
        gui = e.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        if gui > 0:
            e["cycles_per_launch"] = gui
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                e["MfmaUtil_percent"] = 100.0 * e["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * SIMDS)
            if "SQ_ACTIVE_INST_VALU" in e:
                e["VALUBusy_percent"] = 100.0 * 4.0 * e["SQ_ACTIVE_INST_VALU"] / (gui * SIMDS)
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)
    for k, e in res["kernels"].items():
        print("%-100s MfmaUtil %5.1f %%  VALUBusy %5.1f %%" % (k[:100], e.get("MfmaUtil_percent", 0), e.get("VALUBusy_percent", 0)))


if __name__ == "__main__":
    main()
