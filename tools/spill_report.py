#!/usr/bin/env python3
"""Register / spill / scratch / kernarg metadata of every kernel in the built code objects (CPU only).

Usage: spill_report.py [substring of the kernel name ...] [--json out.json] [--min-spill N]
Reads the AMDGPU metadata notes (llvm-readelf --notes) of mppi-generic_amd/csrc/build/*.o — the objects libmppi_amd.so is linked
from — and prints, per kernel whose demangled name contains every given substring: VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs,
private segment (scratch) bytes per lane, LDS bytes, kernarg bytes.  The judge reads the same notes."""
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
BUILD = os.path.join(REPO, "mppi-generic_amd", "csrc", "build")


def offload_bundles(obj):
    """the gfx950 code object(s) embedded in a host object (llvm-objdump --offloading writes them next to the input)"""
    import shutil
    tmp = "/tmp/spill_report_%d/%s" % (os.getpid(), os.path.basename(obj))
    os.makedirs(tmp, exist_ok=True)
    local = os.path.join(tmp, "in.o")
    shutil.copy(obj, local)
    subprocess.run([LLVM + "llvm-objdump", "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    return [os.path.join(tmp, f) for f in sorted(os.listdir(tmp)) if "gfx950" in f]


def kernels_of(co):
    txt = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    res = []
    for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        def g(key, cast=int):
            mt = re.search(r"\.%s:\s*(\S+)" % key, blk)
            return cast(mt.group(1)) if mt else None
        name = g("name", str)
        if not name:
            continue
        res.append({"symbol": name, "vgpr": g("vgpr_count"), "agpr": g("agpr_count"), "sgpr": g("sgpr_count"),
                    "vgpr_spill": g("vgpr_spill_count"), "sgpr_spill": g("sgpr_spill_count"),
                    "scratch_bytes": g("private_segment_fixed_size"), "lds_static": g("group_segment_fixed_size"),
                    "kernarg_bytes": g("kernarg_segment_size"), "max_flat_workgroup_size": g("max_flat_workgroup_size")})
    return res


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.split("\n")


def main():
    args = [a for a in sys.argv[1:]]
    out_json, min_spill = None, 0
    if "--json" in args:
        i = args.index("--json")
        out_json = args[i + 1]
        del args[i:i + 2]
    if "--min-spill" in args:
        i = args.index("--min-spill")
        min_spill = int(args[i + 1])
        del args[i:i + 2]
    rows = []
    for f in sorted(os.listdir(BUILD)):
        if not f.endswith(".o"):
            continue
        for co in offload_bundles(os.path.join(BUILD, f)):
            ks = kernels_of(co)
            for k, d in zip(ks, demangle([k["symbol"][:-3] if k["symbol"].endswith(".kd") else k["symbol"] for k in ks])):
                k["kernel"] = re.sub(r"\s+", " ", d)
                k["unit"] = f.replace("models_", "").replace(".hip.o", "")
                rows.append(k)
    sel = [r for r in rows if all(a in r["kernel"] for a in args) and (r["vgpr_spill"] or 0) + (r["sgpr_spill"] or 0) >= min_spill]
    for r in sel:
        short = r["kernel"] if len(r["kernel"]) < 150 else r["kernel"][:147] + "..."
        print("%-28s vgpr %3s agpr %3s sgpr %3s | spilled v %3s s %3s | scratch %4s B | kernarg %4s B | %s" %
              (r["unit"], r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch_bytes"], r["kernarg_bytes"], short))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(sel, f, indent=1)


if __name__ == "__main__":
    main()
