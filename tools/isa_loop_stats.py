#!/usr/bin/env python3
"""Instruction mix of the hottest loop of a kernel in a gfx950 disassembly (llvm-objdump -d of the device code object).

Usage: isa_loop_stats.py lib.s <substring of the mangled kernel name> [more substrings ...]
Finds the backward branch spanning the most instructions (the step loop) and prints instruction counts by class
(VALU / MFMA / LDS / VMEM / SALU / waitcnt / transcendental / division helpers)."""
import re
import sys
from collections import Counter


def functions(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^([0-9a-f]+) <(.*)>:", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(2), []
        elif name and re.match(r"^\s+[sv]_|^\s+(ds|global|buffer|flat|scratch)_", line):
            body.append(line.strip())
    if name:
        yield name, body


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_div_", "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "valu_trans/div"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")) or "dpp" in op:
        return "cross_lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, keys = sys.argv[1], sys.argv[2:]
    for name, body in functions(path):
        if not all(k in name for k in keys):
            continue
        addr = []
        for ln in body:
            m = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
            addr.append(int(m.group(1), 16) if m else None)
        best = None
        for i, ln in enumerate(body):
            m = re.match(r"s_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)", ln)
            if not m or addr[i] is None:
                continue
            off = int(m.group(1) or m.group(2))
            if off >= 32768:
                off -= 65536
            tgt = addr[i] + 4 + 4 * off
            if tgt < addr[i]:
                j = next((k for k, a in enumerate(addr) if a == tgt), None)
                if j is not None and (best is None or i - j > best[1] - best[0]):
                    best = (j, i)
        print("kernel:", name[:160])
        print("  instructions in function:", len(body))
        if best:
            loop = body[best[0]:best[1] + 1]
            c = Counter(classify(l.split()[0]) for l in loop)
            print("  largest loop: %d instructions  %s" % (len(loop), dict(c)))
            ops = Counter(l.split()[0] for l in loop)
            print("  top ops:", ops.most_common(14))


if __name__ == "__main__":
    main()
