#!/usr/bin/env python3
"""Loops of one kernel of ONE model translation unit, with a per-opcode histogram of each loop (CPU only).

Usage: isa_tu.py <csrc/models/xxx.hip | path> <substring(s) of the mangled kernel name, '+'-separated> [min_len] [--ops N]
Compiles the unit for gfx950 device-only (cached in /tmp/isa/<name>.co by mtime of the tree), disassembles it and lists
every backward branch of the kernel spanning >= min_len instructions: instruction mix, then the N most frequent opcodes.
The kernel's disassembly is written to /tmp/isa/<name>.<kernel-index>.s."""
import os
import re
import subprocess
import sys
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
WORK = "/tmp/isa"


def newest_source_mtime():
    t = 0
    for root in (os.path.join(REPO, "include"), os.path.join(REPO, "mppi-generic_amd", "csrc")):
        for d, dirs, files in os.walk(root):
            dirs[:] = [x for x in dirs if x != "build"]
            for f in files:
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def disassemble(tu):
    os.makedirs(WORK, exist_ok=True)
    name = os.path.splitext(os.path.basename(tu))[0]
    co, s = os.path.join(WORK, name + ".co"), os.path.join(WORK, name + ".s")
    if not os.path.exists(s) or os.path.getmtime(s) < newest_source_mtime():
        extra = os.environ.get("MPPI_HIPCC_EXTRA", "").split()
        sys.path.insert(0, os.path.join(REPO, "mppi-generic_amd"))
        import buildlib  # the unit's own flags (buildlib.UNIT_FLAGS): the ISA that is counted is the ISA that ships
        extra = buildlib.unit_flags(tu) + extra
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"] + extra +
                       ["-I" + os.path.join(REPO, "include"), "-I" + os.path.join(REPO, "mppi-generic_amd", "csrc"),
                        "--cuda-device-only", "--no-gpu-bundle-output", "-c", tu, "-o", co], check=True)
        with open(s, "w") as f:
            subprocess.run([LLVM + "llvm-objdump", "-d", co], stdout=f, check=True)
    return s, name


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")) or "dpp" in op:
        return "xlane"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "valu_cmpsel"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "vmem"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    nops = 16
    if "--ops" in sys.argv:
        nops = int(sys.argv[sys.argv.index("--ops") + 1])
        args = [a for a in args if a != str(nops)] if str(nops) in args[2:] else args
    tu, keys = args[0], args[1].split("+")
    min_len = int(args[2]) if len(args) > 2 else 40
    if not os.path.exists(tu):
        tu = os.path.join(REPO, "mppi-generic_amd", "csrc", "models", tu)
    path, name = disassemble(tu)
    lines = open(path).read().split("\n")
    heads = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]{16} <", l)] + [len(lines)]
    idx = 0
    for a, b in zip(heads, heads[1:]):
        if not all(k in lines[a] for k in keys):
            continue
        body = lines[a + 1:b]
        open(os.path.join(WORK, "%s.%d.s" % (name, idx)), "w").write("\n".join(body))
        ins = []
        for l in body:
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+): ([0-9A-Fa-f]+)", l)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), int(m.group(4), 16)))
        addr = {x[0]: i for i, x in enumerate(ins)}
        print("kernel[%d]: %s" % (idx, lines[a][18:220]))
        print("  instructions: %d   scratch ops: %d" % (len(ins), sum(1 for x in ins if x[1].startswith("scratch"))))
        for ad, op, w in ins:
            if op.startswith("s_cbranch") or op == "s_branch":
                simm = w & 0xFFFF
                if simm >= 0x8000:
                    simm -= 0x10000
                tgt = ad + 4 + simm * 4
                if tgt < ad and tgt in addr and addr[ad] - addr[tgt] + 1 >= min_len:
                    loop = ins[addr[tgt]:addr[ad] + 1]
                    c = Counter(classify(o) for _, o, _ in loop)
                    print("  loop %#x -> %#x  n=%d  %s" % (tgt, ad, len(loop), dict(sorted(c.items()))))
                    print("     ", Counter(o for _, o, _ in loop).most_common(nops))
        idx += 1


if __name__ == "__main__":
    main()
