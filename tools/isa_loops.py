#!/usr/bin/env python3
"""Lists every loop (backward branch) of one kernel in libmppi_amd.so with its instruction mix — the role loops of the
pipelined kernels show up as separate entries.  Usage: isa_loops.py <substring of the mangled kernel name> [min_len]
Writes the kernel's disassembly to /tmp/isa/kernel.s."""
import os
import re
import struct
import subprocess
import sys
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def disassemble(workdir="/tmp/isa"):
    os.makedirs(workdir, exist_ok=True)
    so = os.path.join(REPO, "mppi-generic_amd", "lib", "libmppi_amd.so")
    fat = os.path.join(workdir, "fat.bin")
    subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so], check=True)
    b = open(fat, "rb").read()
    n = struct.unpack_from("<Q", b, 24)[0]
    o = 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", b, o)
        o += 24
        triple = b[o:o + tl].decode()
        o += tl
        if "gfx950" in triple:
            open(os.path.join(workdir, "dev.co"), "wb").write(b[off:off + size])
    out = os.path.join(workdir, "lib.s")
    with open(out, "w") as f:
        subprocess.run([LLVM + "llvm-objdump", "-d", os.path.join(workdir, "dev.co")], stdout=f, check=True)
    return out


def main():
    key = sys.argv[1]
    min_len = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    lines = open(disassemble()).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if re.match(r"^[0-9a-f]{16} <", l):
            if start is not None:
                end = i
                break
            if key in l:
                start = i
    body = lines[start + 1:end]
    open("/tmp/isa/kernel.s", "w").write("\n".join(body))
    ins = []
    for l in body:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+): ([0-9A-Fa-f]+)", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), int(m.group(4), 16)))
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    print(lines[start][:200])
    print("instructions:", len(ins), " scratch ops:", sum(1 for x in ins if x[1].startswith("scratch")))
    for a, op, w in ins:
        if op.startswith("s_cbranch") or op == "s_branch":
            simm = w & 0xFFFF
            if simm >= 0x8000:
                simm -= 0x10000
            tgt = a + 4 + simm * 4
            if tgt < a and tgt in addr and addr[a] - addr[tgt] + 1 >= min_len:
                c = Counter()
                for _, o2, _ in ins[addr[tgt]:addr[a] + 1]:
                    if o2.startswith("v_mfma"):
                        c["mfma"] += 1
                    elif o2.startswith("v_"):
                        c["valu"] += 1
                    elif o2.startswith("ds_"):
                        c["lds"] += 1
                    elif o2.startswith("s_waitcnt"):
                        c["wait"] += 1
                    elif o2.startswith("s_"):
                        c["salu"] += 1
                    else:
                        c["vmem"] += 1
                print("%#x -> %#x  n=%d  %s" % (tgt, a, addr[a] - addr[tgt] + 1, dict(c)))


if __name__ == "__main__":
    main()
