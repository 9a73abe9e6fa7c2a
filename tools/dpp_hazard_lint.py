#!/usr/bin/env python3
"""Checks the gfx950 code objects of libmppi_amd.so for DPP read-after-write hazards around hand-written DPP instructions.

LSTMQuadRows (include/mppi_amd/utils/nn_helpers/lstm_quad.hpp) issues `v_fmac_f32_dpp` from inline assembly, which the
compiler's hazard recogniser does not look into.  The hardware needs
  * 2 wait states between a VALU write of a VGPR and a DPP read of that VGPR,
  * 5 wait states between a VALU write of EXEC and a DPP instruction.
An instruction is one wait state, `s_nop N` is N + 1.  The script extracts the device code of the library, disassembles it and
walks every basic-block-agnostic window in front of each *_dpp instruction (conservative: the window is taken in program
order and stops at the function start).  Exit status 1 and a listing if a hazard is found.

Usage: dpp_hazard_lint.py [path/to/libmppi_amd.so]   (CPU only)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INSTR = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//")


def vgprs(operand):
    """register numbers an operand like v12 or v[4:7] names"""
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def lint_disassembly(text, name):
    hazards, n_dpp = [], 0
    window = []  # (wait states it provides, opcode, operands) of the instructions in front, newest last
    for line in text.splitlines():
        if line.endswith(">:"):
            window = []
            continue
        m = INSTR.match(line)
        if not m:
            continue
        op, rest = m.group(1), m.group(2)
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if op.endswith("_dpp"):
            n_dpp += 1
            src0 = ops[1].split()[0] if len(ops) > 1 else ""
            need = vgprs(src0)
            dist = 0
            for states, wop, wops in reversed(window):
                is_valu = wop.startswith("v_")
                if is_valu and dist < 2 and wops and (vgprs(wops[0].split()[0]) & need):
                    hazards.append((name, line.strip(), "VGPR written %d wait state(s) before: %s %s" % (dist, wop, ", ".join(wops))))
                if is_valu and dist < 5 and (wop.startswith("v_cmpx") or (wops and wops[0].startswith("exec"))):
                    hazards.append((name, line.strip(), "EXEC written by the VALU %d wait state(s) before: %s" % (dist, wop)))
                dist += states
                if dist >= 5:
                    break
        states = 1
        if op == "s_nop":
            try:
                states = int(ops[0], 0) + 1
            except (ValueError, IndexError):
                states = 1
        window.append((states, op, ops))
        if len(window) > 8:
            window.pop(0)
    return n_dpp, hazards


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "mppi-generic_amd", "lib", "libmppi_amd.so")
    work = tempfile.mkdtemp(prefix="dpp_lint_")
    try:
        local = os.path.join(work, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([LLVM + "llvm-objdump", "--offloading", local], check=True, stdout=subprocess.DEVNULL, cwd=work)
        total, hazards = 0, []
        for f in sorted(os.listdir(work)):
            if "gfx950" not in f:
                continue
            text = subprocess.run([LLVM + "llvm-objdump", "-d", os.path.join(work, f)], check=True, capture_output=True, text=True).stdout
            n, h = lint_disassembly(text, f)
            total += n
            hazards += h
        print("%d DPP instructions checked, %d hazard(s)" % (total, len(hazards)))
        for h in hazards[:50]:
            print("  %s\n    %s\n    %s" % h)
        return 1 if hazards else 0
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
