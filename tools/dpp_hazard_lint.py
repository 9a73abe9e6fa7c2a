#!/usr/bin/env python3
"""Checks the gfx950 code objects of libmppi_amd.so for DPP read-after-write hazards around hand-written DPP instructions.

LSTMQuadRows (include/mppi_amd/utils/nn_helpers/lstm_quad.hpp) issues `v_fmac_f32_dpp` from inline assembly, which the
compiler's hazard recogniser does not look into.  The hardware needs
  * 2 wait states between a VALU write of a VGPR and a DPP read of that VGPR,
  * 5 wait states between a VALU write of EXEC and a DPP instruction.
An instruction is one wait state, `s_nop N` is N + 1.  The script extracts the device code of the library, disassembles it and
walks every basic-block-agnostic window in front of each *_dpp instruction (conservative: the window is taken in program
order and stops at the function start).  Exit status 1 and a listing if a hazard is found.

Round 3 added two checks of the cross-lane exchanges every four-lanes-per-rollout model runs on (ds_bpermute_b32 for the 36
values a step exchanges, v_readlane_b32 for kernel arguments spilled from SGPRs to VGPR lanes), after a NaN that appeared and
disappeared with the spill pattern of one instantiation (the (64, 4, 2) block of the suspension model):
  * a register an LDS instruction (ds_bpermute / ds_read ...) is still loading may not be touched before an s_waitcnt has
    retired that instruction (LDS returns in order: `lgkmcnt(n)` leaves at most the n newest outstanding; scalar loads count
    on the same counter and return out of order, so with one of them in flight only lgkmcnt(0) proves anything);
  * 4 wait states between a VALU write of an SGPR and a v_readlane / v_writelane that uses it as its lane select.
These instructions are the compiler's own, so a finding would be a compiler bug — none was found in 1.1 M instructions.

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


def all_vgprs(ops):
    out = set()
    for o in ops:
        for tok in re.findall(r"v\[\d+:\d+\]|v\d+", o):
            out |= vgprs(tok)
    return out


def sgprs(operand):
    m = re.fullmatch(r"s(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def lint_cross_lane(text, name):
    """(LDS results consumed before their s_waitcnt, lane-select SGPRs read too early) in program order per function"""
    hazards, n_checked = [], 0
    pending = []   # outstanding LGKM operations, oldest first: (kind, dest VGPRs, text)
    recent = []    # (wait states, opcode, operands) of the last few instructions
    # the walk is per basic block: what is in flight at a join is not known from a linear listing, so the state is dropped
    # after every branch and at every branch target (the exchanges of a rollout step are straight-line code)
    lines = text.splitlines()
    targets = set()
    for line in lines:
        mm = re.match(r"^\s+s_c?branch\w*\s+(\d+)\s*//\s*([0-9A-Fa-f]+):", line)
        if mm:
            simm = int(mm.group(1))
            if simm >= 0x8000:
                simm -= 0x10000
            targets.add(int(mm.group(2), 16) + 4 + 4 * simm)
    for line in lines:
        if line.endswith(">:"):
            pending, recent = [], []
            continue
        m = INSTR.match(line)
        if not m:
            continue
        ma = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        if ma and int(ma.group(1), 16) in targets:
            pending, recent = [], []
        op, rest = m.group(1), m.group(2)
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if op == "s_waitcnt":
            mm = re.search(r"lgkmcnt\((\d+)\)", rest)
            if mm or rest.strip() in ("0", ""):
                n = int(mm.group(1)) if mm else 0
                if any(k == "smem" for k, _, _ in pending):
                    pending = [] if n == 0 else pending
                else:
                    pending = pending[len(pending) - n:] if n else []
        elif op.startswith(("s_branch", "s_cbranch", "s_setpc", "s_endpgm", "s_swappc")):
            pending, recent = [], []
            continue
        else:
            touched = all_vgprs(ops)
            for kind, dest, txt in pending:
                if kind == "lds" and dest & touched:
                    hazards.append((name, line.strip(), "touches the destination of an LDS instruction still in flight: " + txt))
            if op.startswith(("v_readlane", "v_writelane")) and len(ops) >= 3:
                n_checked += 1
                sel = sgprs(ops[2].split()[0])
                dist = 0
                for states, wop, wops in reversed(recent):
                    if dist >= 4:
                        break
                    if sel and wop.startswith("v_") and wops and (sgprs(wops[0].split()[0]) & sel):
                        hazards.append((name, line.strip(), "lane select written by the VALU %d wait state(s) before: %s" % (dist, wop)))
                    dist += states
            if op.startswith("ds_"):
                n_checked += op.startswith(("ds_bpermute", "ds_permute"))
                is_store = op.startswith(("ds_write", "ds_store"))
                dest = set() if is_store or not ops else vgprs(ops[0].split()[0])
                pending.append(("lds", dest, line.strip()))
            elif op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
                pending.append(("smem", set(), line.strip()))
        states = 1
        if op == "s_nop":
            try:
                states = int(ops[0], 0) + 1
            except (ValueError, IndexError):
                states = 1
        recent.append((states, op, ops))
        if len(recent) > 8:
            recent.pop(0)
    return n_checked, hazards


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


def _lint_code_object(path):
    text = subprocess.run([LLVM + "llvm-objdump", "-d", path], check=True, capture_output=True, text=True).stdout
    f = os.path.basename(path)
    n, h = lint_disassembly(text, f)
    nx, hx = lint_cross_lane(text, f)
    return n, h, nx, hx


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "mppi-generic_amd", "lib", "libmppi_amd.so")
    work = tempfile.mkdtemp(prefix="dpp_lint_")
    try:
        local = os.path.join(work, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([LLVM + "llvm-objdump", "--offloading", local], check=True, stdout=subprocess.DEVNULL, cwd=work)
        total, total_x, hazards = 0, 0, []
        files = [os.path.join(work, f) for f in sorted(os.listdir(work)) if "gfx950" in f]
        # one process per code object (disassembly + the scan, which is pure Python): the library holds ~25 of them
        import concurrent.futures
        with concurrent.futures.ProcessPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as pool:
            for n, h, nx, hx in pool.map(_lint_code_object, files):
                total += n
                total_x += nx
                hazards += h + hx
        print("%d DPP instructions checked, %d hazard(s)" % (total, len(hazards)))
        print("%d ds_bpermute / v_readlane / v_writelane instructions checked" % total_x)
        for h in hazards[:50]:
            print("  %s\n    %s\n    %s" % h)
        return 1 if hazards else 0
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
