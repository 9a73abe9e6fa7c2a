#!/usr/bin/env python3
"""Rough in-order issue model of a straight-line region of gfx950 ISA (one wave alone on its SIMD).

Usage: isa_sim.py /tmp/isa/kernel.s <start addr hex> <end addr hex> [--list]
Prints, for the region: instruction count, the in-order finish time under the latency model below, the dependency-only
critical path (what a perfect static schedule could reach with unlimited issue), and the pure issue bound.  Branches
inside the region are assumed not taken (the step loops' groups of 4 are one basic block plus wait loops that fall
through).  Latencies (cycles) are the ones measured with tools/ubench/valu_latency.hip on MI355X: dependent VALU 6.3,
back-to-back independent VALU 2.25; transcendental / LDS / SALU values are estimates."""
import re
import sys

VALU_LAT, VALU_ISSUE = 6.3, 2.25
TRANS_LAT, TRANS_ISSUE = 12.0, 4.5
LDS_LAT, LDS_ISSUE = 64.0, 2.25
SALU_LAT, SALU_ISSUE = 2.0, 1.0
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def regs(tok):
    tok = tok.strip().strip("|").lstrip("-")
    tok = re.sub(r"^(abs|neg)\((.*)\)$", r"\2", tok)
    m = re.match(r"^([vs])(\d+)$", tok)
    if m:
        return [m.group(1) + m.group(2)]
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if m:
        return [m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if tok in ("vcc", "exec", "scc", "m0", "vcc_lo", "vcc_hi"):
        return [tok[:3] if tok.startswith("vcc") else tok]
    return []


def parse(line):
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
    if not m:
        return None
    op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
    toks = [t for t in re.split(r",\s*", args) if t] if args else []
    toks = [t.split(" ")[0] for t in toks]  # drop modifiers like offset:.. clamp
    dst, src = [], []
    ndst = 1
    if op.startswith(("ds_write", "s_waitcnt", "s_nop", "s_sleep", "s_branch", "s_cbranch", "s_barrier", "s_cmp",
                      "s_bitcmp", "global_store", "buffer_store", "s_endpgm")):
        ndst = 0
    if op.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")) or (op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co")) and op.endswith("e64")):
        ndst = 2
    for i, t in enumerate(toks):
        (dst if i < ndst else src).extend(regs(t))
    if op.startswith("v_cmp") and op.endswith("e32"):
        dst, src = ["vcc"], [r for t in toks for r in regs(t)]
    if op.startswith("v_cmpx"):
        dst = ["exec"] + dst
    if op.startswith(("v_cndmask", "v_div_fmas", "v_addc", "v_subb")) and op.endswith("e32"):
        src.append("vcc")
    if op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co")) and op.endswith("e32"):
        dst.append("vcc")
    if op.startswith(("s_cmp", "s_bitcmp")) or op.startswith(("s_add", "s_sub", "s_and", "s_or", "s_xor", "s_lshl", "s_lshr", "s_min", "s_max", "s_andn2", "s_orn2", "s_abs")):
        dst.append("scc")
    if op.startswith(("s_cselect", "s_addc", "s_subb", "s_cbranch_scc")):
        src.append("scc")
    if op.startswith("s_cbranch_vcc"):
        src.append("vcc")
    if op.startswith("s_cbranch_exec"):
        src.append("exec")
    if op.startswith("v_") and not op.startswith("v_readfirstlane"):
        src.append("exec")
    if op.startswith(("v_fmac", "v_mac", "v_pk_fmac")) or op.startswith(("v_writelane",)):
        src.extend(dst)
    if op.endswith("_saveexec_b64"):
        dst.append("exec")
        src.append("exec")
    return addr, op, dst, src


def kind(op):
    if op.startswith(TRANS):
        return TRANS_LAT, TRANS_ISSUE
    if op.startswith("v_"):
        return VALU_LAT, VALU_ISSUE
    if op.startswith("ds_"):
        return LDS_LAT, LDS_ISSUE
    return SALU_LAT, SALU_ISSUE


def main():
    path, lo, hi = sys.argv[1], int(sys.argv[2], 16), int(sys.argv[3], 16)
    ins = [p for p in (parse(l) for l in open(path)) if p and lo <= p[0] <= hi]
    ready = {}      # register -> cycle its value is available (in-order model)
    depth = {}      # register -> dependency-only completion time
    t = 0.0
    crit = 0.0
    issue_bound = 0.0
    pending_lds = []
    for addr, op, dst, src in ins:
        lat, iss = kind(op)
        issue_bound += iss
        if op.startswith("s_waitcnt"):
            if "lgkmcnt" in " ".join(sys.argv) or True:
                if pending_lds:
                    t = max(t, max(pending_lds))
                    pending_lds = []
            continue
        start = t
        dstart = 0.0
        for r in src:
            if not op.startswith("ds_") or True:
                start = max(start, ready.get(r, 0.0)) if not r.startswith("v") or r not in LDS_DST else start
            dstart = max(dstart, depth.get(r, 0.0))
        for r in src:
            if r in LDS_DST:
                pass
        done = start + lat
        for r in dst:
            ready[r] = done
            depth[r] = dstart + lat
            if op.startswith("ds_read"):
                LDS_DST.add(r)
            else:
                LDS_DST.discard(r)
        if op.startswith("ds_"):
            pending_lds.append(done)
        crit = max(crit, dstart + lat)
        t = start + iss
        if "--list" in sys.argv:
            print("%8.1f %8.1f  %s" % (start, dstart, op))
    print("instructions %d  in-order %.0f cycles  critical path %.0f cycles  issue bound %.0f cycles" %
          (len(ins), max([t] + list(ready.values())), crit, issue_bound))


LDS_DST = set()
if __name__ == "__main__":
    main()
