#!/usr/bin/env python3
"""Instruction count per rollout step on the wave that carries the recurrence, from the ISA of the built kernels (CPU only).

Usage: isa_step_count.py [out.json]
For every entry of KERNELS: compile the model's translation unit device-only (tools/isa_tu.py's cache), take the kernel
whose mangled name contains all the given substrings, list its loops (backward branches) and pick the INNERMOST loop with
the most packed-fp32 / MFMA instructions — the dynamics wave's step loop (sampler and cost waves use neither) — and divide
its instruction counts by the steps one trip of that loop covers.  bench.py multiplies `instructions_per_step` with the
issue interval it measures live (mppi_measure_issue_interval) to get a floor that does not depend on the timing of the
kernel being judged (roofline.issue_floor); tests/test_abi.py checks that the committed file is what the current sources
compile to."""
import json
import os
import re
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_tu  # noqa: E402

# (key, translation unit, kernel-name substrings, steps per trip of the dynamics loop, marker class)
KERNELS = [
    # (template tails: <.., BZ = 1, DRAW_IN_LOOP, FOLD_Z = false, ROWS_HBM = false, STREAM_MERGE = true> — the instantiation the
    #  iterations of the headline run — / <.., DRAW_IN_LOOP, ROWS_HBM = false>)
    ("cartpole_pipeline_dynamics_wave", "cartpole.hip",
     ["rolloutPipelineKernel", "Cartpole", "GaussianDistribution", "ELi1ELb1ELb0ELb0ELb1EE"], 8, "valu_pk"),  # round 4: eight steps per trip
    ("autorally_mfma_pipeline_dynamics_wave", "autorally_nn.hip",
     ["rolloutPipelineRepKernel", "NeuralNetModelMFMA", "GaussianDistribution", "ELb1ELb0EE"], None, "mfma"),
    ("lstm_mfma_pipeline_dynamics_wave", "bicycle_slip_lstm.hip",
     ["rolloutPipelineRepKernel", "BicycleSlipLSTMMFMA", "GaussianDistribution", "ELb1ELb0EE"], None, "mfma"),
]


def loops_of(tu, keys):
    path, _ = isa_tu.disassemble(os.path.join(isa_tu.REPO, "mppi-generic_amd", "csrc", "models", tu))
    lines = open(path).read().split("\n")
    heads = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]{16} <", l)] + [len(lines)]
    out = []
    for a, b in zip(heads, heads[1:]):
        if not all(k in lines[a] for k in keys):
            continue
        ins = []
        for l in lines[a + 1:b]:
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+): ([0-9A-Fa-f]+)", l)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), int(m.group(4), 16)))
        addr = {x[0]: i for i, x in enumerate(ins)}
        loops = []
        for ad, op, w in ins:
            if op.startswith("s_cbranch") or op == "s_branch":
                simm = w & 0xFFFF
                if simm >= 0x8000:
                    simm -= 0x10000
                tgt = ad + 4 + simm * 4
                if tgt < ad and tgt in addr and addr[ad] - addr[tgt] + 1 >= 40:
                    loops.append([o for _, o, _ in ins[addr[tgt]:addr[ad] + 1]])
        out.append((lines[a][18:], loops))
    return out


def main():
    res = {}
    # the units' device-only compiles (tools/isa_tu.py's cache) side by side: they are what this tool's time goes to
    import concurrent.futures
    units = sorted({os.path.join(isa_tu.REPO, "mppi-generic_amd", "csrc", "models", k[1]) for k in KERNELS})
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(units)) as pool:
        list(pool.map(isa_tu.disassemble, units))
    for key, tu, subs, steps, marker in KERNELS:
        cands = loops_of(tu, subs)
        if not cands:
            res[key] = {"error": "kernel not found"}
            continue
        name, loops = cands[0]
        best = None
        for ops in loops:
            c = Counter(isa_tu.classify(o) for o in ops)
            score = (c.get(marker, 0), -len(ops))
            if c.get(marker, 0) and (best is None or score > best[0]):
                best = (score, ops, c)
        if best is None:
            res[key] = {"error": "no loop with %s instructions" % marker, "kernel": name[:160]}
            continue
        _, ops, c = best
        if steps is None:  # MFMA networks: 20 v_mfma per step and wave for the AutoRally MLP, 36 for the LSTM + MLP (round 5:
            # the 8 MFMAs of either output layer are 16 packed fmas on the vector unit now; rounds 1-4: 28 / 44)
            per_step = 36 if "LSTM" in name else 20
            steps = max(1, round(c["mfma"] / per_step))
        vec = sum(v for k, v in c.items() if k in ("valu", "valu_pk", "valu_cmpsel", "trans", "xlane", "mfma"))
        res[key] = {"kernel": name[:200], "loop_instructions": len(ops), "steps_per_trip": steps,
                    "instructions_per_step": round(len(ops) / steps, 2),
                    "vector_instructions_per_step": round(vec / steps, 2),
                    "mix_per_trip": dict(sorted(c.items()))}
    txt = json.dumps(res, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
