#!/usr/bin/env python3
"""Soak of the role-pipelined Robust MPPI kernels on the four-lanes-per-rollout RACER models and the NN models (the kernels
that run at the register limit of their 960-thread block; see the replica-divergent store finding in DESIGN.md §5): many
launches under injected noise and the in-kernel Philox stream, every cost finite, the same noise gives the same bits every
time, and the fused kernel gives the same bits as the pipelined one.
Usage: python tools/soak_robust_pipeline.py [launches] [out.json]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import mppi_generic_amd as m  # noqa: E402
from common import host_noise  # noqa: E402
import test_rmppi as tr  # noqa: E402

target = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
MODELS = ("elevation", "lstm_steering", "suspension", "complete", "autorally", "lstm", "di")
K, T = 1024, 40
per_case = max(1, target // (2 * len(MODELS)))
total = nonfinite = mismatches = fused_diff = 0
t0 = time.time()
report = {}
for name in MODELS:
    cfg = tr._rm_cfg(name, K=K, T=T)
    engines = {}
    for variant in (m.MPPI_KERNEL_PIPELINE, m.MPPI_KERNEL_FUSED):
        eng, _, _ = tr._make_pair(cfg, thr=40.0, kernel_variant=variant)
        eng.setFeedbackGains(tr._gains(T, eng.STATE_DIM, eng.CONTROL_DIM), False)
        engines[variant] = eng
    S = engines[m.MPPI_KERNEL_PIPELINE].STATE_DIM
    dx = np.zeros(S, np.float32)
    dx[:min(S, 7)] = np.array([0.3, -0.2, 0.1, 0.05, 0.02, 0.01, 0.0], np.float32)[:S]
    x0 = np.stack([cfg["x0"], cfg["x0"] + dx])
    slabs = [host_noise(1, K, T, 2, seed=700 + i)[0] for i in range(3)]
    for mode in ("injected", "philox"):
        ref, bad = {}, 0
        for it in range(per_case):
            i = it % 3
            for variant, eng in engines.items():
                if variant == m.MPPI_KERNEL_FUSED and it >= 6:
                    continue  # the fused kernel: a few launches as the reference bits
                if mode == "injected":
                    eng.injectNoise(slabs[i])
                else:
                    eng.injectNoise(None)
                    eng.setSeed(100 + i)
                costs = eng.rolloutCosts(x0, 1)
                if variant == m.MPPI_KERNEL_PIPELINE:
                    total += 1
                    bad += not np.isfinite(costs).all()
                    if i in ref:
                        mismatches += not np.array_equal(ref[i].view(np.uint32), costs.view(np.uint32))
                    else:
                        ref[i] = costs.copy()
                else:
                    fused_diff += not np.array_equal(ref[i].view(np.uint32), costs.view(np.uint32))
        nonfinite += bad
        report["%s %s" % (name, mode)] = {"launches": per_case, "non_finite": int(bad)}
    for eng in engines.values():
        eng.close()
    print("%-14s %d launches per noise mode" % (name, per_case), flush=True)
out = {"launches": total, "launches_with_non_finite_costs": int(nonfinite), "launches_not_bit_reproducible": int(mismatches),
       "fused_kernel_launches_with_other_bits": int(fused_diff), "K": K, "T": T, "seconds": round(time.time() - t0, 1),
       "cases": report}
print(json.dumps({k: v for k, v in out.items() if k != "cases"}))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(json.dumps(out, indent=1) + "\n")
sys.exit(1 if (nonfinite or mismatches or fused_diff) else 0)
