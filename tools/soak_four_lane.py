#!/usr/bin/env python3
"""Soak of the four-lanes-per-rollout RACER models (ds_bpermute exchanges, DPP-row weights, spilled kernel arguments) under
INJECTED noise, one and two systems per launch, every registered four-lane block shape: many launches, every cost finite,
and the same injected noise gives the same bits every time.

Background: the (64, 4, 2) block of the suspension model once returned NaN costs for injected noise at -O3 (round 2; see
csrc/models/racer_dubins_elevation_suspension.hip).  Usage: python tools/soak_four_lane.py [launches] [out.json]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from common import host_noise, make_engine  # noqa: E402
from test_racer_dubins_elevation import elevation_cfg  # noqa: E402
from test_racer_dubins_lstm_steering import steering_cfg  # noqa: E402
from test_racer_dubins_lstm_unc import uncertainty_cfg  # noqa: E402
from test_racer_dubins_suspension import suspension_cfg  # noqa: E402

target = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
MODELS = (("elevation", elevation_cfg, [(64, 4)]), ("lstm_steering", steering_cfg, [(64, 4)]),
          ("suspension", suspension_cfg, [(64, 4), (32, 4)]), ("complete", uncertainty_cfg, [(64, 4), (32, 4)]))
K, T = 2048, 40
cases = []
for name, mk, shapes in MODELS:
    for D in (1, 2):
        for bx, by in shapes:
            if D == 1 and bx == 32:
                continue
            if name == "complete" and D == 2 and bx == 64:
                continue  # not instantiated: (32, 4, 2) is that model's two-system block
            cases.append((name, mk, D, bx, by))
per_case = max(1, target // len(cases))
total, nonfinite, mismatches = 0, 0, 0
t0 = time.time()
report = {}
for name, mk, D, bx, by in cases:
    cfg = mk(K=K, T=T, D=D)
    eng = make_engine(cfg, block_x=bx, block_y=by)
    x0 = np.tile(cfg["x0"], (D, 1))
    ref = {}
    n_slabs = 4
    slabs = [host_noise(1, K, T, 2, seed=900 + i) for i in range(n_slabs)]
    bad = 0
    for it in range(per_case):
        i = it % n_slabs
        eng.injectNoise(slabs[i])
        costs = eng.rolloutCosts(x0, 1)
        total += 1
        if not np.isfinite(costs).all():
            bad += 1
        if i in ref:
            if not np.array_equal(ref[i].view(np.uint32), costs.view(np.uint32)):
                mismatches += 1
        else:
            ref[i] = costs.copy()
    nonfinite += bad
    report["%s D=%d (%d,%d)" % (name, D, bx, by)] = {"launches": per_case, "non_finite": bad}
    eng.close()
    print("%-14s D=%d shape=(%d,%d,%d): %d launches, %d with non-finite costs" % (name, D, bx, by, D, per_case, bad), flush=True)
out = {"launches": total, "launches_with_non_finite_costs": nonfinite, "launches_not_bit_reproducible": mismatches,
       "K": K, "T": T, "seconds": round(time.time() - t0, 1), "cases": report}
print(json.dumps({k: v for k, v in out.items() if k != "cases"}))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(json.dumps(out, indent=1) + "\n")
sys.exit(1 if (nonfinite or mismatches) else 0)
