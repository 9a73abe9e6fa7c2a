#!/usr/bin/env python3
"""Random sizes through every form of the elevation-map RACER models against the oracle (trajectory costs bit for bit):
partial blocks, horizons that are not a multiple of the four replica lanes' group, one and two systems, the one-lane and
the four-lane form, fused and role-pipelined kernel.  Usage: timeout 900 python tools/racer_shape_sweep.py [trials] [seed]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from common import host_noise, make_engine, make_oracle, ulp_diff  # noqa: E402
from test_racer_dubins_elevation import elevation_cfg  # noqa: E402
from test_racer_dubins_lstm_steering import steering_cfg  # noqa: E402
from test_racer_dubins_lstm_unc import uncertainty_cfg  # noqa: E402
from test_racer_dubins_suspension import suspension_cfg  # noqa: E402

MODELS = {"elevation": elevation_cfg, "lstm_steering": steering_cfg, "suspension": suspension_cfg, "complete": uncertainty_cfg}
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for trial in range(trials):
    name = list(MODELS)[trial % len(MODELS)]
    K = int(rng.choice([1, 3, 15, 16, 17, 63, 64, 65, 100, 1000, int(rng.integers(1, 3000))]))
    T = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, int(rng.integers(1, 130))]))
    D = int(rng.choice([1, 1, 2]))
    by = int(rng.choice([1, 4, 4]))
    variant = int(rng.choice([0, 1, 2])) if (by == 4 and D == 1) else int(rng.choice([0, 1]))
    cfg = MODELS[name](K=K, T=T, D=D)
    eps = host_noise(1, K, T, 2, seed=trial)
    o = make_oracle(cfg)
    (o.tube_compute_control if D == 2 else o.vanilla_compute_control)(cfg["x0"], 1, eps)
    shape = {} if (by == 4 and D == 2) else dict(block_x=64, block_y=by)  # two systems, four lanes: the registered default
    try:
        eng = make_engine(cfg, kernel_variant=variant, **shape)
        eng.injectNoise(eps)
        eng.computeControl(cfg["x0"], 1)
        worst = int(ulp_diff(eng.getSampledCostSeq(), o.costs()).max())
        du = float(np.abs(eng.getControlSeq() - o.control()).max())
        ok = worst == 0 and du <= 1e-5
        eng.close()
    except Exception as e:  # noqa: BLE001
        worst, du, ok = -1, float("nan"), False
        print("  exception:", str(e)[:120])
    bad += not ok
    print("%-13s K=%-5d T=%-4d D=%d by=%d variant=%d: costs %d ulp, controls %.1e %s" % (name, K, T, D, by, variant, worst, du, "" if ok else "  <-- MISMATCH"),
          flush=True)
print("sweep:", trials, "trials,", bad, "mismatch(es)")
sys.exit(1 if bad else 0)
