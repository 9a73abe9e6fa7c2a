#!/usr/bin/env python3
"""Times optimisation iterations of the registered workloads for several block shapes (HIP events on the engine's
stream).  Usage: python tools/time_workloads.py [cartpole|autorally|di|lstm|racer] ..."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from common import autorally_cfg, bicycle_lstm_cfg, cartpole_cfg, di_cfg, make_engine  # noqa: E402


def run(name, cfg, shapes, n=100):
    for sh in shapes:
        bx, by = sh[0], sh[1]
        variant = sh[2] if len(sh) > 2 else 0
        try:
            eng = make_engine(cfg, block_x=bx, block_y=by, kernel_variant=variant)
        except Exception as e:  # noqa: BLE001
            print(name, (bx, by), "skip:", e)
            continue
        x0 = np.tile(cfg["x0"], (cfg["D"], 1))
        eng.uploadState(x0)
        eng.optimize(20)
        tot, roll = eng.timeIterations(n)
        print("%-10s K=%d T=%d shape=(%d,%d,%d) variant=%d: iteration %.1f us, rollout kernel %.1f us" %
              (name, cfg["K"], cfg["T"], bx, by, cfg["D"], variant, tot / n * 1e3, roll / n * 1e3), flush=True)
        eng.close()


which = sys.argv[1:] or ["cartpole", "autorally", "di", "lstm", "racer"]
if "cartpole" in which:
    run("cartpole", cartpole_cfg(K=16384, T=100), [(64, 1, 1), (64, 1, 2), (32, 1)])
    run("cartpole", cartpole_cfg(K=2048, T=100), [(64, 1)])
if "autorally" in which:
    # (0, 0, v): the model's default shape — (64, 4), or (64, 8) in the eight-lanes-per-rollout A/B build
    run("autorally", autorally_cfg(K=16384, T=150, lambda_=1.0), [(0, 0, 1), (0, 0, 2), (32, 4), (32, 8)], n=30)
if "di" in which:
    run("di-tube", di_cfg(K=8192, T=150, tube=True), [(64, 1, 1), (0, 0, 0), (32, 1, 2)])
if "lstm" in which:
    cfg = bicycle_lstm_cfg(K=65536, T=200, lambda_=1.0)
    run("lstm", cfg, [(0, 0, 1), (0, 0, 2)], n=10)
    cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
    run("lstm+colored", cfg, [(0, 0, 1), (0, 0, 2)], n=10)
    cfg = cartpole_cfg(K=16384, T=100)
    cfg["colored"] = ([1.0], 0.97, 0.0)
    run("cartpole+colored", cfg, [(64, 1, 1), (64, 1, 2)], n=50)
if "racer" in which:
    from common import racer_cfg  # noqa: E402
    from test_racer_dubins_elevation import elevation_cfg  # noqa: E402
    from test_racer_dubins_lstm_steering import steering_cfg  # noqa: E402
    run("racer", racer_cfg(K=16384, T=100), [(64, 1, 1), (64, 1, 2)], n=50)
    run("racer-elev", elevation_cfg(K=16384, T=100), [(64, 1, 1), (64, 1, 2), (64, 4, 1), (64, 4, 2)], n=50)
    run("racer-flat", elevation_cfg(K=16384, T=100, with_map=False), [(64, 1, 2)], n=50)
    run("racer-lstm", steering_cfg(K=16384, T=100), [(64, 1, 1), (64, 4, 1), (64, 4, 2)], n=50)
    from test_racer_dubins_suspension import suspension_cfg  # noqa: E402
    run("racer-suspension", suspension_cfg(K=16384, T=100), [(64, 1, 1), (64, 4, 1), (64, 4, 2)], n=30)
    from test_racer_dubins_lstm_unc import uncertainty_cfg  # noqa: E402
    run("racer-uncertainty", uncertainty_cfg(K=16384, T=100), [(64, 1, 1), (64, 4, 1), (64, 4, 2)], n=30)
    cfg = steering_cfg(K=16384, T=100)
    cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
    run("racer-lstm+colored", cfg, [(0, 0, 0)], n=30)
