#!/usr/bin/env python3
"""Rollout-kernel and iteration times (HIP events on the engine's stream, best of several passes) of the benchmarked
workloads for the library MPPI_AMD_LIB selects — the A/B companion of `buildlib.py --variant`.
Usage: [MPPI_AMD_LIB=.../libmppi_amd_<tag>.so] python tools/ab_kernels.py [cartpole autorally ditube lstm racer robust_ar robust_lstm robust_di robust_racer robust_racer_all] [--json out]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import mppi_generic_amd as m  # noqa: E402
from common import autorally_cfg, bicycle_lstm_cfg, cartpole_cfg, di_cfg, make_engine  # noqa: E402


def vanilla(cfg, n, passes=5, **kw):
    eng = make_engine(cfg, **kw)
    eng.uploadState(np.tile(cfg["x0"], (cfg["D"], 1)))
    eng.optimize(max(20, n // 4))
    best = (1e9, 1e9)
    for _ in range(passes):
        tot, roll = eng.timeIterations(n)
        best = min(best, (tot / n * 1e3, roll / n * 1e3))
    eng.close()
    return {"iteration_us": round(best[0], 3), "rollout_kernel_us": round(best[1], 3)}


def robust(cfg, thr, n, passes=4):
    eng = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], 0.0, 1, seed=42)
    if cfg["dyn"] is not None:
        eng.setDynamicsParams(cfg["dyn"])
    eng.setCostParams(cfg["cost"])
    for name, blob in cfg.get("blobs", {}).items():
        eng.setModelBlob(name, blob)
    if cfg["ranges"] is not None:
        eng.setControlRanges(cfg["ranges"])
    eng.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"])
    eng.setRMPPIParams(thr, 9, 32)
    g = np.random.default_rng(5).uniform(-0.3, 0.3, (cfg["T"], eng.STATE_DIM, eng.CONTROL_DIM)).astype(np.float32)
    x = cfg["x0"].copy()
    for _ in range(5):
        eng.updateImportanceSamplingControl(x, 1)
        eng.setFeedbackGains(g)
        eng.computeControl(x, 1)
    best = 1e9
    for _ in range(passes):
        _, roll = eng.timeIterations(n)
        best = min(best, roll / n * 1e3)
    eng.close()
    return {"rollout_kernel_us": round(best, 3)}


def main():
    args = [a for a in sys.argv[1:]]
    out_json = None
    if "--json" in args:
        i = args.index("--json")
        out_json = args[i + 1]
        del args[i:i + 2]
    which = args or ["cartpole", "autorally", "ditube", "lstm", "racer", "robust_ar", "robust_di", "robust_racer"]
    res = {"library": m.library_path()}
    if "cartpole" in which:
        res["cartpole_16384x100"] = vanilla(cartpole_cfg(K=16384, T=100), 400)
    if "autorally" in which:
        res["autorally_16384x150"] = vanilla(autorally_cfg(K=16384, T=150, lambda_=1.0), 60)
    if "ditube" in which:
        res["di_tube_8192x150"] = vanilla(di_cfg(K=8192, T=150, tube=True), 200)
    if "lstm" in which:
        cfg = bicycle_lstm_cfg(K=65536, T=200, lambda_=1.0)
        cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
        res["lstm_colored_65536x200"] = vanilla(cfg, 10, passes=3)
    if "racer" in which:
        from test_racer_dubins_elevation import elevation_cfg
        from test_racer_dubins_lstm_steering import steering_cfg
        from test_racer_dubins_lstm_unc import uncertainty_cfg
        from test_racer_dubins_suspension import suspension_cfg
        res["racer_elevation_16384x100"] = vanilla(elevation_cfg(K=16384, T=100), 50, passes=3)
        res["racer_lstm_steering_16384x100"] = vanilla(steering_cfg(K=16384, T=100), 40, passes=3)
        res["racer_suspension_16384x100"] = vanilla(suspension_cfg(K=16384, T=100), 40, passes=3)
        res["racer_complete_16384x100"] = vanilla(uncertainty_cfg(K=16384, T=100), 30, passes=3)
    if "robust_ar" in which:
        cfg = autorally_cfg(K=16384, T=150, lambda_=1.0)
        cfg["D"] = 2
        cfg["control_cost_coeff"] = [0.2, 0.1]
        res["robust_autorally_16384x150"] = robust(cfg, 500.0, 20)
    if "robust_lstm" in which:
        cfg = bicycle_lstm_cfg(K=16384, T=150, lambda_=1.0)
        cfg["D"] = 2
        cfg["control_cost_coeff"] = [0.2, 0.1]
        res["robust_bicycle_lstm_16384x150"] = robust(cfg, 500.0, 10, passes=3)
    if "robust_di" in which:
        cfg = di_cfg(K=8192, T=150, tube=True)
        cfg["control_cost_coeff"] = [0.3, 0.2]
        cfg["ranges"] = [[-3.0, 3.0], [-3.0, 3.0]]
        res["robust_di_8192x150"] = robust(cfg, 25.0, 50)
    if "robust_racer_all" in which:
        from test_racer_dubins_elevation import elevation_cfg
        from test_racer_dubins_lstm_steering import steering_cfg
        from test_racer_dubins_suspension import suspension_cfg
        for name, mk in (("elevation", elevation_cfg), ("lstm_steering", steering_cfg), ("suspension", suspension_cfg)):
            cfg = mk(K=16384, T=100, D=2)
            cfg["control_cost_coeff"] = [0.2, 0.1]
            res["robust_racer_%s_16384x100" % name] = robust(cfg, 2000.0, 10, passes=3)
    if "robust_racer" in which:
        from test_racer_dubins_lstm_unc import uncertainty_cfg
        cfg = uncertainty_cfg(K=16384, T=100, D=2)
        cfg["control_cost_coeff"] = [0.2, 0.1]
        res["robust_racer_complete_16384x100"] = robust(cfg, 2000.0, 10, passes=3)
    for k, v in res.items():
        print("%-36s %s" % (k, v), flush=True)
    if out_json:
        with open(out_json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
