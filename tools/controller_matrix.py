#!/usr/bin/env python3
"""Every controller kind x every registered model at benchmark-like sizes: creation, one computeControl, finite results.
A configuration sweep (block-shape selection, LDS budgets, blob plumbing), not a parity test — those are in tests/."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import mppi_generic_amd as m  # noqa: E402
from common import autorally_cfg, bicycle_lstm_cfg, cartpole_cfg, di_cfg, make_engine, racer_cfg  # noqa: E402
from test_racer_dubins_elevation import elevation_cfg as _elev  # noqa: E402
from test_racer_dubins_lstm_steering import steering_cfg as _steer  # noqa: E402
from test_racer_dubins_suspension import suspension_cfg as _susp  # noqa: E402
from test_racer_dubins_lstm_unc import uncertainty_cfg as _unc  # noqa: E402

MODELS = {
    "cartpole": lambda: cartpole_cfg(K=16384, T=100),
    "double_integrator": lambda: di_cfg(K=8192, T=150, tube=False),
    "autorally_nn": lambda: autorally_cfg(K=16384, T=150, lambda_=1.0),
    "bicycle_slip_lstm": lambda: bicycle_lstm_cfg(K=16384, T=200, lambda_=1.0),
    "racer_dubins": lambda: racer_cfg(K=16384, T=100),
    "racer_dubins_elevation": lambda: _elev(K=16384, T=100),
    "racer_dubins_elevation_lstm_steering": lambda: _steer(K=16384, T=100),
    "racer_dubins_elevation_suspension": lambda: _susp(K=16384, T=100),
    "racer_dubins_elevation_lstm_unc": lambda: _unc(K=16384, T=100),
}
ROBUST_MODELS = ("cartpole", "double_integrator", "autorally_nn", "bicycle_slip_lstm", "racer_dubins", "racer_dubins_elevation", "racer_dubins_elevation_lstm_steering",
                 "racer_dubins_elevation_suspension", "racer_dubins_elevation_lstm_unc")
bad = 0
for name, mk in MODELS.items():
    for kind in ("vanilla", "tube", "colored", "robust"):
        if kind == "robust" and name not in ROBUST_MODELS:
            continue   # the elevation-map models are registered for Vanilla, Tube and Colored MPPI
        cfg = mk()
        S = len(cfg["x0"])
        try:
            if kind == "robust":
                eng = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], 0.0, 1, seed=42)
                if cfg["dyn"] is not None:
                    eng.setDynamicsParams(cfg["dyn"])
                eng.setCostParams(cfg["cost"])
                for k, v in cfg.get("blobs", {}).items():
                    eng.setModelBlob(k, v)
                if cfg["ranges"] is not None:
                    eng.setControlRanges(cfg["ranges"])
                eng.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"])
                eng.setRMPPIParams(1000.0, 9, 32)
                C = eng.CONTROL_DIM
                eng.setFeedbackGains(np.zeros((cfg["T"], S, C), np.float32))
                eng.updateImportanceSamplingControl(cfg["x0"], 1)
            else:
                cfg["D"] = 2 if kind == "tube" else 1
                if kind == "colored":
                    cfg["colored"] = ([1.0] * len(cfg["std_dev"]), 0.97, 0.0)
                eng = make_engine(cfg)
            t0 = time.perf_counter()
            eng.computeControl(cfg["x0"], 1)
            eng.computeControl(cfg["x0"], 1)
            dt = (time.perf_counter() - t0) / 2
            ok = np.isfinite(eng.getControlSeq()).all() and np.isfinite(eng.getTargetStateSeq()).all()
            print("%-18s %-8s %s  computeControl %.0f us" % (name, kind, "ok" if ok else "NON-FINITE", dt * 1e6), flush=True)
            bad += 0 if ok else 1
            eng.close()
        except Exception as e:  # noqa: BLE001
            print("%-18s %-8s ERROR %s" % (name, kind, str(e)[:150]), flush=True)
            bad += 1
print("controller matrix:", "all ok" if bad == 0 else "%d problems" % bad)
sys.exit(1 if bad else 0)
