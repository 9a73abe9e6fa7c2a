#!/usr/bin/env python3
"""Long horizons: the LDS-row variant with small blocks against the rows-in-HBM variant of the fused rollout kernel, and the
achieved HBM rate of the latter (algorithmic bytes B_alg = 4 (2 K T C + 2 K + 2 T C), SURVEY.md §8d — here the sample tensor
really is written once and read once)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from common import cartpole_cfg, autorally_cfg, make_engine  # noqa: E402


def run(label, cfg, hbm, n=20, **kw):
    if hbm:
        os.environ["MPPI_AMD_ROWS_IN_HBM"] = "1"
    try:
        eng = make_engine(cfg, **kw)
    except Exception as e:  # noqa: BLE001
        print("%-28s %s" % (label, str(e)[:100]))
        return
    finally:
        os.environ.pop("MPPI_AMD_ROWS_IN_HBM", None)
    eng.uploadState(np.tile(cfg["x0"], (cfg["D"], 1)))
    eng.optimize(3)
    tot, roll = eng.timeIterations(n)
    C = eng.CONTROL_DIM
    b_alg = 4.0 * (2.0 * cfg["K"] * cfg["T"] * C + 2.0 * cfg["K"] + 2.0 * cfg["T"] * C)
    us = roll / n * 1e3
    print("%-28s K=%d T=%d: rollout kernel %9.1f us, iteration %9.1f us, B_alg %.1f MB -> %.0f GB/s" % (
        label, cfg["K"], cfg["T"], us, tot / n * 1e3, b_alg / 1e6, b_alg / (us * 1e-6) / 1e9))
    eng.close()


for T in (1500, 5000):
    cfg = cartpole_cfg(K=16384, T=T)
    run("cartpole LDS rows (auto)", cfg, False)
    run("cartpole HBM rows (64,1)", cfg, True, kernel_variant=1)
cfg = cartpole_cfg(K=65536, T=5000)
run("cartpole HBM rows (64,1)", cfg, True, kernel_variant=1, n=5)
cfg = autorally_cfg(K=16384, T=1000)
run("autorally LDS rows (auto)", cfg, False, n=5)
run("autorally HBM rows (64,4)", cfg, True, kernel_variant=1, n=5)
