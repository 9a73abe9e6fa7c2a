#!/usr/bin/env python3
"""Soak run of the pipelined rollout kernels (progress-counter protocols between role waves): many launches at odd
sizes, any hang shows up as the caller's timeout.  Usage: timeout 600 python tools/soak.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from common import autorally_cfg, bicycle_lstm_cfg, cartpole_cfg, di_cfg, make_engine  # noqa: E402

rng = np.random.default_rng(0)
total = 0
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    K = int(rng.integers(65, 20000))
    T = int(rng.integers(1, 160))
    for name, cfg, kw in (("cartpole", cartpole_cfg(K=K, T=T), {}),
                          ("di_tube", di_cfg(K=max(64, K // 2), T=T, tube=True), {}),
                          ("autorally", autorally_cfg(K=K, T=T, lambda_=1.0), dict(block_x=64, block_y=4, kernel_variant=2)),
                          ("lstm", bicycle_lstm_cfg(K=K, T=T, lambda_=1.0), dict(block_x=64, block_y=4, kernel_variant=2))):
        eng = make_engine(cfg, **kw)
        x0 = np.tile(cfg["x0"], (cfg["D"], 1))
        eng.uploadState(x0)
        n = 300 if name in ("cartpole", "di_tube") else 60
        eng.optimize(n)
        u = eng.getOptimalControlSeq()
        assert np.isfinite(u).all(), (name, K, T)
        total += n
        eng.close()
    print("round %d ok (K=%d, T=%d)" % (rnd, K, T), flush=True)
print("soak ok:", total, "iterations")
