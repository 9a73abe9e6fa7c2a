#!/usr/bin/env python3
"""Iteration time against the number of rollouts (fixed T): how much of the chip the BASELINE sizes leave idle."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from common import autorally_cfg, cartpole_cfg, make_engine  # noqa: E402

for name, mk, kw, Ks in (("cartpole T=100", lambda K: cartpole_cfg(K=K, T=100), {}, (4096, 16384, 32768, 65536, 131072, 262144)),
                         ("autorally T=150", lambda K: autorally_cfg(K=K, T=150, lambda_=1.0),
                          dict(block_x=64, block_y=4, kernel_variant=2), (4096, 16384, 32768, 65536))):
    for K in Ks:
        cfg = mk(K)
        eng = make_engine(cfg, **kw)
        eng.uploadState(np.tile(cfg["x0"], (cfg["D"], 1)))
        eng.optimize(20)
        n = 100
        tot, roll = eng.timeIterations(n)
        print("%-16s K=%-7d iteration %8.1f us  rollout kernel %8.1f us  -> %.2f G rollout-steps/s" %
              (name, K, tot / n * 1e3, roll / n * 1e3, K * cfg["T"] / (tot / n * 1e-3) / 1e9), flush=True)
        eng.close()
