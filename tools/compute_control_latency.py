#!/usr/bin/env python3
"""Wall-clock latency of one mppi_compute_control call (host hand-over, rollout + merge, finalize kernel, results back):
the PCIe-inclusive figure a control loop sees, as opposed to the device-resident iteration bench.py reports."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from common import cartpole_cfg, make_engine  # noqa: E402

IDLE_ONLY = len(sys.argv) > 1 and sys.argv[1] == "idle"  # tools/compute_control_trace.sh: K = 16384, idle-stream calls only
for K in ((16384,) if IDLE_ONLY else (2048, 16384)):
    cfg = cartpole_cfg(K=K, T=100)
    eng = make_engine(cfg)
    x = cfg["x0"].copy()
    for _ in range(50):
        eng.computeControl(x, 1)
    n = 300 if IDLE_ONLY else 1000
    t0 = time.perf_counter()
    for _ in range(n):
        eng.computeControl(x, 1)
    t1 = time.perf_counter()
    for _ in range(n):
        eng.computeControl(x, 1)
        eng.getControlSeq()
        eng.slideControlSequence(1)
    t2 = time.perf_counter()
    # time until the CONTROL is back, from an idle stream (the state trajectory of the previous call has landed)
    acc = 0.0
    for _ in range(n):
        eng.getTargetStateSeq()
        ta = time.perf_counter()
        eng.computeControl(x, 1)
        acc += time.perf_counter() - ta
    # ... and until control AND state trajectory are back
    t3 = time.perf_counter()
    for _ in range(n):
        eng.computeControl(x, 1)
        eng.getTargetStateSeq()
    t4 = time.perf_counter()
    print("K=%d: computeControl back to back %.1f us; + getControlSeq + slide %.1f us; control ready (idle stream) %.1f us; "
          "control + state trajectory %.1f us" % (K, (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6, acc / n * 1e6,
                                                   (t4 - t3) / n * 1e6))
