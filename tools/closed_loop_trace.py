#!/usr/bin/env python3
"""Closed loop of mppi_compute_control + getControlSeq + slide (Cartpole K=16384, T=100) with the split hand-over on or off
(argv[1]: split | single): period, and the library's host stamps inside the calls (median).  Under
`rocprofv3 --kernel-trace` (tools/closed_loop_trace.sh) the same loop gives the device side: which kernels overlap."""
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
mode = sys.argv[1] if len(sys.argv) > 1 else "split"
if mode == "single":
    os.environ["MPPI_AMD_SPLIT_FINALIZE"] = "0"
import numpy as np  # noqa: E402
import mppi_generic_amd as m  # noqa: E402
from common import cartpole_cfg, make_engine  # noqa: E402

lib = m.load_library()
cfg = cartpole_cfg(K=16384, T=100)
eng = make_engine(cfg)
x = cfg["x0"].copy()
for _ in range(100):
    eng.computeControl(x, 1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rows = []
buf = (C.c_double * 8)()
t0 = time.perf_counter()
for _ in range(n):
    eng.computeControl(x, 1)
    eng.getControlSeq()
    eng.slideControlSequence(1)
period = (time.perf_counter() - t0) / n * 1e6
for _ in range(n):
    eng.computeControl(x, 1)
    assert lib.mppi_debug_host_stamps(eng._h, buf) == 0
    rows.append(list(buf)[:7])
    eng.getControlSeq()
    eng.slideControlSequence(1)
a = np.median(np.asarray(rows), axis=0)
names = ["inputs_written", "ingest_enqueued", "rollout_enqueued", "merge_enqueued", "finalize_enqueued", "flag0_seen", "results_copied"]
out = {"mode": mode, "closed_loop_period_us": round(period, 2)}
out.update({nm: round(float(v), 2) for nm, v in zip(names, a)})
print(json.dumps(out))
eng.getTargetStateSeq()
eng.close()
