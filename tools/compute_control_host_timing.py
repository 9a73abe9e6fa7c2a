#!/usr/bin/env python3
"""Where the HOST spends a mppi_compute_control call (Cartpole K=16384, T=100, idle stream): the stamps the library takes inside
the call (mppi_debug_host_stamps) — inputs written, ingest enqueued, rollout enqueued, merge enqueued, finalize enqueued, flag 0
seen, results copied — median over many calls.  Together with tools/compute_control_trace.sh (the device side of the same calls)
this is the account of the ~42 us a control loop sees."""
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import mppi_generic_amd as m  # noqa: E402
from common import cartpole_cfg, make_engine  # noqa: E402

lib = m.load_library()
cfg = cartpole_cfg(K=16384, T=100)
eng = make_engine(cfg)
x = cfg["x0"].copy()
for _ in range(100):
    eng.computeControl(x, 1)
rows = []
buf = (C.c_double * 8)()
for _ in range(1000):
    eng.getTargetStateSeq()  # idle stream
    eng.computeControl(x, 1)
    assert lib.mppi_debug_host_stamps(eng._h, buf) == 0
    rows.append(list(buf)[:7])
a = np.median(np.asarray(rows), axis=0)
names = ["inputs_written", "ingest_enqueued", "rollout_enqueued", "merge_enqueued", "finalize_enqueued", "flag0_seen", "results_copied"]
out = {n: round(float(v), 2) for n, v in zip(names, a)}
out["unit"] = "us since the call's first statement, median of 1000 idle-stream calls"
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "compute_control_host_timing.json"), "w"), indent=1)
