#!/usr/bin/env python3
"""Phases of combineKernel (the merge launch of a Cartpole iteration, K=16384, T=100: 256 block records -> u*), from s_memtime
stamps inside the kernel (A/B build with -DMPPI_COMBINE_TIMING, never a product build).

  python mppi-generic_amd/buildlib.py --variant timing_merge engine_iteration.hip -DMPPI_COMBINE_TIMING                       (CPU)
  MPPI_AMD_LIB=$PWD/mppi-generic_amd/lib/libmppi_amd_timing_merge.so python tools/combine_timing.py [out.json]      (GPU box)"""
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


def main():
    lib = C.CDLL(m.library_path())
    cfg = cartpole_cfg(K=16384, T=100)
    eng = make_engine(cfg)
    eng.uploadState(cfg["x0"])
    eng.optimize(50)
    rows = []
    for rep in range(20):
        eng.optimize(3)
        buf = (C.c_ulonglong * 32)()
        assert lib.mppi_debug_read_combine_timing(buf, 32) == 32
        rows.append(np.frombuffer(buf, np.uint64).reshape(4, 8)[:2].astype(np.float64).copy())
    t = np.stack(rows)  # [rep][wave][stamp]: waves 0 / 1 of block (0, 0): column waves
    tot, roll = eng.timeIterations(200)
    tick_us = 2400.0  # s_memtime runs at the shader clock (~2.4 GHz, tools/pipe_timing_cartpole.py calibrates it live)
    d = (t[:, :, 1:7] - t[:, :, 0:6]) / tick_us
    names = ["entry -> all loads of the wave issued", "loads arrive (the one memory round trip) + min over the lane's records",
             "rho: wave all-reduce (DPP)", "four scale factors (exp) + eta / column partial sums of the lane",
             "eta, sum w^2 and the four column sums: wave all-reduces (DPP)", "division + store"]
    out = {"workload": "combineKernel, Cartpole K=16384 T=100: 256 records x 104 floats, one wave per 4 columns",
           "iteration_us": round(tot / 200 * 1e3, 2), "rollout_kernel_us": round(roll / 200 * 1e3, 2),
           "unit": "microseconds, median over 20 launches, wave 0 / wave 1",
           "phases": {n: [round(float(np.median(d[:, w, i])), 3) for w in range(2)] for i, n in enumerate(names)},
           "kernel_body_us": [round(float(np.median((t[:, w, 6] - t[:, w, 0]) / tick_us)), 3) for w in range(2)]}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
