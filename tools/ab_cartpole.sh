#!/bin/bash
# A/B of cartpole.hip variants in ONE session (MPPI_AMD_LIB selects the library): rollout kernel / iteration, event-timed
for lib in "" cp_generic cp_generic_noslp; do
  if [ -n "$lib" ]; then export MPPI_AMD_LIB=$PWD/mppi-generic_amd/lib/libmppi_amd_$lib.so; else unset MPPI_AMD_LIB; fi
  timeout 60 python - <<PY
import sys
sys.path[:0] = ["$PWD", "$PWD/tests", "$PWD/oracle"]
import numpy as np
from common import cartpole_cfg, make_engine
cfg = cartpole_cfg(K=16384, T=100)
e = make_engine(cfg); e.uploadState(cfg["x0"]); e.optimize(200)
best = (1e9, 1e9)
for r in range(5):
    tot, roll = e.timeIterations(400)
    best = min(best, (tot / 400 * 1e3, roll / 400 * 1e3))
print("variant %-18s iteration %.3f us  rollout kernel %.3f us" % ("${lib:-product}", best[0], best[1]))
PY
done
