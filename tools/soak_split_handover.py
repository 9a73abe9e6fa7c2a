#!/usr/bin/env python3
"""Soak of the split hand-over (engine_controllers.hip: split_finalize): long closed loops with the entry points a plant mixes in at random
— trajectory reads, statistics, model steps, parameter updates, optimisation calls — run once with the split and once with the
single launch; every host-visible result must be the same bits, and nothing may hang (run under `timeout`).
Usage: timeout 600 python tools/soak_split_handover.py [cycles (20000)]"""
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from common import autorally_cfg, cartpole_cfg, make_engine  # noqa: E402


def run(cfg, split, cycles, seed):
    if split:
        os.environ.pop("MPPI_AMD_SPLIT_FINALIZE", None)
    else:
        os.environ["MPPI_AMD_SPLIT_FINALIZE"] = "0"
    eng = make_engine(cfg)
    rng = np.random.default_rng(seed)
    h = hashlib.sha256()
    x = cfg["x0"].copy()
    for i in range(cycles):
        eng.computeControl(x, 1)
        u = eng.getControlSeq()
        h.update(u.tobytes())
        r = rng.integers(0, 12)
        if r == 0:
            h.update(eng.getTargetStateSeq().tobytes())
        elif r == 1:
            h.update(eng.getTargetOutputSeq().tobytes())
        elif r == 2:
            st = eng.getStats()
            h.update(np.array([st.real_sys.baseline, st.real_sys.normalizer], np.float32).tobytes())
        elif r == 3:
            eng.optimize(1, True)
            h.update(eng.getOptimalControlSeq().tobytes())
        elif r == 4:
            eng.setNumIters(1 + int(rng.integers(0, 2)))
        un = np.ascontiguousarray(u.reshape(-1, eng.CONTROL_DIM)[0], np.float32)
        xs = x.copy()
        eng.modelStep(xs, un)
        if np.all(np.isfinite(xs)) and np.max(np.abs(xs)) < 1e3:
            x = xs
        else:
            x = cfg["x0"].copy()
        eng.slideControlSequence(1)
    h.update(eng.getTargetStateSeq().tobytes())
    eng.close()
    return h.hexdigest()


cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ok = True
for name, cfg, n in (("cartpole", cartpole_cfg(K=2048, T=100, soft=True), cycles),
                     ("autorally", autorally_cfg(K=1024, T=60), cycles // 10)):
    a = run(cfg, True, n, 7)
    b = run(cfg, False, n, 7)
    print(name, n, "cycles:", "same" if a == b else "DIFFERENT", a[:16], b[:16], flush=True)
    ok = ok and a == b
sys.exit(0 if ok else 1)
