"""Wall clock of a Tube-MPPI mppi_compute_control (double integrator, K=8192, T=150) from an idle stream; under
rocprofv3 --kernel-trace its calls show the launches of a call (profiles/r06_compute_control_merge_control.json: tube_experiment_not_kept)"""
import os, sys, time
REPO=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (REPO, REPO+"/oracle", REPO+"/tests"): sys.path.insert(0,p)
import numpy as np
from common import di_cfg, make_engine
cfg=di_cfg(K=8192,T=150,tube=True)
eng=make_engine(cfg)
x=cfg["x0"].copy()
for _ in range(50): eng.computeControl(x,1)
n=300; acc=0.0
for _ in range(n):
    eng.getTargetStateSeq()
    t0=time.perf_counter(); eng.computeControl(x,1); acc+=time.perf_counter()-t0
print("di_tube computeControl idle stream %.1f us"%(acc/n*1e6))
