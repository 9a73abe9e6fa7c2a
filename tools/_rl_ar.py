import os, sys, time
REPO=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (REPO, REPO+"/oracle", REPO+"/tests"): sys.path.insert(0,p)
import numpy as np
import mppi_generic_amd as m
from common import autorally_cfg
cfg = autorally_cfg(K=16384, T=150, lambda_=1.0)
eng = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], 0.0, 1, seed=42)
eng.setCostParams(cfg["cost"])
for name, blob in cfg["blobs"].items():
    eng.setModelBlob(name, blob)
eng.setControlRanges(cfg["ranges"]); eng.setSamplingParams(cfg["std_dev"], [0.2, 0.1]); eng.setRMPPIParams(500.0, 9, 32)
g = np.random.default_rng(5).uniform(-0.3, 0.3, (cfg["T"], 7, 2)).astype(np.float32)
x = cfg["x0"].copy()
def step():
    eng.updateImportanceSamplingControl(x, 1)
    eng.setFeedbackGains(g)
    eng.computeControl(x, 1)
for _ in range(5): step()
best=1e9
for rep in range(5):
    n=40; t1=time.perf_counter()
    for _ in range(n): eng.computeControl(x, 1)
    t2=time.perf_counter()
    best=min(best,(t2-t1)/n*1e6)
print("%s: robust AutoRally-NN computeControl %.1f us"%(os.environ.get("MPPI_AMD_LIB","main"),best))
