import os, sys, time
REPO=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (REPO, REPO+"/oracle", REPO+"/tests"): sys.path.insert(0,p)
import numpy as np
from common import autorally_cfg, bicycle_lstm_cfg, di_cfg, make_engine
for name,cfg in (("autorally", autorally_cfg(K=16384, T=150, lambda_=1.0)), ("lstm", bicycle_lstm_cfg(K=16384, T=200, lambda_=1.0)), ("di_tube", di_cfg(K=8192,T=150,tube=True))):
    eng=make_engine(cfg)
    x=cfg["x0"].copy()
    for _ in range(20): eng.computeControl(x,1)
    n=200; t0=time.perf_counter()
    for _ in range(n): eng.computeControl(x,1)
    t1=time.perf_counter()
    tot,roll=eng.timeIterations(50)
    print("%s: computeControl %.1f us, iteration %.1f us"%(name,(t1-t0)/n*1e6, tot/50*1e3))
