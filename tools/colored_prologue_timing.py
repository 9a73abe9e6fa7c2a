import sys, os
REPO=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (REPO, REPO+"/oracle", REPO+"/tests"): sys.path.insert(0,p)
import numpy as np
from common import bicycle_lstm_cfg, make_engine
for colored in (False, True):
    cfg = bicycle_lstm_cfg(K=65536, T=200, lambda_=1.0)
    if colored: cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
    eng = make_engine(cfg, block_x=64, block_y=4, kernel_variant=2)
    eng.uploadState(cfg["x0"]); eng.optimize(3)
    tot, roll = eng.timeIterations(10)
    print("lstm colored=%s: iteration %.1f us" % (colored, tot/10*1e3), flush=True)
    eng.close()
