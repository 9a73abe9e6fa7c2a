import os, sys
REPO=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (REPO, REPO+"/oracle", REPO+"/tests"): sys.path.insert(0,p)
import numpy as np, psutil
from common import cartpole_cfg, di_cfg, autorally_cfg, make_engine
proc=psutil.Process()
for rnd in range(3):
    for i in range(60):
        for cfg in (cartpole_cfg(K=2048,T=50), di_cfg(K=1024,T=40,tube=True)):
            e=make_engine(cfg); e.computeControl(cfg["x0"],1); e.modelStep(cfg["x0"], np.zeros(e.CONTROL_DIM,np.float32)); e.close()
    e=make_engine(autorally_cfg(K=1024,T=30)); e.computeControl(autorally_cfg(K=1024,T=30)["x0"],1); e.close()
    print("round",rnd,"rss MB",proc.memory_info().rss>>20, flush=True)
print("LIFECYCLE OK")
