import sys, os
REPO=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (REPO, REPO+"/oracle", REPO+"/tests"): sys.path.insert(0,p)
import numpy as np
from common import cartpole_cfg, make_engine
for variant in (2,1):
    res=[]
    for T in (20,60,100,140,200):
        cfg=cartpole_cfg(K=16384,T=T)
        eng=make_engine(cfg,kernel_variant=variant)
        eng.uploadState(cfg["x0"]); eng.optimize(20)
        tot,roll=eng.timeIterations(100)
        res.append((T,roll/100*1e3)); eng.close()
    (b,a)=np.polyfit([r[0] for r in res],[r[1] for r in res],1)
    print("variant",variant,res,"fixed %.2f us + %.1f ns/step"%(a,b*1e3))
