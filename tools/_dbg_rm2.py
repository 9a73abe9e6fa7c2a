import os, sys
REPO=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (REPO, REPO+"/oracle", REPO+"/tests"): sys.path.insert(0,p)
import numpy as np
import mppi_generic_amd as m
from common import host_noise, ulp_diff
import test_rmppi as tr
model="suspension"
cfg = tr._rm_cfg(model, K=320, T=2)
K,T=320,2
out=[]
for i in range(28):
    res={}
    for variant in (m.MPPI_KERNEL_FUSED, m.MPPI_KERNEL_PIPELINE):
        cost = m.QuadraticCostParams28()
        coeffs, goal = [0.0]*28, [0.0]*28
        coeffs[i]=1.0
        cost.s_coeffs[:] = coeffs; cost.s_goal[:] = goal
        cfg["cost"]=cost
        eng, orc, rob = tr._make_pair(cfg, thr=40.0, save_samples=True, kernel_variant=variant)
        S, C = eng.STATE_DIM, eng.CONTROL_DIM
        g = tr._gains(T, S, C)
        eng.setFeedbackGains(g, False)
        mean = (0.3 * np.sin(np.arange(T * C, dtype=np.float32) * 0.2)).reshape(T, C)
        eng.updateImportanceSampler(mean)
        eps = host_noise(1, K, T, C)[0]
        eng.injectNoise(eps)
        dx = np.zeros(S, np.float32)
        dx[:7] = np.array([0.3, -0.2, 0.1, 0.05, 0.02, 0.01, 0.0], np.float32)
        x0 = np.stack([cfg["x0"], cfg["x0"] + dx])
        res[variant]=eng.rolloutCosts(x0, 2).copy()
        eng.close()
    a,b=res[m.MPPI_KERNEL_FUSED],res[m.MPPI_KERNEL_PIPELINE]
    d=ulp_diff(a,b)
    print("output",i,"differ",(d!=0).sum(axis=1),"max ulp",d.max(), a[0][:2], b[0][:2])
