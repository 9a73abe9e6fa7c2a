import os, sys
REPO=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (REPO, REPO+"/oracle", REPO+"/tests"): sys.path.insert(0,p)
import numpy as np
import mppi_generic_amd as m
import pyoracle as po
from common import host_noise, ulp_diff
import test_rmppi as tr
model, acc_all, mode = sys.argv[1], sys.argv[2]=="1", sys.argv[3]
cfg = tr._rm_cfg(model, K=int(os.environ.get("DBG_K","1000")), T=int(os.environ.get("DBG_T","37")))
res={}
for variant in (m.MPPI_KERNEL_FUSED, m.MPPI_KERNEL_PIPELINE):
    eng, orc, rob = tr._make_pair(cfg, thr=40.0, save_samples=True, kernel_variant=variant)
    S, C, T, K = eng.STATE_DIM, eng.CONTROL_DIM, cfg["T"], cfg["K"]
    g = tr._gains(T, S, C)
    eng.setFeedbackGains(g, acc_all)
    mean = (0.3 * np.sin(np.arange(T * C, dtype=np.float32) * 0.2)).reshape(T, C)
    eng.updateImportanceSampler(mean)
    if mode == "injected":
        eps = host_noise(1, K, T, C)[0]
        eng.injectNoise(eps)
    dx = np.zeros(S, np.float32)
    dx[:min(S, 7)] = np.array([0.3, -0.2, 0.1, 0.05, 0.02, 0.01, 0.0], np.float32)[:S]
    x0 = np.stack([cfg["x0"], cfg["x0"] + dx])
    res[variant]=(eng.rolloutCosts(x0, 2).copy(), eng.getSampledControls().copy())
    eng.close()
a,b=res[m.MPPI_KERNEL_FUSED],res[m.MPPI_KERNEL_PIPELINE]
d=ulp_diff(a[0],b[0])
print("S",S,"costs differ:", (d!=0).sum(axis=1), "max ulp", d.max())
for z in range(2):
    idx=np.nonzero(d[z])[0]
    print("sys",z,"first idx",idx[:40], "n",len(idx))
    if len(idx): print(a[0][z][idx[:5]], b[0][z][idx[:5]])
du=ulp_diff(a[1],b[1])
print("controls differ:", [(du[z]!=0).any(axis=(1,2)).sum() for z in range(2)])
for z in range(2):
    k=np.nonzero((du[z]!=0).any(axis=(1,2)))[0]
    if len(k):
        kk=k[0]; tt=np.nonzero((du[z][kk]!=0).any(axis=1))[0]
        print("sys",z,"rollout",kk,"first t",tt[:10], a[1][z][kk][tt[0]], b[1][z][kk][tt[0]])
