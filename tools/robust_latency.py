import os, sys, time
REPO=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (REPO, REPO+"/oracle", REPO+"/tests"): sys.path.insert(0,p)
import numpy as np
import mppi_generic_amd as m
from common import autorally_cfg, di_cfg
cfg = di_cfg(K=8192, T=150, tube=True, num_iters=1)
eng = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], 0.0, 1, seed=42)
eng.setDynamicsParams(cfg["dyn"]); eng.setCostParams(cfg["cost"]); eng.setSamplingParams(cfg["std_dev"], [0.3, 0.2]); eng.setRMPPIParams(25.0, 9, 32)
g = np.random.default_rng(5).uniform(-0.3, 0.3, (cfg["T"], 4, 2)).astype(np.float32)
x = cfg["x0"].copy()
def step():
    eng.updateImportanceSamplingControl(x, 1)
    eng.setFeedbackGains(g)
    eng.computeControl(x, 1)
for _ in range(20): step()
n=200; t0=time.perf_counter()
for _ in range(n): step()
t1=time.perf_counter()
for _ in range(n): eng.computeControl(x, 1)
t2=time.perf_counter()
def idle_latency(n):
    """latency of ONE computeControl on an idle device (the caller then has the control sequence; the state trajectories
    are re-rolled behind the hand-over and fetched by the next call that needs them)"""
    tot = 0.0
    for _ in range(n):
        eng.synchronize(); time.sleep(0.001)
        a = time.perf_counter(); eng.computeControl(x, 1); tot += time.perf_counter() - a
    return tot / n * 1e6
t3 = idle_latency(50)
print("robust DI K=8192 T=150: updateIS+gains+computeControl %.1f us; computeControl back to back %.1f us, on an idle device %.1f us"%((t1-t0)/n*1e6,(t2-t1)/n*1e6,t3))

# AutoRally-NN under Robust MPPI (one lane per rollout and system, LDS forward)
cfg = autorally_cfg(K=16384, T=150, lambda_=1.0)
eng = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], 0.0, 1, seed=42)
eng.setCostParams(cfg["cost"])
for name, blob in cfg["blobs"].items():
    eng.setModelBlob(name, blob)
eng.setControlRanges(cfg["ranges"]); eng.setSamplingParams(cfg["std_dev"], [0.2, 0.1]); eng.setRMPPIParams(500.0, 9, 32)
g = np.random.default_rng(5).uniform(-0.3, 0.3, (cfg["T"], 7, 2)).astype(np.float32)
x = cfg["x0"].copy()
for _ in range(5): step()
n=50; t0=time.perf_counter()
for _ in range(n): step()
t1=time.perf_counter()
for _ in range(n): eng.computeControl(x, 1)
t2=time.perf_counter()
t3 = idle_latency(50)
print("robust AutoRally-NN K=16384 T=150: updateIS+gains+computeControl %.1f us; computeControl back to back %.1f us, on an idle device %.1f us"%((t1-t0)/n*1e6,(t2-t1)/n*1e6,t3))
