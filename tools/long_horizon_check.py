import sys
sys.path[:0] = ["/root/repo", "/root/repo/oracle", "/root/repo/tests"]
import numpy as np
from common import autorally_cfg, cartpole_cfg, make_engine
for name, cfg in (("ar T=400", autorally_cfg(K=256, T=400)), ("ar T=1000", autorally_cfg(K=256, T=1000)), ("cartpole T=700", cartpole_cfg(K=256, T=700)), ("cartpole T=1500", cartpole_cfg(K=256, T=1500))):
    try:
        eng = make_engine(cfg)
        eng.computeControl(cfg["x0"], 1)
        print(name, "ok", np.isfinite(eng.getControlSeq()).all())
    except Exception as e:
        print(name, "error:", str(e)[:160])
from common import di_cfg, racer_cfg
for name, cfg in (("di T=800", di_cfg(K=256, T=800, tube=False)), ("di tube T=500", di_cfg(K=256, T=500, tube=True)), ("racer T=600", racer_cfg(K=256, T=600)), ("cartpole tube T=1000", dict(cartpole_cfg(K=256, T=1000), D=2))):
    try:
        eng = make_engine(cfg)
        eng.computeControl(cfg["x0"], 1)
        print(name, "ok", np.isfinite(eng.getControlSeq()).all())
    except Exception as e:
        print(name, "error:", str(e)[:160])
