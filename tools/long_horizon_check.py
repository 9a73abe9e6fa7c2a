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
