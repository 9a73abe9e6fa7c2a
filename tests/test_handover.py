"""The two hand-overs of mppi_compute_control give the same results: device-mapped host memory + flags (default: the call
returns when the control sequence is out, the state trajectories follow behind it) and pinned copies + one stream
synchronisation (MPPI_AMD_NO_SPIN=1, read when the handle is created).  Vanilla and Robust MPPI."""
import os

import numpy as np
import pytest

import mppi_generic_amd as m
from common import cartpole_cfg, make_engine
import test_rmppi as tr

pytestmark = pytest.mark.gpu


def _with_env(value, make):
    old = os.environ.get("MPPI_AMD_NO_SPIN")
    if value is None:
        os.environ.pop("MPPI_AMD_NO_SPIN", None)
    else:
        os.environ["MPPI_AMD_NO_SPIN"] = value
    try:
        return make()
    finally:
        if old is None:
            os.environ.pop("MPPI_AMD_NO_SPIN", None)
        else:
            os.environ["MPPI_AMD_NO_SPIN"] = old


def test_vanilla_handovers_agree(gpu):
    cfg = cartpole_cfg(K=2048, T=60, soft=True)
    out = []
    for env in (None, "1"):
        eng = _with_env(env, lambda: make_engine(cfg))
        x = cfg["x0"].copy()
        seqs = []
        for i in range(4):
            eng.computeControl(x, 1)
            u = eng.getControlSeq().copy()
            if i % 2 == 0:  # every other call leaves the trajectories unread: the next call has to wait for the kernel itself
                seqs.append((u, eng.getTargetStateSeq().copy(), eng.getTargetOutputSeq().copy()))
            else:
                seqs.append((u,))
            eng.slideControlSequence(1)
        out.append(seqs)
        eng.close()
    for a, b in zip(*out):
        for p, q in zip(a, b):
            assert np.array_equal(p, q)


@pytest.mark.parametrize("num_iters", [1, 3])
def test_tube_handovers_agree(gpu, num_iters):
    """Tube MPPI: every optimisation pass needs both trajectories on the host (nominal <- actual when the actual system is the
    better one), the final smoothing pass hands the control sequences over first"""
    from common import di_cfg
    cfg = di_cfg(K=1024, T=60, tube=True, num_iters=num_iters)
    out = []
    for env in (None, "1"):
        eng = _with_env(env, lambda: make_engine(cfg))
        x = cfg["x0"].copy()
        rec = []
        for i in range(5):
            eng.computeControl(x, 1)
            st = eng.getStats()
            r = [eng.getControlSeq().copy(), eng.getNominalControlSeq().copy(),
                 np.array([st.real_sys.baseline, st.nominal_sys.baseline, st.nominal_state_used], np.float32)]
            if i % 2 == 0:
                r += [eng.getTargetStateSeq().copy(), eng.getNominalStateSeq().copy()]
            rec.append(r)
            eng.slideControlSequence(1)
            x = x + np.float32(0.05 * (i + 1))  # push the actual system away from the nominal one
        out.append(rec)
        eng.close()
    for a, b in zip(*out):
        assert len(a) == len(b)
        for p, q in zip(a, b):
            assert np.array_equal(p, q)
    used = {int(r[2][2]) for r in out[0]}
    assert np.isfinite(out[0][-1][0]).all() and len(used) >= 1


@pytest.mark.parametrize("model", ["di", "autorally"])
def test_robust_handovers_agree(gpu, model):
    """Robust MPPI: both control sequences and the statistics come back with the first flag of each system; the nominal
    state trajectory — the next updateImportanceSamplingControl builds its candidates from it — behind the second"""
    cfg = tr._rm_cfg(model, K=1024, T=40, num_iters=2)
    out = []
    for env in (None, "1"):
        eng, orc, rob = _with_env(env, lambda: tr._make_pair(cfg, thr={"di": 25.0}.get(model, 500.0)))
        S, C, T = eng.STATE_DIM, eng.CONTROL_DIM, cfg["T"]
        x = cfg["x0"].copy()
        rec = []
        for i in range(4):
            eng.updateImportanceSamplingControl(x, 1)
            eng.setFeedbackGains(tr._gains(T, S, C, seed=10 + i, scale=0.3))
            eng.computeControl(x, 1)
            st = eng.getStats()
            r = [eng.getControlSeq().copy(), eng.getNominalControlSeq().copy(),
                 np.array([st.real_sys.baseline, st.nominal_sys.baseline, st.real_sys.normalizer], np.float32),
                 np.array(eng.getRMPPIState()[1:3], np.float32)]
            if i % 2 == 0:
                r += [eng.getTargetStateSeq().copy(), eng.getNominalStateSeq().copy(), eng.getTargetOutputSeq().copy()]
            rec.append(r)
            x = x + np.float32(0.01)
        out.append(rec)
        eng.close()
    for a, b in zip(*out):
        assert len(a) == len(b)
        for p, q in zip(a, b):
            assert np.array_equal(p, q)
    assert np.isfinite(out[0][-1][0]).all()
