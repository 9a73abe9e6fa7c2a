"""DoubleIntegratorRobustCost (reference: cost_functions/double_integrator/double_integrator_robust_cost.cu:10-41, the device
overload) and the three set-ups of examples/double_integrator_CORL2020.cu that run on it: Vanilla (:228-262), Tube (:405-439)
and Robust MPPI (:602-645), each with K = 1024, T = 50, dt = 0.02, lambda = 2, sigma = 1, crash_cost = 100, x0 = (2, 0, 0, 1).

The reference holds no known-answer test for this cost (tests/cost_functions/ has none for the double integrator's robust
cost), so the oracle's restatement is pinned on values derived by hand from the source: the track term is piecewise linear
in the normalised distance from the centre line r = 2 (inner radius 1.875, outer 2.125): 0 on the centre line, half the
crash cost half way to either edge (the DEVICE constants; the host overload's 0.75 / 0.1 are the quirk SURVEY.md Appendix A
lists and DESIGN.md §7 decides), the crash cost on and beyond the edge.
"""
import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import SEED, di_cfg, host_noise, make_engine, make_oracle, ulp_diff

U_TOL = 1e-5


def robust_cfg(K=1024, T=50, tube=False, num_iters=1):
    cfg = di_cfg(K=K, T=T, tube=tube, num_iters=num_iters)
    cfg["model"] = "double_integrator_robust"
    cfg["cost"].crash_cost = 100  # double_integrator_CORL2020.cu:253, :431, :632
    return cfg


def _cost(oracle, s):
    return oracle.state_cost(s)[0]


def _track_cost64(r, crash=100.0):
    """float64 restatement written from the formulas, not from the oracle"""
    nd = abs(r - 2.0) / 0.125
    if nd <= 0.5:
        return nd / 0.5 * (0.5 * crash)
    if nd <= 1.0:
        return (nd - 0.5) / 0.5 * (crash - 0.5 * crash) + 0.5 * crash
    return crash


def test_robust_cost_hand_values():
    orc = make_oracle(robust_cfg(K=64, T=4))
    # on the centre line at the desired speed and angular momentum (|v| = 2 tangential, p x v = 4): nothing to pay
    assert _cost(orc, [2.0, 0.0, 0.0, 2.0]) == 0.0
    # half way to the outer edge: half the crash cost; on the edge and beyond: the crash cost (device constants)
    assert _cost(orc, [2.0625, 0.0, 0.0, 2.0]) == pytest.approx(50.0 + (2.0625 * 2 - 4) ** 2, rel=1e-6)
    assert _cost(orc, [2.125, 0.0, 0.0, 2.0]) == pytest.approx(100.0 + (2.125 * 2 - 4) ** 2, rel=1e-6)
    assert _cost(orc, [1.875, 0.0, 0.0, 2.0]) == pytest.approx(100.0 + (1.875 * 2 - 4) ** 2, rel=1e-6)
    assert _cost(orc, [3.0, 0.0, 0.0, 2.0]) == pytest.approx(100.0 + (3.0 * 2 - 4) ** 2, rel=1e-6)
    # a quarter of the way: the shallow branch is linear through the origin
    assert _cost(orc, [0.0, 2.03125, -2.0, 0.0]) == pytest.approx(25.0 + (2.03125 * 2 - 4) ** 2, rel=1e-6)
    # speed and angular-momentum errors enter SQUARED (the circle cost takes their absolute value)
    assert _cost(orc, [2.0, 0.0, 0.0, 1.0]) == pytest.approx(1.0 + 4.0, rel=1e-6)
    # the host overload's constants would give 0.1 * 100 * (0.5 / 0.75) = 6.67 at the half-way point, not 50
    assert abs(_cost(orc, [2.0625, 0.0, 0.0, 2.0]) - (6.6667 + 0.015625)) > 40


def test_robust_cost_against_float64_on_random_states():
    orc = make_oracle(robust_cfg(K=64, T=4))
    rng = np.random.default_rng(SEED)
    for _ in range(2000):
        r = rng.uniform(1.7, 2.3)
        th = rng.uniform(0, 2 * np.pi)
        s = np.array([r * np.cos(th), r * np.sin(th), rng.uniform(-3, 3), rng.uniform(-3, 3)], np.float32)
        s64 = s.astype(np.float64)
        want = _track_cost64(np.hypot(s64[0], s64[1])) + (np.hypot(s64[2], s64[3]) - 2.0) ** 2 + \
            (s64[0] * s64[3] - s64[1] * s64[2] - 4.0) ** 2
        got = _cost(orc, s)
        # the branch a state within one float of a knee takes may differ; the function is continuous there
        assert got == pytest.approx(want, rel=2e-5, abs=2e-4)


def test_circle_and_robust_cost_share_params_and_differ():
    a, b = make_oracle(di_cfg(K=64, T=4, tube=False)), make_oracle(robust_cfg(K=64, T=4))
    s = [2.2, 0.0, 0.0, 2.0]
    assert _cost(a, s) == pytest.approx(1000.0 + 0.4, rel=1e-6)  # circle cost: crash_cost 1000 beyond the edge
    assert _cost(b, s) == pytest.approx(100.0 + 0.16, rel=1e-5)


# ------------------------------------------------------------------ GPU: the CORL2020 set-ups ---------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1], ids=["pipeline", "fused"])
def test_vanilla_robust_cost_corl2020(gpu, variant):
    """examples/double_integrator_CORL2020.cu:228-262 (VanillaMPPIController<Dyn, RCost, ..., 50, 1024>), 20 closed-loop steps
    with the state disturbed as the example's noisy plant does"""
    cfg = robust_cfg()
    eng, orc = make_engine(cfg, kernel_variant=variant), make_oracle(cfg)
    x = cfg["x0"].copy()
    rng = np.random.default_rng(3)
    for i in range(20):
        eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=200 + i)
        eng.injectNoise(eps)
        eng.computeControl(x, 1)
        orc.vanilla_compute_control(x, 1, eps)
        u_o = orc.control().copy()
        assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0, i
        assert np.abs(eng.getControlSeq() - u_o).max() <= U_TOL, i
        assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= 1e-4
        eng.updateImportanceSampler(u_o)  # re-synchronised closed loop (tests/test_closed_loop_parity.py)
        eng.slideControlSequence(1)
        orc.vanilla_slide(1)
        x, _ = orc.model_step(x, u_o[0])
        x = x + (rng.standard_normal(4) * np.array([0, 0, 1, 1]) * np.sqrt(0.02)).astype(np.float32)
    assert (orc.costs() < 50 * 100).any()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [{}, {"kernel_variant": 1}, {"block_x": 64, "block_y": 1}],
                         ids=["folded-pipeline", "fused", "pipeline-64x1x2"])
def test_tube_robust_cost_corl2020(gpu, kw):
    """examples/double_integrator_CORL2020.cu:405-439 (TubeMPPIController<Dyn, RCost, ..., 50, 1024>)"""
    cfg = robust_cfg(tube=True)
    eng, orc = make_engine(cfg, **kw), make_oracle(cfg)
    x = cfg["x0"].copy()
    drift = []
    for i in range(4):
        eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=300 + i)
        eng.injectNoise(eps)
        eng.computeControl(x, 1)
        orc.tube_compute_control(x, 1, eps)
        du = float(max(np.abs(eng.getControlSeq() - orc.control()).max(),
                       np.abs(eng.getNominalControlSeq() - orc.nominal_control()).max()))
        drift.append(du)
        assert eng.getStats().nominal_state_used == orc.stats()["nominal_state_used"]
        if i == 0:  # one call from identical state: the parity bar
            assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0
            assert du <= U_TOL
            assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= 1e-4
        else:
            # FREE-RUNNING from the second call on (Tube keeps two control sequences and a nominal state; each carries the
            # ~1e-7 of the previous call, and this cost squares its speed / angular-momentum errors): the distance is
            # reported, and only bounded loosely as a sanity check — see tests/test_closed_loop_parity.py for the argument
            np.testing.assert_allclose(eng.getSampledCostSeq(), orc.costs(), rtol=2e-4)
            assert du <= 2e-4, drift
        x = x + np.array([0.03, -0.02, 0.2, -0.1], np.float32) * (i + 1)
    print("\ntube robust-cost free-running u* distance per call:", ["%.2g" % d for d in drift])
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["injected", "philox"])
def test_rmppi_robust_cost_corl2020(gpu, mode):
    """examples/double_integrator_CORL2020.cu:602-645 (RobustMPPIController<Dyn, RCost, ..., 50, 1024>,
    value_function_threshold 20): rollout costs of both systems bit for bit, then the closed loop of
    updateImportanceSamplingControl + computeControl"""
    cfg = robust_cfg(tube=True, num_iters=1)
    cfg["control_cost_coeff"] = [0.3, 0.2]
    K, T = cfg["K"], cfg["T"]

    def pair(**kw):
        eng = m.RobustMPPIController(cfg["model"], K, T, cfg["dt"], cfg["lambda_"], cfg["alpha"], 1, seed=SEED, **kw)
        eng.setDynamicsParams(cfg["dyn"])
        eng.setCostParams(cfg["cost"])
        eng.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"])
        eng.setRMPPIParams(20.0, 9, 32)
        orc = make_oracle(cfg)
        return eng, orc, po.RobustOracle(orc, 20.0, 9, 32)

    g = np.random.default_rng(1).uniform(-0.4, 0.4, (T, 4, 2)).astype(np.float32)
    eng, orc, rob = pair(save_samples=True)
    eng.setFeedbackGains(g, False)
    rob.set_gains(g, False)
    mean = (0.3 * np.sin(np.arange(T * 2, dtype=np.float32) * 0.2)).reshape(T, 2)
    eng.updateImportanceSampler(mean)
    if mode == "injected":
        eps = host_noise(1, K, T, 2)[0]
        eng.injectNoise(eps)
    else:
        eps = po.philox_normal(SEED, 0, K, T, 2)
    x0 = np.stack([cfg["x0"], cfg["x0"] + np.array([0.05, -0.04, 0.1, 0.05], np.float32)])
    got = eng.rolloutCosts(x0, 1)
    means = np.tile(mean, (2, 1, 1))
    v = orc.set_gaussian_controls(means, eps, 1, 0)
    want, v_fb = rob.rollout_costs(x0, means, v)
    assert ulp_diff(got, want).max() == 0
    assert ulp_diff(eng.getSampledControls(), v_fb).max() == 0

    eng.close()
    eng, orc, rob = pair()
    x = cfg["x0"].copy()
    drift = []
    for i in range(5):
        e3 = host_noise(2, K, T, 2, seed=400 + i)
        eng.injectNoise(e3[1:] if i == 0 else e3)
        eng.updateImportanceSamplingControl(x, 1)
        rob.update_importance_sampling(x, 1, e3[0])
        _, best_g, stride_g, _ = eng.getRMPPIState()
        _, best_o, stride_o, _ = rob.state()
        assert (best_g, stride_g) == (best_o, stride_o)
        eng.setFeedbackGains(g, False)
        rob.set_gains(g, False)
        eng.computeControl(x, 1)
        rob.compute_control(x, 1, e3[1:])
        du = float(max(np.abs(eng.getControlSeq() - orc.control()).max(),
                       np.abs(eng.getNominalControlSeq() - orc.nominal_control()).max()))
        drift.append(du)
        # call 0 starts from identical state: the parity bar.  Later calls run free (two control sequences, the nominal state
        # and the candidate search all carry the previous call's ~1e-6): reported, loosely bounded
        assert du <= (U_TOL if i == 0 else 2e-4), drift
        x, _ = orc.model_step(x, orc.control()[0])
        x = x + np.array([0.02, -0.01, 0.1, -0.05], np.float32)
    print("\nRMPPI robust-cost free-running u* distance per call:", ["%.2g" % d for d in drift])
    eng.close()
