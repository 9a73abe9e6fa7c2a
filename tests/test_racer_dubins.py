"""RacerDubins + QuadraticCost: the oracle restatement pinned against the reference's own known
answers (tests/dynamics/racer_dubins_model_test.cu: ComputeDynamics :36-162, TestUpdateState :321-380 — values copied as
data, EXPECT_FLOAT_EQ is 4 ulp), then the HIP plugins against the oracle."""
import math

import numpy as np
import pytest

import pyoracle as po
from common import host_noise, make_engine, make_oracle, racer_cfg, ulp_diff

# (state, control, expected derivative of the first six states + untouched seventh) from ComputeDynamics
DYN_KAT = [
    ([0, 0, 0, 0, 0, 0, 0], [0, 0], [4.9, 0, 0, 0, 0, 0, 0]),
    ([1, math.pi / 2, 0, 3, 0, 0, 0], [1, 0], [4.9 + 1.3 - 3.7, 0, 0, 1, 0, 0, 0]),
    ([1, 0, 0, 3, 0, 0, 0], [-1, 0], [4.9 - 3.7, 0, 1, 0, 0, 0.33, 0]),
    ([1, 0, 0, 3, 0, 0.33, 0], [-1, 0], [4.9 - 0.33 * 2.5 - 3.7, 0, 1, 0, 0, 0.33, 0]),
    ([1, 0, 0, 3, 0, 1.0, 0], [1, 0], [4.9 - 2.5 + 1.3 - 3.7, 0, 1, 0, 0, -0.9, 0]),
    ([-1, 0, 0, 3, 0, 0, 0], [1, 0], [4.9 + 3.7 + 1.3, 0, -1, 0, 0, 0, 0]),
    ([-1, 0, 0, 3, 0, 1.0, 0], [-1, 0], [4.9 + 2.5 + 3.7, 0, -1, 0, 0, 0, 0]),
    ([-3, 0, 0, 3, 0, 1.0, 0], [-1, 0], [4.9 + 2.5 + 3.7 * 3, 0, -3, 0, 0, 0, 0]),
    ([4, 0, 0, 3, 0, 1.0, 0], [-1, 0], [4.9 - 2.5 - 3.7 * 4, 0, 4, 0, 0, 0, 0]),
    ([1, math.pi, 0, 3, 0, 0, 0], [0, 1], [4.9 - 3.7, (1 / .3) * math.tan(0), -1, 0, 1 * 5 * 0.6, 0, 0]),
]
# (state, derivative, dt, expected next state) from TestUpdateState
UPD_KAT = [
    ([0, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 0], 0.1, [0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 1]),
    ([0, math.pi - 0.1, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, -1, 1], 1.0, [1.0, 1.0 - math.pi - 0.1, 1.0, 1.0, 0.5, 0, 1]),
    ([0, math.pi - 0.1, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 2, 1], 1.0, [1.0, 1.0 - math.pi - 0.1, 1.0, 1.0, 0.5, 1.0, 1]),
    ([0, -math.pi + 0.1, 0, 0, 0, 0, 0], [1, -1, 1, 1, 1, 1, 1], 1.0, [1.0, math.pi + 0.1 - 1.0, 1.0, 1.0, 0.5, 1, 1]),
]


def close(a, b, ulps=4, atol=2e-7):
    a, b = np.float32(a), np.float32(b)
    return abs(float(a) - float(b)) <= atol or ulp_diff(np.array([a]), np.array([b])).max() <= ulps


def test_oracle_reproduces_reference_compute_dynamics_kat():
    o = make_oracle(racer_cfg(K=64, T=4))
    for x, u, want in DYN_KAT:
        got = o.state_deriv(np.array(x, np.float32), np.array(u, np.float32))
        for i in range(7):
            assert close(got[i], want[i]), (x, u, i, got[i], want[i])


def test_oracle_reproduces_reference_update_state_kat():
    o = make_oracle(racer_cfg(K=64, T=4))
    for x, xd, dt, want in UPD_KAT:
        got = o.update_state(np.array(x, np.float32), np.array(xd, np.float32), dt)
        for i in range(7):
            assert close(got[i], want[i], ulps=8, atol=3e-7), (x, xd, i, got[i], want[i])


def test_oracle_quadratic_cost_and_det_tan():
    cfg = racer_cfg(K=64, T=4)
    o = make_oracle(cfg)
    y = np.zeros(28, np.float32)
    y[0], y[2], y[3], y[4], y[6] = 1.1, 3.0, 3.0, 0.2, -1.0
    want = 40.0 * 0.25 + 1.0 * 4.0 + 1.0 * 1.0 + 0.5 * 0.04 + 0.05 * 1.0
    assert abs(o.state_cost(y)[0] - want) <= 1e-5 * want
    xs = np.linspace(-1.4, 1.4, 57).astype(np.float32)
    t = po.det_eval(12, xs)
    assert np.abs(t - np.tan(xs.astype(np.float64))).max() <= 4e-6
    assert ulp_diff(t, np.tan(xs.astype(np.float64)).astype(np.float32)).max() <= 4


def test_oracle_closed_loop_reaches_speed():
    """a few MPPI steps with the oracle alone: the controller accelerates towards the goal speed"""
    cfg = racer_cfg(K=512, T=40)
    o = make_oracle(cfg)
    x = cfg["x0"].copy()
    for i in range(60):
        o.vanilla_compute_control(x, 1, host_noise(1, cfg["K"], cfg["T"], 2, seed=100 + i))
        u = o.control()[0].copy()
        x, _ = o.model_step(x, u)
        o.vanilla_slide(1)
    assert 1.4 < x[0] < 1.7 and np.isfinite(x).all()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 2])
def test_racer_dubins_rollout_costs_bit_exact(gpu, variant):
    cfg = racer_cfg(K=1000, T=60)
    eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=3)
    o = make_oracle(cfg)
    o.vanilla_compute_control(cfg["x0"], 1, eps)
    eng = make_engine(cfg, kernel_variant=variant)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    assert ulp_diff(eng.getSampledCostSeq(), o.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - o.control()).max() <= 1e-5
    assert np.abs(eng.getTargetStateSeq() - o.state_traj()).max() <= 1e-4


@pytest.mark.gpu
def test_racer_dubins_model_step_and_closed_loop(gpu):
    cfg = racer_cfg(K=2048, T=60)
    o = make_oracle(cfg)
    eng = make_engine(cfg)
    # model step (enforceConstraints + step) on the device equals the oracle's, including the reference's known answers
    for x, u, _ in DYN_KAT:
        xe, ue = eng.modelStep(np.array(x, np.float32), np.array(u, np.float32))
        xo, uo = o.model_step(np.array(x, np.float32), np.array(u, np.float32))
        assert ulp_diff(xe, xo).max() == 0 and np.array_equal(ue, uo)
    x = cfg["x0"].copy()
    for i in range(100):
        eng.computeControl(x, 1)
        u = eng.getControlSeq()[0].copy()
        x, _ = eng.modelStep(x, u)
        eng.slideControlSequence(1)
    assert 1.45 < x[0] < 1.7 and np.isfinite(x).all()  # at the goal speed of the quadratic cost


@pytest.mark.gpu
def test_target_output_sequence_matches_oracle(gpu):
    """getTargetOutputSeq (computeOutputTrajectoryHelper, controllers/controller.cuh:643-662): outputs after
    initializeDynamics and after every step of the optimal trajectory; for the plain RacerDubins outputs 0..6 are the
    states and 7..27 stay zero"""
    cfg = racer_cfg(K=512, T=40)
    eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=9)
    o = make_oracle(cfg)
    o.vanilla_compute_control(cfg["x0"], 1, eps)
    eng = make_engine(cfg)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    y = eng.getTargetOutputSeq()
    xs, ys = o.output_trajectory(cfg["x0"], o.control())
    assert y.shape == (cfg["T"], 28)
    assert np.abs(y - ys).max() <= 1e-4
    assert np.abs(y[:, :7] - eng.getTargetStateSeq()).max() == 0 and np.abs(y[:, 7:]).max() == 0
