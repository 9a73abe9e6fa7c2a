"""RacerDubinsElevationLSTMSteering (reference: dynamics/racer_dubins/racer_dubins_elevation_lstm_steering.cu)
and the host-side LSTMLSTMHelper (utils/nn_helpers/lstm_lstm_helper.cu).

Pinning.  The reference's known-answer tests for this class are disabled at this snapshot (TestStep is GTEST_SKIP,
tests/dynamics/racer_dubins_elevation_lstm_steering_model_test.cu:322-324; ComputeDynamics / TestUpdateState are commented
out) and every network file under resources/ is a git-LFS stub, so what pins it is
  * compareToElevationWithoutSteering (:775-945, live): with the default (zero) network every state, derivative and output
    except the steering ones equals the plain elevation model's — restated below on the oracle and on the device;
  * the steering equations of computeLSTMSteering (:131-167) evaluated by hand;
  * LSTMHelper itself: tests/test_lstm_helper.py (known answers of tests/nn_helpers/lstm_helper_test.cu);
  * LSTMLSTMHelper: initializeLSTMLSTMTest (tests/nn_helpers/lstm_lstm_helper_test.cu:161-180): all parameters 1, a buffer of
    ones -> hidden = cell = 101."""
import math

import numpy as np
import pytest

import pyoracle as po
from common import host_noise, m, make_engine, make_oracle, ulp_diff
from test_racer_dubins_elevation import elevation_cfg, hills, st

S_VEL, S_YAW, S_X, S_Y, S_STEER, S_BRAKE, S_ROLL, S_PITCH, S_STEER_RATE = range(9)
O_STEER, O_STEER_RATE = 8, 9
H, I = 4, 4
LSTM_PARAMS = 4 * H * H + 4 * H * I + 4 * H
OUT_LAYERS = [8, 20, 1]
OUT_PARAMS = 8 * 20 + 20 + 20 * 1 + 1


def steering_blobs(seed=21, scale=0.4, zero=False):
    rng = np.random.default_rng(seed)
    lstm = np.zeros(LSTM_PARAMS + 2 * H, np.float32) if zero else rng.uniform(-scale, scale, LSTM_PARAMS + 2 * H).astype(np.float32)
    out = np.zeros(OUT_PARAMS, np.float32) if zero else rng.uniform(-scale, scale, OUT_PARAMS).astype(np.float32)
    return {"lstm_weights": lstm, "lstm_output_weights": out}


def steering_cfg(zero=False, **kw):
    cfg = elevation_cfg(**kw)
    cfg["model"] = "racer_dubins_elevation_lstm_steering"
    blobs = dict(cfg.get("blobs", {}))
    blobs.update(steering_blobs(zero=zero))
    cfg["blobs"] = blobs
    return cfg


def test_lstm_lstm_helper_reference_known_answer():
    """initializeLSTMLSTMTest: init LSTM(8, 60) + {68, 100, 20}, prediction hidden size 10, init_len 6, all ones"""
    helper = m.LSTMLSTMHelper(8, 60, [68, 100, 20], 8, 10, [18, 2], 6)
    helper.setInitParams(np.ones_like(helper.init_lstm), np.ones_like(helper.init_output))
    hidden, cell = helper.initializeLSTM(np.ones((8, 10), np.float32))
    assert np.all(hidden == 101.0) and np.all(cell == 101.0)


def test_lstm_lstm_helper_against_numpy():
    """random initialiser: recurrent updates over the last init_len columns only, output network on [h ; x] of the last one"""
    rng = np.random.default_rng(3)
    Ii, Hi, layers, Hp, init_len = 3, 5, [8, 7, 8], 4, 4
    helper = m.LSTMLSTMHelper(Ii, Hi, layers, 4, Hp, [8, 20, 1], init_len)
    lstm = rng.uniform(-0.5, 0.5, helper.init_lstm.size).astype(np.float32)
    out = rng.uniform(-0.5, 0.5, helper.init_output.size).astype(np.float32)
    helper.setInitParams(lstm, out)
    buf = rng.uniform(-1, 1, (Ii, 9)).astype(np.float32)
    hidden, cell = helper.initializeLSTM(buf)

    L = lstm.astype(np.float64)
    Wm = [L[g * Hi * Hi:(g + 1) * Hi * Hi].reshape(Hi, Hi) for g in range(4)]
    o = 4 * Hi * Hi
    Wi = [L[o + g * Hi * Ii:o + (g + 1) * Hi * Ii].reshape(Hi, Ii) for g in range(4)]
    o += 4 * Hi * Ii
    b = [L[o + g * Hi:o + (g + 1) * Hi] for g in range(4)]
    h, c = L[o + 4 * Hi:o + 5 * Hi].copy(), L[o + 5 * Hi:o + 6 * Hi].copy()
    sig = lambda v: 1 / (1 + np.exp(-v))
    for t in range(9 - init_len, 9):
        x = buf[:, t].astype(np.float64)
        gi, gf, go = sig(Wm[0] @ h + Wi[0] @ x + b[0]), sig(Wm[1] @ h + Wi[1] @ x + b[1]), sig(Wm[2] @ h + Wi[2] @ x + b[2])
        gc = np.tanh(Wm[3] @ h + Wi[3] @ x + b[3])
        c = gi * gc + gf * c
        h = go * np.tanh(c)
    act = np.concatenate([h, buf[:, -1].astype(np.float64)])
    O = out.astype(np.float64)
    W1, b1 = O[:56].reshape(7, 8), O[56:63]
    W2, b2 = O[63:63 + 56].reshape(8, 7), O[63 + 56:]
    y = W2 @ np.tanh(W1 @ act + b1) + b2
    assert np.abs(np.concatenate([hidden, cell]) - y).max() <= 2e-6
    with pytest.raises(ValueError):   # the buffer must hold at least init_len samples (lstm_lstm_helper.cu:53)
        helper.initializeLSTM(buf[:, :init_len - 1])


def test_oracle_zero_network_equals_elevation_model():
    """compareToElevationWithoutSteering (:775-945) on the oracle: random states and controls, forward and reverse gear"""
    rng = np.random.default_rng(8)
    heights, transform = hills()
    for gear in (1, -1):
        cfg_a, cfg_b = steering_cfg(zero=True, K=64, T=4), elevation_cfg(K=64, T=4)
        cfg_a["dyn"].base.gear_sign = gear
        cfg_b["dyn"].base.gear_sign = gear
        a, b = make_oracle(cfg_a), make_oracle(cfg_b)
        for trial in range(200):
            x = st(*rng.uniform(-1, 1, 9))
            x[S_X], x[S_Y] = rng.uniform(-20, 20, 2)
            x[9:19] = rng.uniform(-0.05, 0.05, 10)
            u = rng.uniform(-1, 1, 2).astype(np.float32)
            xa, da, ya = a.model_step_full(x, u, 0.1)
            xb, db, yb = b.model_step_full(x, u, 0.1)
            keep_s = [i for i in range(19) if i not in (S_STEER, S_STEER_RATE)]
            keep_o = [i for i in range(28) if i not in (O_STEER, O_STEER_RATE, 10, 11, 12, 27)]
            assert np.array_equal(xa[keep_s], xb[keep_s]) and np.array_equal(da[keep_s], db[keep_s])
            assert np.array_equal(ya[keep_o], yb[keep_o])


def test_oracle_steering_equations_by_hand():
    """computeLSTMSteering (:131-167) + updateState (:243-268) with a network whose output is a known constant: all weights
    zero, last bias of the output network = 0.3 -> the network adds 0.3 * 5 to the steering acceleration"""
    cfg = steering_cfg(zero=True, K=64, T=4, with_map=False)
    cfg["blobs"]["lstm_output_weights"][-1] = 0.3
    p = cfg["dyn"].base
    o = make_oracle(cfg)
    for steer, rate, cmd in ((0.1, -0.2, 0.5), (-0.3, 0.4, -1.0), (0.0, 0.0, 1.0), (0.2, 3.0, -1.0)):
        x = st(1.0, 0.0, 0.0, 0.0, steer, 0.0, 0.0, 0.0, rate)
        xn, xd, y = o.model_step_full(x, np.array([0.2, cmd], np.float32), 0.05)
        parametric = (cmd * p.steer_command_angle_scale - steer) * p.steering_constant
        acc = min(max((parametric - rate) * p.steer_accel_constant - rate * p.steer_accel_drag_constant, -p.max_steer_rate),
                  p.max_steer_rate) + 0.3 * 5.0
        assert abs(xd[S_STEER_RATE] - acc) <= 1e-5 * max(1.0, abs(acc)) and xd[S_STEER] == np.float32(rate)
        assert abs(xn[S_STEER_RATE] - (rate + acc * 0.05)) <= 1e-5 * max(1.0, abs(acc))
        want_steer = min(max(steer + rate * 0.05, -p.max_steer_angle), p.max_steer_angle)
        assert abs(xn[S_STEER] - want_steer) <= 1e-6
        assert y[O_STEER] == xn[S_STEER] and y[O_STEER_RATE] == xn[S_STEER_RATE]


def test_oracle_network_state_carries_over_steps():
    """the hidden / cell state of a rollout persists from step to step and starts from the blob's (h0, c0)"""
    cfg = steering_cfg(K=64, T=12, with_map=False)
    o = make_oracle(cfg)
    u = np.tile(np.array([0.3, 0.4], np.float32), (12, 1))
    xs = o.state_trajectory(cfg["x0"], u)
    cfg2 = steering_cfg(K=64, T=12, with_map=False)
    cfg2["blobs"]["lstm_weights"][LSTM_PARAMS:] = 0.0   # another initial state
    xs2 = make_oracle(cfg2).state_trajectory(cfg["x0"], u)
    assert np.isfinite(xs).all() and np.abs(xs[:, S_STEER_RATE] - xs2[:, S_STEER_RATE]).max() > 1e-4
    # the same constant input gives a different increment at every step while the recurrent state settles
    incr = np.diff(xs[:, S_STEER_RATE])
    assert np.unique(np.round(incr, 6)).size > 6


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("with_map,block_y,variant", [(True, 1, 1), (False, 1, 1), (True, 4, 1), (True, 4, 2), (False, 4, 2)])
def test_lstm_steering_rollout_costs_bit_exact(gpu, with_map, block_y, variant):
    """block_y = 1: one lane per rollout (network on registers, LSTMRegisters); block_y = 4: four replica lanes share the
    hidden units, the MLP neurons, the wheels and the covariance rows (fused and role-pipelined kernel)"""
    cfg = steering_cfg(K=1000, T=60, with_map=with_map)
    eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=3)
    o = make_oracle(cfg)
    o.vanilla_compute_control(cfg["x0"], 1, eps)
    eng = make_engine(cfg, block_x=64, block_y=block_y, kernel_variant=variant)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    assert np.isfinite(o.costs()).all()
    assert ulp_diff(eng.getSampledCostSeq(), o.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - o.control()).max() <= 1e-5
    assert np.abs(eng.getTargetStateSeq() - o.state_traj()).max() <= 1e-4
    y = eng.getTargetOutputSeq()
    xs, ys = o.output_trajectory(cfg["x0"], o.control())
    finite = np.isfinite(ys)
    assert np.array_equal(np.isfinite(y), finite) and np.abs(y[finite] - ys[finite]).max() <= 1e-4


@pytest.mark.gpu
def test_lstm_steering_zero_network_equals_elevation_on_device(gpu):
    """compareToElevationWithoutSteering on the device: modelStep of both registered models"""
    rng = np.random.default_rng(9)
    a, b = make_engine(steering_cfg(zero=True, K=256, T=8)), make_engine(elevation_cfg(K=256, T=8))
    keep = [i for i in range(19) if i not in (S_STEER, S_STEER_RATE)]
    for trial in range(100):
        x = st(*rng.uniform(-1, 1, 9))
        x[S_X], x[S_Y] = rng.uniform(-20, 20, 2)
        x[9:19] = rng.uniform(-0.05, 0.05, 10)
        u = rng.uniform(-1, 1, 2).astype(np.float32)
        xa, _ = a.modelStep(x, u)
        xb, _ = b.modelStep(x, u)
        assert np.array_equal(xa[keep], xb[keep]), (trial, xa, xb)


@pytest.mark.gpu
def test_lstm_steering_initial_state_and_structure(gpu):
    """setLSTMInitialState == the same values in the blob's tail; another hidden size through "lstm_structure"; Tube"""
    cfg = steering_cfg(K=512, T=40)
    eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=5)
    h0 = np.array([0.3, -0.2, 0.1, 0.05], np.float32)
    c0 = np.array([-0.4, 0.2, 0.0, 0.6], np.float32)
    eng = make_engine(cfg)
    eng.setLSTMInitialState(h0, c0)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    cfg["blobs"]["lstm_weights"][LSTM_PARAMS:LSTM_PARAMS + H] = h0
    cfg["blobs"]["lstm_weights"][LSTM_PARAMS + H:] = c0
    o = make_oracle(cfg)
    o.vanilla_compute_control(cfg["x0"], 1, eps)
    assert ulp_diff(eng.getSampledCostSeq(), o.costs()).max() == 0

    # H = 6, output network {10, 12, 1}
    rng = np.random.default_rng(2)
    cfg = steering_cfg(K=512, T=40)
    Hn = 6
    blobs = {"lstm_structure": np.array([Hn, Hn + 4, 12, 1], np.float32),
             "lstm_weights": rng.uniform(-0.4, 0.4, 4 * Hn * Hn + 4 * Hn * 4 + 6 * Hn).astype(np.float32),
             "lstm_output_weights": rng.uniform(-0.4, 0.4, (Hn + 4) * 12 + 12 + 12 + 1).astype(np.float32)}
    maps = {k: v for k, v in cfg["blobs"].items() if k.startswith("elevation")}
    cfg["blobs"] = {**maps, **blobs}   # the structure first, then the weights
    cfg["blobs"] = dict(sorted(cfg["blobs"].items(), key=lambda kv: 0 if kv[0] == "lstm_structure" else 1))
    o = make_oracle(cfg)
    o.vanilla_compute_control(cfg["x0"], 1, eps)
    with pytest.raises(m.MPPIError):   # the four-lane form is compiled for the default network: asked for, it refuses
        eng = make_engine(cfg, block_x=64, block_y=4)
        eng.injectNoise(eps)
        eng.computeControl(cfg["x0"], 1)
    eng = make_engine(cfg)   # no shape asked for: the default (four lanes) gives way to the one-lane shape
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    assert ulp_diff(eng.getSampledCostSeq(), o.costs()).max() == 0
    eng = make_engine(cfg, block_x=64, block_y=1)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    assert ulp_diff(eng.getSampledCostSeq(), o.costs()).max() == 0
    assert np.abs(eng.getTargetStateSeq() - o.state_traj()).max() <= 1e-4

    cfg = steering_cfg(K=512, T=40, D=2)
    o = make_oracle(cfg)
    o.tube_compute_control(cfg["x0"], 1, eps)
    eng = make_engine(cfg)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    assert ulp_diff(eng.getSampledCostSeq(), o.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - o.control()).max() <= 1e-5


@pytest.mark.gpu
def test_lstm_steering_closed_loop_with_history_buffer(gpu):
    """the reference's control cycle: updateFromBuffer (:215-233) -> initializeLSTM on the host -> hidden / cell to the device
    -> computeControl; colored-noise sampler as in the RACER controllers"""
    cfg = steering_cfg(K=1024, T=64)
    cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
    eng = make_engine(cfg)
    rng = np.random.default_rng(4)
    helper = m.LSTMLSTMHelper(3, 20, [23, 100, 8], 4, 4, OUT_LAYERS, 11)   # the shape of the reference's tests (:26-32)
    helper.setInitParams(rng.uniform(-0.2, 0.2, helper.init_lstm.size), rng.uniform(-0.2, 0.2, helper.init_output.size))
    x = cfg["x0"].copy()
    history = np.zeros((3, 51), np.float32)   # STEER_ANGLE * 0.2, STEER_ANGLE_RATE * 0.2, STEER_CMD
    for i in range(60):
        hidden, cell = helper.initializeLSTM(history)
        eng.setLSTMInitialState(hidden, cell)
        eng.computeControl(x, 1)
        u = eng.getControlSeq()[0].copy()
        x, _ = eng.modelStep(x, u)
        eng.slideControlSequence(1)
        history = np.roll(history, -1, axis=1)
        history[:, -1] = [x[S_STEER] * 0.2, x[S_STEER_RATE] * 0.2, u[1]]
    assert np.isfinite(x).all() and np.isfinite(eng.getControlSeq()).all() and x[S_VEL] > 1.5
