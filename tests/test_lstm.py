"""LSTMHelper + LSTM bicycle-slip dynamics (BASELINE config 5 dynamics)."""
import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import bicycle_lstm_cfg, host_noise, lstm_npz, make_engine, make_oracle, ulp_diff


# ------------------------------------------------------------------ CPU: oracle pinned on the reference's KATs --------
def test_lstm_all_ones_known_answer():
    """reference: tests/nn_helpers/lstm_helper_test.cu:677-724 (forwardCPU) and :726-790 (forwardGPU, same values):
    LSTM(8, 20) + output layer {28, 3}, every parameter, initial state and input 1 -> five consecutive outputs"""
    I, H = 8, 20
    lstm = np.ones(4 * H * H + 4 * H * I + 4 * H + 2 * H, np.float32)
    fnn = np.ones(28 * 3 + 3, np.float32)
    out = po.lstm_forward(I, H, [28, 3], lstm, fnn, np.ones((5, I)))
    want = np.array([28.28055, 28.901096, 28.986588, 28.998184, 28.999756], np.float32)
    # the reference compares its GPU forward to these with 1e-4 (lstm_helper_test.cu:780-784); det::tanh is <= 7 ulp,
    # amplified by the 20 hidden units summed by the all-ones output layer: 2e-5 here
    assert np.abs(out - want[:, None]).max() <= 2e-5, out[:, 0]
    want64 = 20.0 * np.tanh(np.arange(2.0, 7.0)) + 9.0  # closed form: c_t = t + 1, gates saturate at 1
    assert np.abs(out[:, 0] - want64).max() <= 2e-5


def _numpy_lstm(d, xs, h, c):
    """float64 PyTorch-convention LSTM (gate order i, f, g, o) + tanh MLP, the definition the reference loads from"""
    Whh, Wih = d["lstm/weight_hh_l0"], d["lstm/weight_ih_l0"]
    b = d["lstm/bias_hh_l0"] + d["lstm/bias_ih_l0"]
    H = h.size
    outs = []
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    for x in xs:
        z = Whh @ h + Wih @ x + b
        i, f, g, o = sig(z[:H]), sig(z[H:2 * H]), np.tanh(z[2 * H:3 * H]), sig(z[3 * H:])
        c = i * g + f * c
        h = o * np.tanh(c)
        a = np.tanh(d["output/dynamics_W1"] @ np.concatenate([h, x]) + d["output/dynamics_b1"])
        outs.append(d["output/dynamics_W2"] @ a + d["output/dynamics_b2"])
    return np.array(outs)


def test_lstm_blob_order_against_numpy_float64():
    """gate re-ordering i,f,g,o -> i,f,o,c and bias summation of LSTMHelper::loadParams (lstm_helper.cu:549-578)"""
    d = lstm_npz(seed=3)
    d32 = {k: np.asarray(v, np.float64).astype(np.float32).astype(np.float64) for k, v in d.items()}
    d32["lstm/bias_hh_l0"] = (d["lstm/bias_hh_l0"] + d["lstm/bias_ih_l0"]).astype(np.float32).astype(np.float64)
    d32["lstm/bias_ih_l0"] = np.zeros_like(d32["lstm/bias_hh_l0"])
    lstm, fnn = m.lstm_blob_from_npz_dict(d)
    rng = np.random.default_rng(1)
    xs = rng.uniform(-1.5, 1.5, (12, 6)).astype(np.float32)
    got = po.lstm_forward(6, 16, [22, 32, 4], lstm, fnn, xs)
    want = _numpy_lstm(d32, xs.astype(np.float64), d32["lstm/h0"].copy(), d32["lstm/c0"].copy())
    np.testing.assert_allclose(got, want, rtol=3e-5, atol=3e-6)


def test_bicycle_lstm_oracle_kinematics_and_state():
    cfg = bicycle_lstm_cfg(K=64, T=4)
    o = make_oracle(cfg)
    xd = o.state_deriv([0, 0, np.pi / 2, 0, 3, 5, 1], [0, 0])
    np.testing.assert_allclose(xd[:3], [-5, 3, -1], rtol=4e-7)
    # the recurrent state matters: two consecutive model steps from the same x differ from two fresh single steps
    lstm, fnn = cfg["blobs"]["lstm_weights"], cfg["blobs"]["lstm_output_weights"]
    x = np.array([0.1, 1.0, -0.2, 0.3, 0.4, -0.5], np.float32)
    out = po.lstm_forward(6, 16, [22, 32, 4], lstm, fnn, np.stack([x, x]))
    assert np.abs(out[0] - out[1]).max() > 1e-4
    # (the model evaluates its output layer in the split order of the matrix-core networks: oracle_models.hpp, FNN)
    np.testing.assert_array_equal(xd[3:], po.lstm_forward(6, 16, [22, 32, 4], lstm, fnn, [[0, 3, 5, 1, 0, 0]], split_output_sum=True)[0])


# ------------------------------------------------------------------ GPU parity -----------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 8), (8, 16), (16, 4), (64, 1), (64, 4), (32, 4), (64, 4, 1), (64, 4, 2)])
def test_bicycle_lstm_rollout_costs_bit_exact(gpu, shape):
    """reference: tests/nn_helpers/lstm_helper_test.cu forwardGPU (GPU == CPU over y_dim 1..16) + rollout_kernel_tests.cu.
    (64, 4), (32, 4): MFMA forward, recurrent state in registers; the other shapes: LSTMHelper's LDS scheme."""
    cfg = bicycle_lstm_cfg(K=512, T=40)
    eps = host_noise(1, cfg["K"], cfg["T"], 2)[0]
    # a third entry selects the kernel structure for the MFMA shape: 1 = fused, 2 = role-pipelined (the default there)
    variant = shape[2] if len(shape) > 2 else 0
    eng = make_engine(cfg, block_x=shape[0], block_y=shape[1], kernel_variant=variant)
    orc = make_oracle(cfg)
    mean = np.zeros((cfg["T"], 2), np.float32)
    mean[:, 1] = 0.3
    eng.updateImportanceSampler(mean)
    eng.injectNoise(eps)
    g = eng.rolloutCosts(cfg["x0"], 1)
    v = orc.set_gaussian_controls(mean[None], eps, 1, 0)
    c, _ = orc.rollout_costs(cfg["x0"], mean[None], v)
    assert np.isfinite(g).all()
    assert (c < 1e4).sum() > 50, "test config should keep a good share of rollouts on the track"
    assert ulp_diff(g, c).max() == 0, ulp_diff(g, c).max()


@pytest.mark.gpu
def test_bicycle_lstm_compute_control_parity(gpu):
    cfg = bicycle_lstm_cfg(K=1024, T=50, num_iters=2)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    x = cfg["x0"].copy()
    for i in range(3):
        eps = host_noise(2, cfg["K"], cfg["T"], 2, seed=11 + i)
        eng.injectNoise(eps)
        eng.computeControl(x, 1)
        orc.vanilla_compute_control(x, 1, eps)
        assert np.abs(eng.getControlSeq() - orc.control()).max() <= 1e-5
        assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= 1e-4
        x, _ = orc.model_step(x, orc.control()[0])
        xg, _ = eng.modelStep(x, orc.control()[0])
        eng.slideControlSequence(1)
        orc.vanilla_slide(1)


@pytest.mark.gpu
def test_bicycle_lstm_model_step_matches_oracle(gpu):
    cfg = bicycle_lstm_cfg(K=64, T=4)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    x = np.array([-12, 5, 0.3, 0.05, 4, 0.2, -0.1], np.float32)
    u = np.array([0.4, 0.9], np.float32)
    xg, ug = eng.modelStep(x, u)
    xo, uo = orc.model_step(x, u)
    assert ulp_diff(xg, xo).max() == 0 and ulp_diff(ug, uo).max() == 0


@pytest.mark.gpu
def test_bicycle_lstm_requires_blobs(gpu):
    c = m.VanillaMPPIController("bicycle_slip_lstm", 128, 10, 0.02, 1.0)
    with pytest.raises(m.MPPIError) as e:
        c.computeControl(np.zeros(7, np.float32), 1)
    assert e.value.status == 7 and "lstm_weights" in str(e.value)
    with pytest.raises(m.MPPIError) as e:
        c.setModelBlob("lstm_weights", np.zeros(10, np.float32))
    assert e.value.status == 1


@pytest.mark.gpu
def test_trajectory_rerollout_wave_form_equals_mfma_form(gpu):
    """the re-rollout of u* runs on the one-rollout-per-wave form of the LSTM model (lane = gate row / neuron, lstm_wave.hpp);
    the replicated-lane MFMA form gives the same bits, and both agree with the oracle's state trajectory"""
    import os
    cfg = bicycle_lstm_cfg(K=512, T=120)
    eps = host_noise(1, cfg["K"], cfg["T"], 2)
    got = []
    for form in (None, "rep"):
        if form:
            os.environ["MPPI_AMD_FINALIZE_FORM"] = form
        try:
            eng = make_engine(cfg)
            eng.injectNoise(eps)
            eng.computeControl(cfg["x0"], 1)
            got.append((eng.getControlSeq().copy(), eng.getTargetStateSeq().copy(), eng.getTargetOutputSeq().copy()))
            eng.close()
        finally:
            os.environ.pop("MPPI_AMD_FINALIZE_FORM", None)
    for a, b in zip(*got):
        assert np.array_equal(a, b)
    orc = make_oracle(cfg)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    assert np.abs(got[0][1] - orc.state_traj()).max() <= 1e-4
