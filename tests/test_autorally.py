"""AutoRally NeuralNetModel + ARStandardCost (BASELINE config 4)."""
import numpy as np
import pytest

import pyoracle as po
from common import autorally_cfg, host_noise, make_engine, make_oracle, standard_track_map, ulp_diff


# ------------------------------------------------------------------ CPU: oracle pinned on the reference's KATs --------
def test_fnn_all_ones_known_answer():
    """reference: tests/nn_helpers/fnn_helper_test.cu:403-446 — 6-32-32-4, all weights/biases/inputs 1 => 33"""
    n = 6 * 32 + 32 + 32 * 32 + 32 + 32 * 4 + 4
    out = po.fnn_forward([6, 32, 32, 4], np.ones(n), np.ones(6))
    assert np.array_equal(out, np.full(4, 33.0, np.float32))
    # small net of the reference's compute test fixture (generateTestNetwork.py:29-38): 4-3-4 all ones
    out = po.fnn_forward([4, 3, 4], np.ones(4 * 3 + 3 + 3 * 4 + 4), np.ones(4))
    np.testing.assert_allclose(out, 3 * np.tanh(5.0) + 1, rtol=2e-7)


def test_fnn_against_numpy_float64():
    rng = np.random.default_rng(0)
    layers = [6, 32, 32, 4]
    Ws = [rng.uniform(-0.5, 0.5, (layers[i + 1], layers[i])) for i in range(3)]
    bs = [rng.uniform(-0.5, 0.5, layers[i + 1]) for i in range(3)]
    theta = np.concatenate([np.concatenate([W.ravel(), b]) for W, b in zip(Ws, bs)])
    for _ in range(20):
        x = rng.uniform(-2, 2, 6)
        a = x
        for i in range(3):
            a = Ws[i].astype(np.float32).astype(np.float64) @ a + bs[i].astype(np.float32)
            if i < 2:
                a = np.tanh(a)
        np.testing.assert_allclose(po.fnn_forward(layers, theta, x), a, rtol=2e-5, atol=2e-6)


def test_ar_dynamics_known_answers():
    """reference: tests/dynamics/ar_dynamics_nn_test.cu:251-279 (kinematics) and :445-481 (all-ones network => xdot[3..6] = 33)"""
    cfg = autorally_cfg(K=64, T=4)
    o = make_oracle(cfg)
    o.set_blob("dynamics_weights", np.ones(1412, np.float32))
    xd = o.state_deriv([0, 0, 0, 0, 1, 2, 0], [0, 0])
    np.testing.assert_allclose(xd[:3], [1, 2, 0], rtol=4e-7, atol=1e-7)
    xd = o.state_deriv([0, 0, np.pi / 2, 0, 3, 5, 1], [0, 0])
    np.testing.assert_allclose(xd[:3], [-5, 3, -1], rtol=4e-7)
    xd = o.state_deriv(np.ones(7), np.ones(2))
    assert np.array_equal(xd[3:], np.full(4, 33.0, np.float32))


def test_ar_standard_cost_pieces():
    """reference: tests/cost_functions/autorally/ar_standard_cost_test.cu — speed / slip / crash / track terms"""
    cfg = autorally_cfg(K=64, T=4)
    o = make_oracle(cfg)
    cmap, _ = standard_track_map()
    # on the track centre line y = 5 at x = -12: map value = (x+13)/30 at both ends of the car (+-0.5 m)
    # (x chosen off the texel boundaries: front x = -11.487 -> column 30, back x = -12.487 -> column 10)
    c, crash = o.state_cost([-11.987, 5.01, 0, 0, 6, 0, 0, 0])
    want_track = 200.0 * (abs(cmap[300, 10]) + abs(cmap[300, 30])) / 2
    assert crash == 0 and abs(c - want_track) < 1e-4
    # speed term: (4 - 6)^2 * 4.25
    c2, _ = o.state_cost([-11.987, 5.01, 0, 0, 4, 0, 0, 0])
    assert abs((c2 - c) - 4.25 * 4.0) < 1e-3
    # slip: atan(1/4)^2 * 10, beyond max_slip_ang adds crash_coeff
    c3, _ = o.state_cost([-11.987, 5.01, 0, 0, 4, 1, 0, 0])
    assert abs((c3 - c2) - 10.0 * np.arctan(0.25) ** 2) < 1e-3
    c4, _ = o.state_cost([-12, 5, 0, 0, 0.5, 4, 0, 0])
    assert c4 > 10000
    # off the track (y = 0): map value 5 >= boundary_threshold -> crash flag, crash cost
    c5, crash5 = o.state_cost([-12, 0, 0, 0, 6, 0, 0, 0])
    assert crash5 == 1 and c5 > 10000
    # roll over
    _, crash6 = o.state_cost([-12, 5, 0, 2.0, 6, 0, 0, 0])
    assert crash6 == 1
    # sticky crash flag is the caller's: crash = 1 in -> crash cost even on the track
    c7, _ = o.state_cost([-11.987, 5.01, 0, 0, 6, 0, 0, 0], 0, 1)
    assert abs((c7 - c) - 10000) < 1e-2
    # texture clamp addressing: far outside the map
    c8, crash8 = o.state_cost([1000, -1000, 0, 0, 6, 0, 0, 0])
    assert np.isfinite(c8) and crash8 == 1


# ------------------------------------------------------------------ GPU parity -----------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 8), (8, 16), (16, 4), (64, 1), (64, 4), (32, 4), (64, 4, 1), (64, 4, 2)])
def test_autorally_rollout_costs_bit_exact(gpu, shape):
    """reference: tests/dynamics/ar_dynamics_nn_test.cu (GPU == CPU over y_dim 1..16) + rollout_kernel_tests.cu.
    Shapes (64, 4) and (32, 4) are the MFMA forward (4 lanes of a rollout = MFMA k-groups), the others the LDS path."""
    cfg = autorally_cfg(K=512, T=40)
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
@pytest.mark.parametrize("K,T", [(200, 37), (64, 1), (130, 3), (72, 2), (512, 41)])
@pytest.mark.parametrize("mode", ["injected", "philox"])
def test_autorally_pipeline_ragged_and_short_horizons(gpu, K, T, mode):
    """the MFMA pipeline kernel (dynamics waves + two samplers + two relaying cost waves): partially filled last block,
    odd horizons (the last pair of steps is a single step), horizons shorter than one sampler trip / one relay round,
    both noise sources, non-zero likelihood-ratio coefficients and a crash-prone start so that the status word travels
    through the relay.  Reference edge cases: mppi_common.cu:1305-1310 (ragged K), rollout_kernel_tests.cu (T sweep)."""
    cfg = autorally_cfg(K=K, T=T)
    cfg["control_cost_coeff"] = [0.7, 0.3]
    cfg["x0"] = np.array([-12.0, 5.0, 0.4, 0.0, 6.0, 0.5, 0.0], np.float32)
    eng = make_engine(cfg, block_x=64, block_y=4, kernel_variant=2, save_samples=True)
    orc = make_oracle(cfg)
    mean = (0.3 * np.sin(np.arange(T * 2, dtype=np.float32) * 0.2)).reshape(T, 2)
    eng.updateImportanceSampler(mean)
    if mode == "injected":
        eps = host_noise(1, K, T, 2)[0]
        eng.injectNoise(eps)
    else:
        eps = po.philox_normal(42, 0, K, T, 2)
    g = eng.rolloutCosts(cfg["x0"], 1)
    v = orc.set_gaussian_controls(mean[None], eps, 1, 0)
    c, vc = orc.rollout_costs(cfg["x0"], mean[None], v)
    assert g.shape == c.shape and np.isfinite(g).all()
    assert ulp_diff(g, c).max() == 0, ulp_diff(g, c).max()
    assert ulp_diff(eng.getSampledControls(), vc).max() == 0


@pytest.mark.gpu
def test_autorally_compute_control_parity(gpu):
    cfg = autorally_cfg(K=1024, T=50, num_iters=2)
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
        eng.slideControlSequence(1)
        orc.vanilla_slide(1)


@pytest.mark.gpu
@pytest.mark.parametrize("block_x", [64, 32])
def test_autorally_tube_parity(gpu, block_x):
    """Tube-MPPI on the NN model: actual + nominal system in one launch on the MFMA forward, blocks of 64 or 32 rollouts
    (the latter is what long horizons fall back to); reference: controllers/Tube-MPPI/tube_mppi_controller.cu:157-299"""
    cfg = autorally_cfg(K=512, T=40, num_iters=1)
    cfg["D"] = 2
    eng, orc = make_engine(cfg, block_x=block_x, block_y=4), make_oracle(cfg)
    x = cfg["x0"].copy()
    for i in range(3):
        eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=21 + i)
        eng.injectNoise(eps)
        eng.computeControl(x, 1)
        orc.tube_compute_control(x, 1, eps)
        assert np.abs(eng.getControlSeq() - orc.control()).max() <= 1e-5
        assert np.abs(eng.getNominalControlSeq() - orc.nominal_control()).max() <= 1e-5
        assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= 1e-4
        x = x + np.array([0.02, -0.01, 0.01, 0.0, 0.05, 0.0, 0.0], np.float32)  # the actual state drifts off the nominal


@pytest.mark.gpu
def test_autorally_requires_blobs(gpu):
    import mppi_generic_amd as m
    c = m.VanillaMPPIController("autorally_nn", 128, 10, 0.02, 1.0)
    with pytest.raises(m.MPPIError) as e:
        c.computeControl(np.zeros(7, np.float32), 1)
    assert e.value.status == 7 and "dynamics_weights" in str(e.value)
    with pytest.raises(m.MPPIError) as e:
        c.setModelBlob("dynamics_weights", np.zeros(10, np.float32))
    assert e.value.status == 1
