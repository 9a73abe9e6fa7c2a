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
    # EXPECT_FLOAT_EQ (4 ulp) in the reference; the reference's own summation order gives 33 exactly (test_fnn_output_order.py),
    # the matrix-core networks' order (the model's, FNN::split_output_sum) lands 1 ulp below: 32 x tanh(7) is not 32
    assert ulp_diff(xd[3:], np.full(4, 33.0, np.float32)).max() <= 4


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


def _ar_oracle(**params):
    """oracle with ARStandardCost on the reference's generated standard track map, parameters as the reference test sets them"""
    import mppi_generic_amd as m
    cfg = autorally_cfg(K=64, T=4)
    cost = m.ARStandardCostParams()
    cost.setTransformFromBounds(-13.0, 17.0, -10.0, 20.0)
    for k, v in params.items():
        setattr(cost, k, v)
    cfg["cost"] = cost
    return make_oracle(cfg)


def test_ar_cost_reference_known_answers_terms():
    """the oracle's ARStandardCost terms against the values the REFERENCE's own tests hold
    (tests/cost_functions/autorally_standard_cost_test.cu): coorTransformTest :184-210, getSpeedCostTest :721-748,
    getStablizingCostTest :750-801, getCrashCostTest :803-828"""
    import mppi_generic_amd as m
    # coorTransformTest: r_c1 = (0,1,2), r_c2 = (3,4,5), trs = (6,7,8), (x, y) = (0, 10) -> (36, 47, 58)
    o = _ar_oracle()
    cost = m.ARStandardCostParams()
    cost.r_c1[:] = [0, 1, 2]
    cost.r_c2[:] = [3, 4, 5]
    cost.trs[:] = [6, 7, 8]
    o.set_cost_params(cost)
    assert np.array_equal(o.ar_coor_transform(0.0, 10.0), np.array([36, 47, 58], np.float32))
    # getSpeedCostTest
    s = np.zeros(7, np.float32)
    s[4] = 10
    assert _ar_oracle(desired_speed=25, speed_coeff=10).ar_cost_term("speed", s)[0] == np.float32(15 * 15 * 10)
    assert _ar_oracle(desired_speed=0, speed_coeff=100).ar_cost_term("speed", s)[0] == np.float32(10 * 10 * 100)
    # getStablizingCostTest: EXPECT_FLOAT_EQ = 4 ulp
    o = _ar_oracle(slip_coeff=25, crash_coeff=1000, max_slip_ang=0.5)
    s = np.zeros(7, np.float32)
    s[4] = 0.1
    assert o.ar_cost_term("stabilizing", s) == (0.0, 0)
    s[5] = 0.01
    v, crash = o.ar_cost_term("stabilizing", s)
    assert ulp_diff(np.float32(v), np.float32(0.2483460072)) <= 4 and crash == 0
    s[5] = 0.2
    v, crash = o.ar_cost_term("stabilizing", s)
    assert ulp_diff(np.float32(v), np.float32(1030.6444)) <= 4 and crash == 0
    s[3], s[5] = 1.6, 0.0
    assert o.ar_cost_term("stabilizing", s) == (0.0, 1)
    s[3] = -1.6
    assert o.ar_cost_term("stabilizing", s) == (0.0, 1)
    # getCrashCostTest
    o = _ar_oracle(crash_coeff=10000)
    s = np.zeros(7, np.float32)
    s[4] = 10
    assert o.ar_cost_term("crash", s, 0)[0] == 0.0
    assert o.ar_cost_term("crash", s, 1)[0] == 10000.0


def _reference_costmap_value(state, width=30, height=30, x_min=-13, x_max=17, y_min=-10, y_max=20, ppm=20):
    """calculateStandardCostmapValue of the reference test (autorally_standard_cost_test.cu:830-848), float64"""
    x, y, th = state
    xf, yf = x + 0.5 * np.cos(th), y + 0.5 * np.sin(th)
    xb, yb = x - 0.5 * np.cos(th), y - 0.5 * np.sin(th)
    nx, ny = max(min(xf - x_min, x_max - x_min), 0.0), max(min(yf - y_min, y_max - y_min), 0.0)
    front = abs(height / 2.0 - ny) + nx / width
    nx = max(min(xb - x_min + 1.0 / (width * ppm), x_max - x_min), 0.0)
    ny = max(min(yb - y_min + 1.0 / (height * ppm), y_max - y_min), 0.0)
    back = abs(height / 2.0 - ny) + nx / width
    return (front + back) / 2.0


def test_ar_cost_reference_track_cost():
    """getTrackCostTest (autorally_standard_cost_test.cu:850-895): track_coeff 1, slop 0, boundary_threshold 1 on
    track_map_standard.npz; the reference compares its GPU result with calculateStandardCostmapValue within 0.001
    (states 0, 1) and 0.1 (states 2, 3 — the car's ends sit exactly on texel boundaries there), crash = 1 everywhere"""
    o = _ar_oracle(track_coeff=1, track_slop=0.0, boundary_threshold=1.0)
    for (x, y, th), tol in [((-13.5, -10, 0.0), 0.001), ((0, -10.0, 0.0), 0.001), ((0.0, 0.0, np.pi / 2), 0.1),
                            ((3.0, 0.0, np.pi / 2), 0.1)]:
        v, crash = o.ar_cost_term("track", [x, y, th, 0, 0, 0, 0])
        assert abs(v - _reference_costmap_value((x, y, th))) <= tol, (x, y, th, v)
        assert crash == 1


def test_ar_cost_reference_compute_cost_individual():
    """computeCostIndividualTest (autorally_standard_cost_test.cu:897-981): state (3, 0, pi/2, 0, 2, 1, 0.1), timestep 1,
    discount 0.9, one coefficient switched on at a time.  Speed, slip, crash and the discounted crash term are
    EXPECT_FLOAT_EQ values; the track term (1116.3333 on the reference's GPU) has both ends of the car exactly ON a texel
    boundary (y_map = 10.5 m and 9.5 m at 20 px/m, x_map = 16 m), where CUDA's texture unit — it truncates the normalised
    coordinate to fixed point before scaling — lands one texel lower than floor(u * width) evaluated in fp32: 0.87 %,
    the same effect the reference's own getTrackCostTest absorbs with its 0.1 tolerance.  Off texel boundaries the
    two agree (states 0 and 1 of test_ar_cost_reference_track_cost)."""
    y = [3.0, 0.0, np.pi / 2, 0.0, 2.0, 1.0, 0.1, 0.0]
    zero = dict(track_coeff=0, speed_coeff=0, crash_coeff=0.0, slip_coeff=0.0, discount=0.9)

    def cost(t=1, **kw):
        p = dict(zero)
        p.update(kw)
        return np.float32(_ar_oracle(**p).state_cost(y, t, 0)[0])

    assert cost() == 0.0
    speed_cost = np.float32(4.0 ** 2 * 4.25)
    assert cost(speed_coeff=4.25) == speed_cost
    slip_cost = np.float32(np.float32(np.arctan(np.float32(0.5))) ** 2 * np.float32(10))
    assert ulp_diff(cost(slip_coeff=10), slip_cost) <= 4
    track_cost = cost(track_coeff=200.0)
    assert abs(track_cost - 1116.3333) <= 0.01 * 1116.3333  # texel-boundary case, see the docstring
    # one texel lower at both ends (what the reference's GPU sampled) reproduces the reference's number to fp32
    cmap, _ = standard_track_map()
    assert abs(200.0 * (cmap[209, 319] + cmap[189, 319]) / 2.0 - 1116.3333) < 1e-3
    assert ulp_diff(cost(crash_coeff=10000), np.float32(9000)) <= 4
    total = cost(speed_coeff=4.25, track_coeff=200, slip_coeff=10, crash_coeff=10000)
    assert ulp_diff(total, np.float32(speed_cost + slip_cost + track_cost + np.float32(9000))) <= 4
    total4 = cost(t=4, speed_coeff=4.25, track_coeff=200, slip_coeff=10, crash_coeff=10000)
    want4 = np.float32(speed_cost + slip_cost + track_cost + np.float32(0.9) ** 4 * np.float32(10000))
    assert ulp_diff(total4, want4) <= 4


def test_ar_cost_reference_overflow():
    """computeCostOverflowTest (autorally_standard_cost_test.cu:983-1032): cost > MAX_COST_VALUE or NaN -> MAX_COST_VALUE (1e16)"""
    y = [3.0, 0.0, np.pi / 2, 0.0, 2.0, 1.0, 0.1, 0.0]
    base = dict(track_coeff=0, speed_coeff=10, crash_coeff=0.0, slip_coeff=0.0)
    assert _ar_oracle(desired_speed=1e16, **base).state_cost(y, 1, 0)[0] == np.float32(1e16)
    assert _ar_oracle(desired_speed=float("nan"), **base).state_cost(y, 1, 0)[0] == np.float32(1e16)


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


@pytest.mark.gpu
def test_trajectory_rerollout_wave_form_equals_mfma_form(gpu):
    """the re-rollout of u* (state / output trajectories of computeControl) runs on the one-rollout-per-wave form of the NN
    model (lane = neuron, fnn_wave.hpp); the replicated-lane MFMA form gives the same bits — Vanilla and Tube"""
    import os
    from common import autorally_cfg, make_engine
    for tube in (False, True):
        cfg = autorally_cfg(K=512, T=150)
        if tube:
            cfg["D"] = 2
        got = []
        for form in (None, "rep"):
            if form:
                os.environ["MPPI_AMD_FINALIZE_FORM"] = form
            try:
                eng = make_engine(cfg, tube=tube)
                x = cfg["x0"].copy()
                eng.computeControl(x, 1)
                got.append((eng.getControlSeq().copy(), eng.getTargetStateSeq().copy(), eng.getTargetOutputSeq().copy()))
                eng.close()
            finally:
                os.environ.pop("MPPI_AMD_FINALIZE_FORM", None)
        for a, b in zip(*got):
            assert np.array_equal(a, b)
        assert np.isfinite(got[0][1]).all() and np.abs(got[0][1][-1] - got[0][1][0]).max() > 1e-3
