"""Pins the CPU oracle against the reference's own known-answer tests and against independent restatements.

Each test names the reference test it reproduces.  These run on CPU (`-m "not gpu"`).
"""
import numpy as np

import pyoracle as po


def test_philox4x32_10_random123_known_answers():
    """Random123 kat_vectors for philox4x32-10 (the generator of rocRAND/cuRAND PHILOX4_32_10)"""
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        assert list(po.philox4x32_10(ctr, key)) == want


def test_philox_normal_statistics_and_shard_invariance():
    """reference sampler tests are statistical: tests/sampling_distributions/colored_noise_tests.cu:98-209 (68-95-99.7)"""
    e = po.philox_normal(42, 0, 4096, 100, 1).ravel()
    assert abs(e.mean()) < 5e-3 and abs(e.std() - 1) < 5e-3
    for k, frac in ((1, 0.6827), (2, 0.9545), (3, 0.9973)):
        assert abs((np.abs(e) < k).mean() - frac) < 3e-3
    # any shard of the K axis sees exactly the slice of the full stream
    full = po.philox_normal(7, 5, 1000, 33, 2)
    part = po.philox_normal(7, 5, 1000, 33, 2, 333, 667)
    assert np.array_equal(full[333:667], part)
    assert not np.array_equal(po.philox_normal(7, 6, 1000, 33, 2), full)  # generation advances the stream


def test_smooth_control_trajectory_kat():
    """reference: tests/controllers/controller_generic_tests.cu:214-238 (EXPECT_FLOAT_EQ = 4 ulp)"""
    hist = np.zeros((2, 3), np.float32)
    u = np.ones((1, 3), np.float32)
    s = po.smooth(u, hist)
    np.testing.assert_allclose(s[0], (1.0 * 17 + 1.0 * 12 + 1.0 * -3) / 35.0, rtol=4e-7)
    u = np.stack([np.ones(3), 2 * np.ones(3)]).astype(np.float32)
    s = po.smooth(u, hist)
    np.testing.assert_allclose(s[0], (1.0 * 17 + 2.0 * 12 + 2.0 * -3) / 35.0, rtol=4e-7)
    np.testing.assert_allclose(s[1], (1.0 * 12 + 2.0 * 17 + 2.0 * 12 + 2.0 * -3) / 35.0, rtol=4e-7)


def test_slide_control_sequence_kat():
    """reference: tests/controllers/controller_generic_tests.cu:240-283 (slide_control_scale_ = 0 pads with zero)"""
    T, C = 100, 2
    u = np.repeat(np.arange(T, dtype=np.float32)[:, None], C, 1)
    u = po.slide(u, 1)
    for i in range(T):
        want = 0 if i + 1 > T - 1 else min(i + 1, T - 1)
        assert np.all(u[i] == want)
    u = po.slide(u, 10)
    for i in range(T):
        want = 0 if i + 10 > T - 2 else min(i + 11, T - 1)
        assert np.all(u[i] == want)


def test_save_control_history():
    """reference: controllers/controller.cuh:602-615"""
    u = np.arange(20, dtype=np.float32).reshape(10, 2)
    h = np.array([[100, 101], [200, 201]], np.float32)
    h1 = po.save_history(1, u, h)
    assert np.array_equal(h1, [[200, 201], [0, 1]])
    h3 = po.save_history(3, u, h)
    assert np.array_equal(h3, [[2, 3], [4, 5]])
    assert np.array_equal(po.save_history(0, u, h), h)


def test_baseline_is_first_minimum_and_norm_exp():
    """reference: core/mppi_common.cu:885-900 (first-occurring min) and tests/mppi_core/normexp_kernel_tests.cu:126-150"""
    costs = np.array([5, 3, 7, 3, 9], np.float32)
    assert po.best_index(costs) == 1 and po.baseline(costs) == 3
    rng = np.random.default_rng(0)
    c = rng.uniform(0, 40, 5000).astype(np.float32)
    w = po.norm_exp(c, 2.0, float(c.min()))
    arg = (np.float32(-2.0) * (c - c.min()).astype(np.float32)).astype(np.float32)  # the fp32 argument expf receives
    ref = np.exp(arg.astype(np.float64))
    np.testing.assert_allclose(w, ref, rtol=5e-7, atol=1e-45)
    # normaliser accumulates in double (mppi_common.cu:1055-1063)
    assert abs(po.normalizer(w) - np.float32(w.astype(np.float64).sum())) == 0


def test_free_energy_formula():
    """reference: core/mppi_common.cu:1065-1081"""
    rng = np.random.default_rng(3)
    w = rng.uniform(0, 1, 2048).astype(np.float32)
    fe, var, mod = po.free_energy(w, 12.5, 0.7)
    norm = w.astype(np.float64).mean()
    v = (w.astype(np.float64) ** 2).mean() - norm ** 2
    assert abs(fe - (-0.7 * np.log(norm) + 12.5)) < 1e-4
    assert abs(var - 0.7 * v) < 1e-5
    weird = 0.7 * v / (norm * np.sqrt(2048.0))
    assert abs(mod - 0.7 * (weird + 0.5 * weird ** 2)) < 1e-6


def test_weighted_reduction_summation_order():
    """reference: tests/mppi_core/weightedreduction_kernel_tests.cu:20-133 — restates the kernel's two-stage order
    (sum_stride consecutive rollouts per partial with weight = w/eta, then the partials in order) in numpy float32"""
    rng = np.random.default_rng(7)
    K, T, C, stride = 1024, 20, 6, 64
    w = (0.001 * rng.normal(1.0, 0.2, K)).astype(np.float32)
    v = rng.normal(1.0, 0.2, (K, T, C)).astype(np.float32)
    eta = np.float32(1000.0)
    got = po.weighted_reduction(w, v, float(eta), stride)
    cells = (K - 1) // stride + 1
    inter = np.zeros((cells, T, C), np.float32)
    weight = (w / eta).astype(np.float32)
    for k in range(K):
        inter[k // stride] = (inter[k // stride] + (weight[k] * v[k]).astype(np.float32)).astype(np.float32)
    want = np.zeros((T, C), np.float32)
    for j in range(cells):
        want = (want + inter[j]).astype(np.float32)
    assert np.array_equal(got, want)
    # and it is the weighted mean
    ref = np.einsum("k,ktc->tc", w.astype(np.float64) / float(eta), v.astype(np.float64))
    np.testing.assert_allclose(got, ref, rtol=2e-5)


def test_set_gaussian_controls_rules():
    """reference: sampling_distributions/gaussian/gaussian.cu:99-127"""
    K, T, C = 400, 12, 2
    o = po.Oracle("double_integrator", K, T, 2, 0.02, 1.0)
    o.set_sampler([[0.5, 2.0], [1.5, 0.25]], [0, 0], pure_noise_pct=0.01, std_dev_decay=0.9)
    rng = np.random.default_rng(0)
    mean = rng.normal(size=(2, T, C)).astype(np.float32)
    eps = rng.normal(size=(K, T, C)).astype(np.float32)
    v = o.set_gaussian_controls(mean, eps, stride=3, iteration=2)
    sd = (np.float32(0.9) * np.float32(0.9)) * np.array([[0.5, 2.0], [1.5, 0.25]], np.float32)
    for d in range(2):
        assert np.array_equal(v[d, 0], mean[d])                       # rollout 0: the mean exactly
        assert np.array_equal(v[d, :, :3], np.broadcast_to(mean[d, :3], (K, 3, C)))  # t < stride: the mean
        k = 100
        assert np.array_equal(v[d, k, 3:], (mean[d, 3:] + sd[d] * eps[k, 3:]).astype(np.float32))
        k = 399  # >= (1 - 0.01) * 400 = 396: zero-mean
        assert np.array_equal(v[d, k, 3:], (sd[d] * eps[k, 3:]).astype(np.float32))
        assert np.array_equal(v[d, 395, 3:], (mean[d, 3:] + sd[d] * eps[395, 3:]).astype(np.float32))


def test_cartpole_rollout_against_independent_numpy_float64():
    """The oracle's cartpole rollout against a from-scratch float64 numpy integration of the same ODE/cost
    (dynamics/cartpole/cartpole_dynamics.cu:89-107, cost_functions/cartpole/cartpole_quadratic_cost.cu:20-31):
    agreement to fp32 accuracy pins the formulas; the reference's own GPU-vs-CPU tolerance is 1e-4 relative
    (tests/mppi_core/rollout_kernel_tests.cu:258)."""
    from common import cartpole_cfg_lr, make_oracle
    cfg = cartpole_cfg_lr(K=64, T=50)
    o = make_oracle(cfg)
    rng = np.random.default_rng(5)
    eps = rng.normal(size=(64, 50, 1)).astype(np.float32)
    mean = (0.3 * np.cos(np.arange(50) * 0.2)).reshape(1, 50, 1).astype(np.float32)
    v = o.set_gaussian_controls(mean, eps, 1, 0)
    costs, vc = o.rollout_costs(cfg["x0"], mean, v)
    cp = cfg["cost"]
    goal = np.array(cp.desired_terminal_state[:], np.float64)
    coef = np.array([cp.cart_position_coeff, cp.cart_velocity_coeff, cp.pole_angle_coeff,
                     cp.pole_angular_velocity_coeff], np.float64)
    lam, alpha, ccoef, sd = cfg["lambda_"], cfg["alpha"], 0.7, 5.0
    want = np.zeros(64)
    for k in range(64):
        x = cfg["x0"].astype(np.float64).copy()
        run = 0.0
        for t in range(50):
            u = float(np.clip(np.float64(v[0, k, t, 0]), -5, 5))
            assert np.float32(u) == vc[0, k, t, 0]
            th, thd = x[2], x[3]
            s, c = np.sin(th), np.cos(th)
            den = 1.0 + 1.0 * s * s
            xd = np.array([x[1], (u + s * (thd * thd + 9.81 * c)) / den, thd,
                           (-u * c - thd * thd * c * s - 2.0 * np.float64(np.float32(9.81)) * s) / den])
            x = x + xd * 0.02
            mu = 0.0 if k >= 0.99 * 64 else float(mean[0, t, 0])
            run += float((coef * (x - goal) ** 2).sum()) + 0.5 * lam * (1 - alpha) * ccoef * mu * (mu - 2 * u) / sd ** 2
        want[k] = run / 50 + cp.terminal_cost_coeff * float((coef * (x - goal) ** 2).sum()) / 50
    np.testing.assert_allclose(costs[0], want, rtol=2e-4)


def test_controller_loop_consistency():
    """vanillaComputeControl == iterate + smooth + clamp assembled by hand (controllers/MPPI/mppi_controller.cu:151-241)"""
    from common import cartpole_cfg, make_oracle, host_noise
    cfg = cartpole_cfg(K=256, T=30, soft=True, num_iters=2)
    eps = host_noise(2, 256, 30, 1)
    o = make_oracle(cfg)
    o.vanilla_compute_control(cfg["x0"], 1, eps)
    o2 = make_oracle(cfg)
    mean = np.zeros((1, 30, 1), np.float32)
    for it in range(2):
        mean = o2.iterate(cfg["x0"], mean, eps[it], 1, it)
    u = po.smooth(mean[0], np.zeros((2, 1), np.float32))
    u = np.clip(u, -5, 5)
    assert np.array_equal(o.control(), u)
    assert np.array_equal(o.state_traj(), o2.state_trajectory(cfg["x0"], po.smooth(mean[0], np.zeros((2, 1), np.float32))))


def test_tube_sticky_scenario_occurs_in_the_oracle():
    """the seeds of test_gpu_parity.py::test_tube_take_over_is_sticky_within_a_call do produce the case it is there for: a call
    whose pass 0 restarts the nominal system from the actual state and whose pass 1 keeps it — and not only as the last call
    (the stale host state the advisor found in round 5 shows in the call AFTER it)"""
    from common import di_cfg
    from test_gpu_parity import _tube_sticky_scenario
    kinds = [r[2] for r in _tube_sticky_scenario(di_cfg(K=256, T=30, tube=True, num_iters=2))]
    assert any(kinds[:-1]), kinds
