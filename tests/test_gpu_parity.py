"""Parity of the HIP engine (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances (fp32 path):
  trajectory costs      bit-exact (0 ulp): same IEEE operations in the same order on both sides (det_math.h)
  control sequence u*   L-inf <= 1e-5 absolute — the bar of BASELINE.json; the only difference is the summation order of
                        the weighted reduction (block-local partials + rescale vs the reference's serial groups)
  baseline rho          exact;  normaliser eta: 1e-6 relative
"""
import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import (cartpole_cfg, cartpole_cfg_lr, di_cfg, host_noise, make_engine, make_oracle, ulp_diff)

pytestmark = pytest.mark.gpu
U_TOL = 1e-5


def _rollout_both(cfg, eps, mean=None, stride=1, **kw):
    K, T, D = cfg["K"], cfg["T"], cfg["D"]
    eng = make_engine(cfg, **kw)
    orc = make_oracle(cfg)
    C = eng.CONTROL_DIM
    mean = np.zeros((T, C), np.float32) if mean is None else mean
    eng.updateImportanceSampler(mean)
    eng.injectNoise(eps)
    x0 = np.tile(cfg["x0"], (D, 1))
    costs_gpu = eng.rolloutCosts(x0, stride)
    means = np.tile(mean, (D, 1, 1))
    v = orc.set_gaussian_controls(means, eps, stride, 0)
    costs_cpu, v_clamped = orc.rollout_costs(x0, means, v)
    return eng, orc, costs_gpu, costs_cpu, v_clamped


@pytest.mark.parametrize("shape", [(64, 1), (32, 1), (64, 4), (16, 4)])
def test_cartpole_rollout_costs_bit_exact(gpu, shape):
    """reference test: tests/mppi_core/rollout_kernel_tests.cu:200-261 (GPU rollout vs CPU rollout over block shapes;
    reference tolerance 1e-4 relative, here 0 ulp)"""
    cfg = cartpole_cfg(K=2048, T=100)
    eps = host_noise(1, cfg["K"], cfg["T"], 1)[0]
    eng, orc, g, c, _ = _rollout_both(cfg, eps, block_x=shape[0], block_y=shape[1])
    assert np.isfinite(g).all()
    assert ulp_diff(g, c).max() == 0, "max ulp diff %d" % ulp_diff(g, c).max()


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("mode", ["injected", "philox"])
def test_kernel_variants_agree(gpu, variant, mode):
    """fused (one wave per 64 rollouts) and pipeline (sampler / dynamics / cost waves) variants: identical costs, both
    noise sources, Cartpole and the two-system Double Integrator"""
    for cfg in (cartpole_cfg_lr(K=1000, T=50), cartpole_cfg(K=2048, T=99, soft=True), di_cfg(K=512, T=33, tube=True)):
        C = len(cfg["std_dev"])
        eng = make_engine(cfg, kernel_variant=variant, save_samples=True)
        orc = make_oracle(cfg)
        mean = (0.3 * np.sin(np.arange(cfg["T"] * C, dtype=np.float32) * 0.2)).reshape(cfg["T"], C)
        eng.updateImportanceSampler(mean)
        if mode == "injected":
            eps = host_noise(1, cfg["K"], cfg["T"], C)[0]
            eng.injectNoise(eps)
        else:
            eps = po.philox_normal(42, 0, cfg["K"], cfg["T"], C)
        x0 = np.tile(cfg["x0"], (cfg["D"], 1))
        g = eng.rolloutCosts(x0, 2)
        means = np.tile(mean, (cfg["D"], 1, 1))
        v = orc.set_gaussian_controls(means, eps, 2, 0)
        c, vc = orc.rollout_costs(x0, means, v)
        assert ulp_diff(g, c).max() == 0
        assert ulp_diff(eng.getSampledControls(), vc).max() == 0
        # and the merged result of one full iteration
        eng.uploadState(x0)
        eng.updateImportanceSampler(mean)
        if mode == "injected":
            eng.injectNoise(eps)
        else:
            eng.setSeed(42)
        eng.uploadState(x0)
        eng.optimize(1)
        u_orc = orc.iterate(x0, means, eps, 1, 0)  # mppi_optimize runs with optimization_stride 1 here
        assert np.abs(eng.getOptimalControlSeq() - u_orc).max() <= U_TOL


def test_cartpole_lr_terminal_nonzero_mean_bit_exact(gpu):
    cfg = cartpole_cfg_lr()
    eps = host_noise(1, cfg["K"], cfg["T"], 1)[0]
    mean = (0.5 * np.sin(np.arange(cfg["T"], dtype=np.float32) * 0.3)).reshape(-1, 1).astype(np.float32)
    eng, orc, g, c, _ = _rollout_both(cfg, eps, mean=mean, stride=3)
    assert ulp_diff(g, c).max() == 0


def test_ragged_rollout_count(gpu):
    """K not a multiple of the block: the last block is partially filled (reference exits, mppi_common.cu:1305-1310)"""
    cfg = cartpole_cfg(K=1000, T=37, soft=True)
    eps = host_noise(1, cfg["K"], cfg["T"], 1)[0]
    eng, orc, g, c, _ = _rollout_both(cfg, eps)
    assert ulp_diff(g, c).max() == 0
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL


@pytest.mark.parametrize("soft", [False, True])
def test_vanilla_compute_control_parity(gpu, soft):
    """one computeControl: u* (smoothed, constrained), nominal state trajectory, baseline, normaliser"""
    cfg = cartpole_cfg(K=2048, T=100, soft=soft)
    eps = host_noise(1, cfg["K"], cfg["T"], 1)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    u_g, u_c = eng.getControlSeq(), orc.control()
    assert np.abs(u_g - u_c).max() <= U_TOL, np.abs(u_g - u_c).max()
    assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= 1e-4
    st, so = eng.getStats(), orc.stats()
    assert st.real_sys.baseline == so["baseline"][0]
    assert abs(st.real_sys.normalizer - so["normalizer"][0]) <= 1e-6 * so["normalizer"][0] + 1e-6
    assert abs(st.real_sys.free_energy_mean - so["free_energy"][0]) <= 1e-4 * abs(so["free_energy"][0]) + 1e-4
    if soft:
        assert so["normalizer"][0] > 10.0  # the average really involves many rollouts


@pytest.mark.parametrize("T", [1, 2, 3, 5, 63, 64, 65, 129])
def test_compute_control_short_and_odd_horizons(gpu, T):
    """the post-processing pass (smoothing over [history | u | last, last], T - 1 re-rollout steps, per-column
    constraints, write-out shared by the 64 lanes of a wave) at horizons around its loop strides, down to T = 1;
    Cartpole and the two-system Double Integrator (Tube)"""
    for cfg, tube in ((cartpole_cfg(K=256, T=T, soft=True), False), (di_cfg(K=256, T=T, tube=True), True)):
        C = len(cfg["std_dev"])
        eps = host_noise(1, cfg["K"], T, C, seed=5)
        eng, orc = make_engine(cfg), make_oracle(cfg)
        eng.injectNoise(eps)
        eng.computeControl(cfg["x0"], 1)
        if tube:
            orc.tube_compute_control(cfg["x0"], 1, eps)
        else:
            orc.vanilla_compute_control(cfg["x0"], 1, eps)
        assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL
        assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= 1e-4


def test_long_horizon_moves_the_rows_or_picks_a_smaller_block(gpu):
    """T * C floats per rollout live in LDS: when the default block's rows do not fit the 160 KiB, mppi_create moves the rows
    to HBM (every kernel variant, both samplers — tests/test_long_horizon.py); a pipeline variant requested for a model that
    has none still says so"""
    cfg = cartpole_cfg(K=300, T=700, soft=True)
    eps = host_noise(1, cfg["K"], cfg["T"], 1, seed=3)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL
    # the role-pipelined kernel at this horizon: rows in HBM as well (round 3), same costs
    pipe = make_engine(cfg, block_x=64, block_y=1, kernel_variant=2)
    pipe.injectNoise(eps)
    pipe.computeControl(cfg["x0"], 1)
    assert np.array_equal(pipe.getSampledCostSeq(), eng.getSampledCostSeq())
    pipe.close()
    ccfg = cartpole_cfg(K=300, T=2000)
    ccfg["colored"] = ([1.0], 0.97, 0.0)
    make_engine(ccfg, kernel_variant=2).close()  # colored noise + pipeline + rows in HBM
    make_engine(ccfg).close()


def test_vanilla_multi_iteration_and_closed_loop(gpu):
    """num_iters = 3 and a 15-step closed loop with slideControlSequence (examples/cartpole_example.cu:63-85)"""
    cfg = cartpole_cfg(K=1024, T=60, soft=True, num_iters=3)
    steps = 15
    eng, orc = make_engine(cfg), make_oracle(cfg)
    x_g = cfg["x0"].copy()
    x_c = cfg["x0"].copy()
    worst = 0.0
    for i in range(steps):
        eps = host_noise(3, cfg["K"], cfg["T"], 1, seed=100 + i)
        eng.injectNoise(eps)
        eng.computeControl(x_g, 1)
        orc.vanilla_compute_control(x_c, 1, eps)
        u_g, u_c = eng.getControlSeq(), orc.control()
        worst = max(worst, float(np.abs(u_g - u_c).max()))
        x_g, _ = eng.modelStep(x_g, u_g[0])
        x_c, _ = orc.model_step(x_c, u_c[0])
        eng.slideControlSequence(1)
        orc.vanilla_slide(1)
    # errors may compound through the closed loop (each step's u* feeds the next mean); still within the bar
    assert worst <= 5e-5, worst
    assert np.abs(x_g - x_c).max() <= 1e-4


def test_fused_philox_matches_oracle_generator(gpu):
    """the fused in-kernel generator draws exactly the oracle's Philox/Box-Muller stream, for every generation"""
    for gen in (0, 1, 7):
        a = m.philox_normal(1234567, gen, 640, 50, 2, 0, 640)
        b = po.philox_normal(1234567, gen, 640, 50, 2, 0, 640)
        assert ulp_diff(a, b).max() == 0
    a = m.philox_normal(99, 3, 1000, 33, 1, 333, 667)  # a shard in the middle, odd sizes
    b = po.philox_normal(99, 3, 1000, 33, 1, 333, 667)
    assert ulp_diff(a, b).max() == 0
    assert abs(float(a.mean())) < 0.02 and abs(float(a.std()) - 1.0) < 0.02


def test_fused_rng_mode_parity(gpu):
    """RNG mode end to end: the samples the kernel used == oracle samples from the oracle's generator; u* parity"""
    cfg = cartpole_cfg(K=2048, T=100, soft=True, num_iters=1)
    eng = make_engine(cfg, save_samples=True)
    orc = make_oracle(cfg)
    eps = np.stack([po.philox_normal(42, g, cfg["K"], cfg["T"], 1) for g in range(3)])
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps[:1])
    assert ulp_diff(eng.getSampledControls(), orc.samples()).max() == 0  # the clamped samples, bit for bit
    assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL
    # two more iterations in a second call: generations 1 and 2 (the generator offset advances like cuRAND's)
    eng.setNumIters(2)
    orc2 = make_oracle(dict(cfg, num_iters=2))
    orc2.set_nominal_control(orc.control())
    orc2.L.oracle_save_history  # (history stays zero: no slide in between)
    eng.computeControl(cfg["x0"], 1)
    orc2.vanilla_compute_control(cfg["x0"], 1, eps[1:])
    assert np.abs(eng.getControlSeq() - orc2.control()).max() <= U_TOL


def test_tube_double_integrator_parity(gpu):
    """Tube-MPPI, two systems per launch (reference: tests/controllers/tube_mppi_test.cu, examples/
    double_integrator_CORL2020.cu); nominal==actual invariance of rollout_kernel_tests.cu:181-198 is implied at step 0"""
    cfg = di_cfg(K=1024, T=50, tube=True, num_iters=2)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    x = cfg["x0"].copy()
    for i in range(4):
        eps = host_noise(2, cfg["K"], cfg["T"], 2, seed=7 + i)
        eng.injectNoise(eps)
        eng.computeControl(x, 1)
        orc.tube_compute_control(x, 1, eps)
        assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL
        assert np.abs(eng.getNominalControlSeq() - orc.nominal_control()).max() <= U_TOL
        assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= 1e-4
        assert np.abs(eng.getNominalStateSeq() - orc.nominal_state_traj()).max() <= 1e-4
        st, so = eng.getStats(), orc.stats()
        # u* differs from the oracle at the 1e-7 level (summation order), so from the second iteration on the costs
        # are no longer bit-identical; the baselines agree to fp32 accuracy
        assert abs(st.real_sys.baseline - so["baseline"][0]) <= 1e-5 * abs(so["baseline"][0])
        assert abs(st.nominal_sys.baseline - so["baseline"][1]) <= 1e-5 * abs(so["baseline"][1])
        assert st.nominal_state_used == so["nominal_state_used"]
        # disturb the actual state so that the two systems diverge
        x = x + np.array([0.05, -0.03, 0.2, -0.1], np.float32) * (i + 1)


def _tube_sticky_scenario(cfg, calls=8):
    """the oracle's closed loop with independent per-system noise and a zero threshold; returns per call (x, eps) and whether the
    call was of the kind the advisor found: the nominal system restarted from the actual state in an EARLIER pass of the call
    and was kept by the last one (nominal_state_used = 1 although the nominal trajectory starts at x)"""
    orc = make_oracle(cfg)
    orc.set_independent_noise(True)
    orc.set_controller_params(nominal_threshold=0.0)
    x = cfg["x0"].copy()
    out = []
    for i in range(calls):
        eps = np.stack([host_noise(cfg["num_iters"], cfg["K"], cfg["T"], 2, seed=100 + 2 * i + d) for d in range(2)], axis=1)
        orc.tube_compute_control(x, 1, eps)
        kept_after_takeover = orc.stats()["nominal_state_used"] == 1 and np.array_equal(orc.nominal_state_traj()[0], x)
        out.append((x.copy(), eps, kept_after_takeover, orc.control().copy(), orc.nominal_control().copy(),
                    orc.state_traj().copy(), orc.nominal_state_traj().copy(), orc.stats()["nominal_state_used"]))
        orc.tube_slide(1)
        x = x + np.array([0.05, -0.03, 0.2, -0.1], np.float32) * (1 + i % 3)
    return out


@pytest.mark.parametrize("low_latency", [True, False], ids=["flags", "copies"])
def test_tube_take_over_is_sticky_within_a_call(gpu, low_latency, monkeypatch):
    """num_iters = 2, independent noise per system, threshold 0, slide between the calls: when pass 0 restarts the nominal system
    from the actual state and pass 1 keeps it, the nominal state the NEXT call and the slide start from is the actual one
    (tube_mppi_controller.cu:268-277: nominal_state_trajectory_ persists across the passes of a call).  Round 5's flag
    hand-over looked at the last pass's choice only and kept the pre-call nominal state on the host."""
    if not low_latency:
        monkeypatch.setenv("MPPI_AMD_NO_SPIN", "1")
    cfg = di_cfg(K=256, T=30, tube=True, num_iters=2)
    rec = _tube_sticky_scenario(cfg)
    assert any(r[2] for r in rec[:-1])
    eng = make_engine(cfg)
    eng.setIndependentNoise(True)
    eng.setNominalThreshold(0.0)
    for x, eps, _, u, un, xs, xn, used in rec:
        eng.injectNoise(eps)
        eng.computeControl(x, 1)
        assert eng.getStats().nominal_state_used == used
        assert np.abs(eng.getControlSeq() - u).max() <= U_TOL
        assert np.abs(eng.getNominalControlSeq() - un).max() <= U_TOL
        assert np.abs(eng.getTargetStateSeq() - xs).max() <= 1e-4
        assert np.abs(eng.getNominalStateSeq() - xn).max() <= 1e-4
        eng.slideControlSequence(1)
    eng.close()


def test_tube_costs_identical_when_states_identical(gpu):
    """reference: tests/mppi_core/rollout_kernel_tests.cu:181-198 — same x0 for both systems => identical costs"""
    cfg = di_cfg(K=512, T=40, tube=True)
    eng = make_engine(cfg)
    eps = host_noise(1, cfg["K"], cfg["T"], 2)
    eng.injectNoise(eps)
    costs = eng.rolloutCosts(np.tile(cfg["x0"], (2, 1)), 1)
    assert np.array_equal(costs[0], costs[1])


def test_di_lds_contract_path(gpu):
    """blockDim.y > 1 (LDS + barrier path of the plugin contract) gives the same costs as the register path"""
    cfg = di_cfg(K=512, T=40, tube=False)
    eps = host_noise(1, cfg["K"], cfg["T"], 2)[0]
    _, _, g1, c, _ = _rollout_both(cfg, eps, block_x=64, block_y=1)
    _, _, g2, _, _ = _rollout_both(cfg, eps, block_x=64, block_y=2)
    assert ulp_diff(g1, c).max() == 0 and ulp_diff(g2, c).max() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [{}, {"kernel_variant": 1}, {"block_x": 64, "block_y": 4}], ids=["pipeline", "fused", "lds-contract"])
def test_time_specific_std_dev(gpu, kw):
    """time_specific_std_dev (gaussian.cuh:64-95, setGaussianControls gaussian.cu:21-43, LR cost :488-493): one sigma per
    (distribution, time step, control); costs 0 ulp against the oracle, with a non-zero likelihood-ratio coefficient and
    std_dev_decay so that both the decayed and the raw table are read"""
    cfg = cartpole_cfg_lr(K=1000, T=50)
    cfg["decay"] = 0.9
    sd = (2.0 + 3.0 * np.abs(np.sin(np.arange(cfg["T"], dtype=np.float32) * 0.37))).reshape(cfg["T"], 1)
    eng, orc = make_engine(cfg, save_samples=True, **kw), make_oracle(cfg)
    eng.setTimeSpecificStdDev(sd)
    orc.set_time_specific_std_dev(sd[None])
    mean = (0.5 * np.cos(np.arange(cfg["T"], dtype=np.float32) * 0.2)).reshape(cfg["T"], 1)
    eps = host_noise(1, cfg["K"], cfg["T"], 1, seed=8)[0]
    eng.updateImportanceSampler(mean)
    eng.injectNoise(eps)
    g = eng.rolloutCosts(cfg["x0"], 1)
    v = orc.set_gaussian_controls(mean[None], eps, 1, 0)
    c, vc = orc.rollout_costs(cfg["x0"], mean[None], v)
    assert ulp_diff(g, c).max() == 0
    assert ulp_diff(eng.getSampledControls(), vc).max() == 0
    # second optimisation iteration: sigma * decay
    eng2, orc2 = make_engine(dict(cfg, num_iters=2), **kw), make_oracle(dict(cfg, num_iters=2))
    eng2.setTimeSpecificStdDev(sd)
    orc2.set_time_specific_std_dev(sd[None])
    e2 = host_noise(2, cfg["K"], cfg["T"], 1, seed=9)
    eng2.injectNoise(e2)
    eng2.computeControl(cfg["x0"], 1)
    orc2.vanilla_compute_control(cfg["x0"], 1, e2)
    assert np.abs(eng2.getControlSeq() - orc2.control()).max() <= U_TOL
    # switching it off restores the per-control sigma
    eng.setTimeSpecificStdDev(None)
    orc.set_time_specific_std_dev(None)
    eng.injectNoise(eps)
    g = eng.rolloutCosts(cfg["x0"], 1)
    c, _ = orc.rollout_costs(cfg["x0"], mean[None], orc.set_gaussian_controls(mean[None], eps, 1, 0))
    assert ulp_diff(g, c).max() == 0


@pytest.mark.gpu
def test_time_specific_std_dev_two_systems(gpu):
    """Tube (two distributions, folded kernel: the distribution index differs between the lanes of a wave)"""
    cfg = di_cfg(K=512, T=33, tube=True)
    cfg["control_cost_coeff"] = [0.4, 0.2]
    sd = 0.5 + np.random.default_rng(1).random((2, cfg["T"], 2)).astype(np.float32)
    for kw in ({}, {"kernel_variant": 1}):
        eng, orc = make_engine(cfg, **kw), make_oracle(cfg)
        eng.setTimeSpecificStdDev(sd)
        orc.set_time_specific_std_dev(sd)
        eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=4)
        eng.injectNoise(eps)
        eng.computeControl(cfg["x0"], 1)
        orc.tube_compute_control(cfg["x0"], 1, eps)
        assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0
        assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [{}, {"kernel_variant": 1}, {"block_x": 64, "block_y": 1}, {"block_x": 32, "block_y": 2}],
                         ids=["folded-pipeline", "fused", "pipeline-64x1x2", "lds-contract"])
def test_independent_noise_per_distribution(gpu, kw):
    """use_same_noise_for_all_distributions = false (sampling_distribution.cuh:20; gaussian.cu:378-394): the two systems of
    Tube-MPPI draw their own noise — slab d of the injected buffer, Philox stream d in the generator mode"""
    cfg = di_cfg(K=1000, T=33, tube=True)
    eng, orc = make_engine(cfg, save_samples=True, **kw), make_oracle(cfg)
    eng.setIndependentNoise(True)
    orc.set_independent_noise(True)
    eps = np.random.default_rng(12).standard_normal((1, 2, cfg["K"], cfg["T"], 2)).astype(np.float32)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    orc.tube_compute_control(cfg["x0"], 1, eps)
    assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL
    assert np.abs(eng.getNominalControlSeq() - orc.nominal_control()).max() <= U_TOL
    v = eng.getSampledControls()
    assert not np.array_equal(v[0], v[1])
    # generator mode: distribution d == Philox stream d of the oracle's generator
    eng2, orc2 = make_engine(cfg, **kw), make_oracle(cfg)
    eng2.setIndependentNoise(True)
    orc2.set_independent_noise(True)
    eng2.setSeed(77)
    eng2.computeControl(cfg["x0"], 1)
    epsp = np.stack([po.philox_normal(77, 0, cfg["K"], cfg["T"], 2, stream=d) for d in range(2)])[None]
    orc2.tube_compute_control(cfg["x0"], 1, epsp)
    assert ulp_diff(eng2.getSampledCostSeq(), orc2.costs()).max() == 0
    assert np.abs(eng2.getControlSeq() - orc2.control()).max() <= U_TOL
    # back to shared noise
    eng2.setIndependentNoise(False)
    orc2.set_independent_noise(False)
    e1 = host_noise(1, cfg["K"], cfg["T"], 2, seed=2)
    eng2.injectNoise(e1)
    eng2.computeControl(cfg["x0"], 1)
    orc2.tube_compute_control(cfg["x0"], 1, e1)
    assert np.abs(eng2.getControlSeq() - orc2.control()).max() <= U_TOL
