"""Horizons whose sample rows do not fit the LDS: the rows-in-HBM form of the fused rollout kernel (the reference keeps its
samples in global memory always and has no horizon limit, sampling_distributions/sampling_distribution.cu:169-205)."""
import os

import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import autorally_cfg, cartpole_cfg, cartpole_cfg_lr, di_cfg, host_noise, make_engine, make_oracle, ulp_diff

pytestmark = pytest.mark.gpu


def _engine_with_hbm_rows(cfg, **kw):
    os.environ["MPPI_AMD_ROWS_IN_HBM"] = "1"
    try:
        return make_engine(cfg, **kw)
    finally:
        del os.environ["MPPI_AMD_ROWS_IN_HBM"]


@pytest.mark.parametrize("mk", [lambda: cartpole_cfg_lr(K=1000, T=50), lambda: di_cfg(K=512, T=33, tube=True),
                                lambda: autorally_cfg(K=200, T=37)], ids=["cartpole", "di-tube", "autorally-mfma"])
@pytest.mark.parametrize("noise", ["injected", "philox"])
def test_hbm_rows_bit_identical_to_lds_rows(gpu, mk, noise):
    """same kernel, same arithmetic, rows in HBM instead of LDS: costs, samples and u* are the same bits (ragged last block,
    two systems per launch, replicated-lane dynamics)"""
    cfg = mk()
    a = make_engine(cfg, kernel_variant=1, save_samples=True)
    b = _engine_with_hbm_rows(cfg, kernel_variant=1, save_samples=True)
    x0 = np.tile(cfg["x0"], (cfg["D"], 1))
    for eng in (a, b):
        if noise == "injected":
            eng.injectNoise(host_noise(1, cfg["K"], cfg["T"], eng.CONTROL_DIM))
        eng.uploadState(x0)
        eng.optimize(2)
    assert np.array_equal(a.getSampledCostSeq(), b.getSampledCostSeq())
    assert np.array_equal(a.getSampledControls(), b.getSampledControls())
    assert np.array_equal(a.getOptimalControlSeq(), b.getOptimalControlSeq())


@pytest.mark.parametrize("mk", [lambda: cartpole_cfg_lr(K=1000, T=50), lambda: cartpole_cfg(K=333, T=7, soft=True),
                                lambda: di_cfg(K=512, T=33, tube=True), lambda: autorally_cfg(K=200, T=37),
                                lambda: autorally_cfg(K=130, T=2)],
                         ids=["cartpole", "cartpole-T7", "di-tube", "autorally-mfma", "autorally-T2"])
@pytest.mark.parametrize("noise", ["injected", "philox"])
def test_pipelined_kernels_hbm_rows_bit_identical_to_lds_rows(gpu, mk, noise):
    """the role-pipelined kernels with the rows in HBM (round 3: the dynamics waves fetch a trip ahead, the clamped control
    reaches the cost waves through the output ring): costs, samples and u* are the bits of the LDS-row kernels — one lane per
    rollout (two samplers), two systems per block, replicated-lane dynamics, horizons shorter than a trip"""
    cfg = mk()
    a = make_engine(cfg, kernel_variant=2, save_samples=True)
    b = _engine_with_hbm_rows(cfg, kernel_variant=2, save_samples=True)
    x0 = np.tile(cfg["x0"], (cfg["D"], 1))
    for eng in (a, b):
        if noise == "injected":
            eng.injectNoise(host_noise(1, cfg["K"], cfg["T"], eng.CONTROL_DIM))
        eng.uploadState(x0)
        eng.optimize(2)
    assert np.array_equal(a.getSampledCostSeq(), b.getSampledCostSeq())
    assert np.array_equal(a.getSampledControls(), b.getSampledControls())
    assert np.array_equal(a.getOptimalControlSeq(), b.getOptimalControlSeq())
    a.close()
    b.close()


@pytest.mark.parametrize("mk,T", [(cartpole_cfg, 2000), (autorally_cfg, 1000)], ids=["cartpole-T2000", "autorally-T1000"])
def test_pipelined_kernels_long_horizon_vs_oracle(gpu, mk, T):
    """horizons whose rows do not fit the LDS stay on the role-pipelined kernels (the default; rows in HBM) and agree with
    the oracle; the fused kernel on the same rows gives the same bits"""
    cfg = mk(K=256, T=T, soft=True) if mk is cartpole_cfg else mk(K=256, T=T)
    eng, orc = make_engine(cfg, kernel_variant=2), make_oracle(cfg)
    C = eng.CONTROL_DIM
    eps = host_noise(1, cfg["K"], cfg["T"], C)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= 1e-5
    costs = eng.getSampledCostSeq().copy()
    eng.close()
    auto = make_engine(cfg)  # no request: the pipelined kernel as well
    fused = make_engine(cfg, kernel_variant=1)
    for e in (auto, fused):
        e.injectNoise(eps)
        e.computeControl(cfg["x0"], 1)
    assert np.array_equal(auto.getSampledCostSeq(), costs) and np.array_equal(fused.getSampledCostSeq(), costs)
    assert np.array_equal(auto.getControlSeq(), fused.getControlSeq())
    auto.close()
    fused.close()


def test_cartpole_T5000_vs_oracle(gpu):
    """T = 5000: 20 KB of samples per rollout — no block shape fits the LDS; mppi_create moves the rows to HBM by itself"""
    cfg = cartpole_cfg(K=2048, T=5000, soft=True)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    eps = host_noise(1, cfg["K"], cfg["T"], 1)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= 1e-5
    assert eng.getStats().real_sys.baseline == orc.stats()["baseline"][0]
    # and the in-kernel Philox stream at the same horizon
    eng.injectNoise(None)
    eng.setSeed(5)
    eng.uploadState(cfg["x0"])
    eng.updateImportanceSampler(np.zeros((cfg["T"], 1), np.float32))
    eng.optimize(1)
    u = orc.iterate(cfg["x0"], np.zeros((cfg["T"], 1), np.float32), po.philox_normal(5, 0, cfg["K"], cfg["T"], 1))[0]
    assert np.abs(eng.getOptimalControlSeq()[0] - u).max() <= 1e-5


def test_colored_sampler_T1000_vs_oracle(gpu):
    """the colored-noise sampler at a horizon whose rows do not fit the LDS (round 3: its prologue GEMM writes the tiles straight
    into the HBM rows; only the table staging tiles stay in LDS) — the reference keeps its samples in global memory for every
    sampler (sampling_distributions/sampling_distribution.cu:169-205) and has no horizon limit"""
    from common import host_spectrum
    cfg = cartpole_cfg(K=256, T=1000, soft=True)
    cfg["colored"] = ([1.0], 0.97, 0.0)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    z = host_spectrum(1, cfg["K"], cfg["T"], 1, seed=3)
    eng.injectNoise(z)
    eng.computeControl(cfg["x0"], 1)
    orc.colored_compute_control(cfg["x0"], 1, z, *cfg["colored"])
    assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= 1e-5
    eng.close()
    # and the same kernels with the rows forced into HBM at a horizon that would fit: the same bits as with LDS rows
    cfg = cartpole_cfg(K=300, T=64, soft=True)
    cfg["colored"] = ([1.0], 0.97, 0.0)
    for variant in (1, 2):  # fused, role-pipelined
        a = make_engine(cfg, kernel_variant=variant)
        b = _engine_with_hbm_rows(cfg, kernel_variant=variant)
        for eng in (a, b):
            eng.uploadState(cfg["x0"])
            eng.optimize(2)
        assert np.array_equal(a.getSampledCostSeq(), b.getSampledCostSeq())
        assert np.array_equal(a.getOptimalControlSeq(), b.getOptimalControlSeq())
        a.close()
        b.close()


def test_rmppi_T1000_vs_oracle(gpu):
    """Robust MPPI at a horizon whose rows of even 32 rollouts x 2 systems overflow the LDS: (64, 1, 2) block, rows in HBM"""
    cfg = di_cfg(K=320, T=1000, tube=True)
    cfg["control_cost_coeff"] = [0.3, 0.2]
    cfg["ranges"] = [[-3.0, 3.0], [-3.0, 3.0]]
    eng = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], cfg["alpha"], 1, seed=42,
                                 save_samples=True)
    eng.setDynamicsParams(cfg["dyn"])
    eng.setCostParams(cfg["cost"])
    eng.setControlRanges(cfg["ranges"])
    eng.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"])
    eng.setRMPPIParams(40.0, 9, 32)
    orc = make_oracle(cfg)
    rob = po.RobustOracle(orc, 40.0, 9, 32)
    T, K = cfg["T"], cfg["K"]
    g = np.random.default_rng(1).uniform(-0.4, 0.4, (T, 4, 2)).astype(np.float32)
    eng.setFeedbackGains(g)
    rob.set_gains(g)
    mean = (0.3 * np.sin(np.arange(T * 2, dtype=np.float32) * 0.02)).reshape(T, 2)
    eng.updateImportanceSampler(mean)
    eps = host_noise(1, K, T, 2)[0]
    eng.injectNoise(eps)
    x0 = np.stack([cfg["x0"], cfg["x0"] + np.array([0.3, -0.2, 0.1, 0.05], np.float32)])
    got = eng.rolloutCosts(x0, 1)
    means = np.tile(mean, (2, 1, 1))
    v = orc.set_gaussian_controls(means, eps, 1, 0)
    want, v_fb = rob.rollout_costs(x0, means, v)
    assert ulp_diff(got, want).max() == 0
    assert ulp_diff(eng.getSampledControls(), v_fb).max() == 0
    eng.close()


def _engine_with_finalize_scratch(cfg, **kw):
    os.environ["MPPI_AMD_FINALIZE_SCRATCH"] = "1"
    try:
        return make_engine(cfg, **kw)
    finally:
        del os.environ["MPPI_AMD_FINALIZE_SCRATCH"]


@pytest.mark.parametrize("mk", [lambda: cartpole_cfg(K=512, T=60, soft=True), lambda: di_cfg(K=512, T=33, tube=True),
                                lambda: autorally_cfg(K=256, T=37)], ids=["cartpole", "di-tube", "autorally-wave-form"])
def test_postprocessing_with_the_sequence_in_hbm_is_bit_identical(gpu, mk):
    """smoothing buffer and control sequence of the post-processing kernels in HBM (long horizons) instead of LDS: controls and
    trajectories are the same bits (one lane per rollout, two systems, the NN model's one-rollout-per-wave form)"""
    cfg = mk()
    got = []
    for make in (make_engine, _engine_with_finalize_scratch):
        eng = make(cfg)
        x = cfg["x0"].copy()
        eng.computeControl(x, 1)
        eng.slideControlSequence(1)
        eng.computeControl(x, 1)
        got.append((eng.getControlSeq().copy(), eng.getTargetStateSeq().copy(), eng.getTargetOutputSeq().copy()))
        eng.close()
    for a, b in zip(*got):
        assert np.array_equal(a, b)


def test_cartpole_T25000_postprocessing_vs_oracle(gpu):
    """T * C = 25 000: neither the sample rows nor the control sequence fit the LDS — rows and post-processing buffers in HBM;
    control sequence and state trajectory against the oracle"""
    cfg = cartpole_cfg(K=128, T=25000, soft=True)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    eps = host_noise(1, cfg["K"], cfg["T"], 1)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= 1e-5
    # (a 25 000-step open-loop trajectory of the cart-pole amplifies the 1e-5 of the controls: compare its first thousand steps)
    assert np.abs(eng.getTargetStateSeq()[:1000] - orc.state_traj()[:1000]).max() <= 1e-3
    eng.close()
