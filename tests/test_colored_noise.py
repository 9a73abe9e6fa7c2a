"""ColoredNoiseDistribution + ColoredMPPI (BASELINE config 5 sampler)."""
import os
import sys

import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import (bicycle_lstm_cfg, cartpole_cfg, di_cfg, host_spectrum, make_engine, make_oracle, ulp_diff)


def _numpy_reference(z, exponents, decay, fmin, offset_t):
    """scripts/colored_noise.py:10-106 / colored_noise.cu:294-392 in numpy float64 with numpy.fft.irfft as the C2R"""
    K, C, F, _ = z.shape
    T = F - 1
    N = 2 * T
    out = np.zeros((K, T, C))
    for c in range(C):
        f = np.arange(F, dtype=np.float32) / np.float32(N)
        cutoff = max(np.float32(fmin), np.float32(1.0) / np.float32(N))
        ok = np.nonzero(f >= cutoff)[0]
        if ok.size:
            f[:ok[0]] = f[ok[0]]
        w = np.power(f, np.float32(-exponents[c] / 2.0), dtype=np.float32)
        sig = np.float32(2.0) * np.sqrt(np.float32((w[1:-1] ** 2).sum(dtype=np.float32) + (w[-1] * np.float32(0.5)) ** 2)) / np.float32(N)
        spec = (z[:, c, :, 0] * w).astype(np.float64) + 1j * (z[:, c, :, 1] * w).astype(np.float64)
        spec[:, 0] = spec[:, 0].real
        spec[:, -1] = spec[:, -1].real
        x = np.fft.irfft(spec, n=N, axis=1) * N  # cuFFT does not normalise
        d = np.zeros(T) if decay == 0 else np.power(np.float32(decay), np.arange(T, dtype=np.float32)).astype(np.float64)
        out[:, :, c] = (x[:, :T] - x[:, offset_t:offset_t + 1] * d[None, :]) / float(sig * 2 * T)
    return out


# ------------------------------------------------------------------ CPU: pin the oracle --------------------------------
@pytest.mark.parametrize("T,exps,decay,fmin,stride", [(50, [1.0, 0.5], 0.97, 0.0, 1), (33, [0.0, 2.0], 0.0, 0.0, 0),
                                                       (100, [1.0, 1.0], 0.9, 0.05, 3)])
def test_oracle_definition_matches_numpy_irfft(T, exps, decay, fmin, stride):
    z = host_spectrum(1, 64, T, 2, seed=5)[0]
    got = po.colored_noise(z, exps, decay, fmin, stride, flavour="definition")
    want = _numpy_reference(z, exps, decay, fmin, stride)
    assert np.abs(got - want).max() <= 3e-6 * max(1.0, np.abs(want).max())


# ---- pinned on the REFERENCE's own implementation: golden vectors made by importing scripts/colored_noise.py -------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colored_noise_reference.npz")
GOLDEN_CASES = ["pink_T50", "white_brown_T33", "cutoff_T100", "config5_T200"]


def _golden(name):
    d = np.load(GOLDEN)
    return d[name + "_z"], d[name + "_y"], list(d[name + "_exponents"]), float(d[name + "_fmin"][0])


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("flavour", ["definition", "gemm"])
def test_oracle_matches_reference_script_golden(name, flavour):
    """tests/golden/make_colored_noise_reference.py ran the reference's scripts/colored_noise.py:powerlaw_psd_gaussian with
    recorded N(0,1) draws; the oracle, fed the same draws as its spectrum (offset removal off: decay 0), must reproduce the
    script's first T samples.  3e-6: fp32 evaluation against the script's float64."""
    z, y, exps, fmin = _golden(name)
    got = po.colored_noise(z, exps, 0.0, fmin, 0, flavour=flavour)
    assert got.shape == y.shape
    assert np.abs(got - y).max() <= 3e-6 * max(1.0, np.abs(y).max()), np.abs(got - y).max()


@pytest.mark.skipif(not os.path.exists("/root/reference/scripts/colored_noise.py"), reason="reference tree not present")
def test_golden_is_what_the_reference_script_produces_now():
    """in the build container: re-import the reference script and regenerate one case -> identical to the committed fixture"""
    sys.path.insert(0, os.path.dirname(GOLDEN))
    import make_colored_noise_reference as mk
    mod = mk.load_reference_module()
    for i, (name, K, T, exps, fmin) in enumerate(mk.CASES):
        z, y = mk.run_case(mod, K, T, exps, fmin, seed=1000 + i)
        zg, yg, _, _ = _golden(name)
        assert np.array_equal(z, zg) and np.array_equal(y, yg)


@pytest.mark.parametrize("T,exps,decay,fmin,stride", [(50, [1.0, 0.5], 0.97, 0.0, 1), (200, [1.0, 1.0], 0.97, 0.0, 1),
                                                       (33, [0.0, 2.0], 0.0, 0.1, 2)])
def test_oracle_gemm_flavour_matches_definition(T, exps, decay, fmin, stride):
    """the folded-table fp32 fma-chain form (what the engine computes) against the step-by-step definition"""
    z = host_spectrum(1, 32, T, 2, seed=6)[0]
    a = po.colored_noise(z, exps, decay, fmin, stride, flavour="definition")
    b = po.colored_noise(z, exps, decay, fmin, stride, flavour="gemm")
    assert np.abs(a - b).max() <= 2e-6 * max(1.0, np.abs(a).max())


def test_colored_noise_statistics():
    """reference: tests/sampling_distributions/colored_noise_tests.cu:98-209 — statistical checks only: unit variance
    (that is what sigma normalises to), zero sample at the offset index, and more low-frequency power for beta > 0"""
    T, K = 128, 512
    z = po.philox_spectrum(42, 0, K, T, 2)
    white = po.colored_noise(z, [0.0, 0.0], 0.0, 0.0, 0)
    red = po.colored_noise(z, [0.0, 2.0], 0.0, 0.0, 0)
    assert abs(white[:, :, 0].std() - 1.0) < 0.03
    assert abs(red[:, :, 1].std() - 1.0) < 0.15
    w_spec = np.abs(np.fft.rfft(white[:, :, 0], axis=1)) ** 2
    r_spec = np.abs(np.fft.rfft(red[:, :, 1], axis=1)) ** 2
    lo, hi = slice(1, 8), slice(40, 64)
    assert r_spec[:, lo].mean() / r_spec[:, hi].mean() > 20 * w_spec[:, lo].mean() / w_spec[:, hi].mean()
    with_offset = po.colored_noise(z, [1.0, 1.0], 1.0, 0.0, 5)
    assert np.abs(with_offset[:, 5, :]).max() < 2e-5  # x[s] - x[s] * decay^s with decay = 1
    # weights: f^(-beta/2) with the below-cutoff entries replaced (colored_noise.cu:302-322)
    w, sigma = po.colored_weights(50, [1.0, 0.0], fmin=0.035)
    f = np.arange(51) / 100.0
    np.testing.assert_allclose(w[0, 4:], f[4:] ** -0.5, rtol=1e-6)
    np.testing.assert_allclose(w[0, :4], f[4] ** -0.5, rtol=1e-6)
    np.testing.assert_allclose(w[1], 1.0)
    np.testing.assert_allclose(sigma[1], 2 * np.sqrt(49 + 0.25) / 100, rtol=1e-6)


def test_philox_spectrum_is_standard_normal_and_shard_invariant():
    z = po.philox_spectrum(7, 3, 256, 60, 2)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    part = po.philox_spectrum(7, 3, 256, 60, 2, 100, 140)
    assert np.array_equal(z[100:140], part)


# ------------------------------------------------------------------ GPU ------------------------------------------------
def _colored_cartpole(K=1024, T=50, **kw):
    cfg = cartpole_cfg(K=K, T=T, soft=True, **kw)
    cfg["colored"] = ([1.0], 0.97, 0.0)
    return cfg


def _colored_bicycle(K=512, T=40, **kw):
    cfg = bicycle_lstm_cfg(K=K, T=T, **kw)
    cfg["colored"] = ([1.0, 0.5], 0.97, 0.0)
    return cfg


@pytest.mark.gpu
@pytest.mark.parametrize("mk,T,stride", [(_colored_cartpole, 50, 1), (_colored_cartpole, 37, 0), (_colored_bicycle, 40, 2),
                                          (_colored_bicycle, 200, 1),
                                          # the radix-4 form's edges: its smallest horizon, one exactly full time block per class,
                                          # five time blocks per class (two passes of MAX_TB4 = 4), offset sample 0 and late
                                          (_colored_cartpole, 16, 0), (_colored_cartpole, 64, 1), (_colored_cartpole, 260, 3),
                                          (_colored_bicycle, 128, 1)])
def test_colored_noise_generator_bit_exact(gpu, mk, T, stride):
    """the MFMA prologue GEMM == the oracle's fp32 fma chains, bit for bit, for injected and Philox spectra"""
    cfg = mk(K=200, T=T)  # K not a multiple of 64: ragged last block
    eng = make_engine(cfg)
    C = eng.CONTROL_DIM
    exps, decay, fmin = cfg["colored"]
    z = host_spectrum(1, cfg["K"], T, C, seed=9)[0]
    eng.injectNoise(z)
    got = eng.sampleNoise(stride)
    want = po.colored_noise(z, exps, decay, fmin, stride, flavour="gemm")
    assert ulp_diff(got, want).max() == 0
    defn = po.colored_noise(z, exps, decay, fmin, stride, flavour="definition")
    assert np.abs(got - defn).max() <= 2e-6 * max(1.0, np.abs(defn).max())
    # in-kernel Philox spectrum
    eng.injectNoise(None)
    eng.setSeed(1234)
    got = eng.sampleNoise(stride)
    zp = po.philox_spectrum(1234, 0, cfg["K"], T, C)
    assert ulp_diff(got, po.colored_noise(zp, exps, decay, fmin, stride, flavour="gemm")).max() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_engine_sampler_matches_reference_script_golden(gpu, name):
    """the engine's in-kernel MFMA GEMM, fed the recorded draws of the reference's scripts/colored_noise.py, reproduces the
    script's output (two controls: the double integrator instantiation; offset removal off)"""
    z, y, exps, fmin = _golden(name)
    K, T = y.shape[0], y.shape[1]
    cfg = di_cfg(K=K, T=T, tube=False)
    cfg["colored"] = (exps, 0.0, fmin)
    eng = make_engine(cfg)
    eng.injectNoise(z)
    got = eng.sampleNoise(0)
    assert np.abs(got - y).max() <= 3e-6 * max(1.0, np.abs(y).max()), np.abs(got - y).max()


@pytest.mark.gpu
@pytest.mark.parametrize("mk,kw", [(_colored_cartpole, {}), (_colored_cartpole, {"kernel_variant": 1}),
                                   (_colored_cartpole, {"block_x": 64, "block_y": 4}), (_colored_bicycle, {}),
                                   (_colored_bicycle, {"kernel_variant": 1}),
                                   (_colored_bicycle, {"block_x": 16, "block_y": 8})])
def test_colored_rollout_costs_bit_exact(gpu, mk, kw):
    cfg = mk()
    eng, orc = make_engine(cfg, **kw), make_oracle(cfg)
    C, T, K = eng.CONTROL_DIM, cfg["T"], cfg["K"]
    exps, decay, fmin = cfg["colored"]
    z = host_spectrum(1, K, T, C, seed=3)[0]
    mean = (0.2 * np.sin(np.arange(T * C, dtype=np.float32) * 0.3)).reshape(T, C)
    eng.updateImportanceSampler(mean)
    eng.injectNoise(z)
    g = eng.rolloutCosts(cfg["x0"], 1)
    eps = po.colored_noise(z, exps, decay, fmin, 1, flavour="gemm")
    v = orc.set_gaussian_controls(mean[None], eps, 1, 0)
    c, _ = orc.rollout_costs(cfg["x0"], mean[None], v)
    assert np.isfinite(g).all()
    assert ulp_diff(g, c).max() == 0


def test_oracle_tsallis_weights_against_float64():
    """core/mppi_common.cu:968-985: w = (S - rho < gamma) ? exp(log(1 - (S - rho)/gamma) / (r - 1)) : 0"""
    cfg = cartpole_cfg(K=512, T=20, soft=True)
    orc = make_oracle(cfg)
    orc.set_colored_mppi_params(gamma=60.0, r_exp=1.5)
    eps = np.random.default_rng(3).standard_normal((512, 20, 1)).astype(np.float32)
    orc.iterate(cfg["x0"], np.zeros((20, 1), np.float32), eps)
    S = orc.costs()[0].astype(np.float64)
    d = S - S.min()
    want = np.where(d < 60.0, np.exp(np.log(np.maximum(1.0 - d / 60.0, 1e-300)) / 0.5), 0.0)
    np.testing.assert_allclose(orc.weights()[0], want, rtol=1e-4, atol=1e-7)  # fp32 costs ~2e4: 1 - d/gamma loses digits near the cut-off
    assert (want == 0).sum() > 0 and (want > 0.5).sum() > 1, "the case should have cut-off and heavy rollouts"


@pytest.mark.gpu
@pytest.mark.parametrize("tsallis,leash", [(True, False), (False, True), (True, True)])
def test_colored_mppi_tsallis_and_leash_parity(gpu, tsallis, leash):
    """ColoredMPPI's Tsallis weights (global baseline -> weights -> weighted mean of the samples in HBM) and state leash
    against the oracle in closed loop; reference: colored_mppi_controller.cu:150-156, 198-206"""
    cfg = _colored_cartpole(K=2048, T=60)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    exps, decay, fmin = cfg["colored"]
    kw = dict(gamma=400.0 if tsallis else 0.0, r_exp=1.7 if tsallis else 0.0,
              state_leash_dist=np.array([0.05, 0.2, 0.02, 0.3], np.float32), leash_active=leash, leash_jump=1)
    eng.setColoredMPPIParams(**kw)
    orc.set_colored_mppi_params(kw["gamma"], kw["r_exp"], kw["state_leash_dist"], leash, 1)
    x = cfg["x0"].copy()
    for i in range(4):
        z = host_spectrum(1, cfg["K"], cfg["T"], 1, seed=40 + i)
        eng.injectNoise(z)
        eng.computeControl(x, 1)
        orc.colored_compute_control(x, 1, z, exps, decay, fmin)
        assert np.abs(eng.getControlSeq() - orc.control()).max() <= 1e-5
        assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= 1e-4
        st, so = eng.getStats().real_sys, orc.stats()
        assert st.baseline == so["baseline"][0] or i > 0
        assert abs(st.normalizer - so["normalizer"][0]) <= 1e-5 * so["normalizer"][0]
        # the measured state drifts away from the model's prediction: the leash has something to do
        x, _ = orc.model_step(x, orc.control()[0])
        x = x + np.array([0.2, -0.4, 0.1, 0.5], np.float32)
        eng.slideControlSequence(1)
        orc.vanilla_slide(1)


@pytest.mark.gpu
def test_colored_mppi_params_argument_checks(gpu):
    eng = make_engine(_colored_cartpole(K=256, T=20))
    with pytest.raises(m.MPPIError) as e:
        eng.setColoredMPPIParams(gamma=10.0, r_exp=1.0)  # r = 1: division by zero in the exponent
    assert e.value.status == 1
    van = m.VanillaMPPIController("cartpole", 128, 10, 0.02, 1.0)
    with pytest.raises(m.MPPIError) as e:
        van._check(van._lib.mppi_set_colored_mppi_params(van._h, 1.0, 2.0, None, 0, 1))
    assert e.value.status == 7


@pytest.mark.gpu
@pytest.mark.parametrize("mk", [_colored_cartpole, _colored_bicycle])
def test_colored_mppi_compute_control_parity(gpu, mk):
    """ColoredMPPI closed loop (colored_mppi_controller.cu:134-240) against the oracle, control L-inf <= 1e-5"""
    cfg = mk(num_iters=2)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    C, T, K = eng.CONTROL_DIM, cfg["T"], cfg["K"]
    exps, decay, fmin = cfg["colored"]
    x = cfg["x0"].copy()
    for i in range(3):
        z = host_spectrum(2, K, T, C, seed=20 + i)
        eng.injectNoise(z)
        eng.computeControl(x, 1)
        orc.colored_compute_control(x, 1, z, exps, decay, fmin)
        assert np.abs(eng.getControlSeq() - orc.control()).max() <= 1e-5
        assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= 1e-4
        x, _ = orc.model_step(x, orc.control()[0])
        eng.slideControlSequence(1)
        orc.vanilla_slide(1)


@pytest.mark.gpu
def test_colored_params_rejected_on_gaussian_handle(gpu):
    c = m.VanillaMPPIController("cartpole", 128, 10, 0.02, 1.0)
    with pytest.raises(m.MPPIError) as e:
        c._check(c._lib.mppi_set_colored_noise_params(c._h, np.ones(1, np.float32), 0.9, 0.0))
    assert e.value.status == 7
