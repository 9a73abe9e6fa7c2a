"""HIP engine vs CPU oracle at the FULL sizes of BASELINE.json's configs (north_star: control-sequence L-inf <= 1e-5).

Small-K tests do not exercise what these do: one block per CU (256 blocks at K = 16384), the 256- / 1024-record merge of
combineKernel, LDS occupancy at T = 150 / 200, ragged tails.  Every test runs ONE computeControl through the C ABI and the
same call on the oracle (OpenMP over rollouts), on the same injected noise or on the same Philox stream, and asserts
  trajectory costs   0 ulp          (reference's own bar: 1e-4 relative, tests/mppi_core/rollout_kernel_tests.cu:200-261)
  baseline rho       exact
  normaliser eta     <= 1e-6 relative
  u*                 <= 1e-5 L-inf  (fp32 tolerance of north_star)
  state trajectory   <= 1e-4
and, for the sharded variants, the same after an 8-way split of K over eight handles with the exchange done by hand.
"""
import ctypes as C

import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import (autorally_cfg, bicycle_lstm_cfg, cartpole_cfg, di_cfg, host_noise, host_spectrum, make_engine,
                    make_oracle, ulp_diff)

pytestmark = pytest.mark.gpu

U_TOL = 1e-5      # north_star: control-sequence L-inf vs reference
ETA_RTOL = 1e-6
X_TOL = 1e-4


def _check_system(eng_stats, orc_stats, z):
    assert eng_stats.baseline == orc_stats["baseline"][z], (eng_stats.baseline, orc_stats["baseline"][z])
    eta = float(orc_stats["normalizer"][z])
    assert abs(eng_stats.normalizer - eta) <= ETA_RTOL * eta, (eng_stats.normalizer, eta)


def _check_vanilla(eng, orc, exact_costs=True):
    if exact_costs:
        dc = int(ulp_diff(eng.getSampledCostSeq(), orc.costs()).max())
        assert dc == 0, "trajectory costs differ by %d ulp" % dc
    else:  # after a second optimisation iteration the mean already carries the ~1e-7 difference of the first u*
        np.testing.assert_allclose(eng.getSampledCostSeq(), orc.costs(), rtol=1e-5)
        return float(np.abs(eng.getControlSeq() - orc.control()).max())
    _check_system(eng.getStats().real_sys, orc.stats(), 0)
    du = float(np.abs(eng.getControlSeq() - orc.control()).max())
    assert du <= U_TOL, du
    assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= X_TOL
    return du


# ------------------------------------------------------------------ config 2: Cartpole K=16384 T=100 ------------------
@pytest.mark.parametrize("variant", [0, 1], ids=["pipeline", "fused"])
@pytest.mark.parametrize("noise", ["injected", "philox"])
@pytest.mark.parametrize("soft", [False, True], ids=["lambda0.25", "lambda200"])
def test_cartpole_16384x100_vs_oracle(gpu, variant, noise, soft):
    """BASELINE headline config (examples/cartpole_example.cu parameters; lambda 0.25 as upstream, and lambda 200 where
    thousands of rollouts carry weight), both kernel structures, injected eps and the in-kernel Philox stream"""
    for num_iters in (1, 2):
        cfg = cartpole_cfg(K=16384, T=100, soft=soft, num_iters=num_iters)
        eng, orc = make_engine(cfg, kernel_variant=variant), make_oracle(cfg)
        if noise == "injected":
            eps = host_noise(num_iters, cfg["K"], cfg["T"], 1)
            eng.injectNoise(eps)
        else:
            eng.setSeed(42)  # generation g of the Philox stream = optimisation iteration g
            eps = np.stack([po.philox_normal(42, g, cfg["K"], cfg["T"], 1) for g in range(num_iters)])
        eng.computeControl(cfg["x0"], 1)
        orc.vanilla_compute_control(cfg["x0"], 1, eps)
        if num_iters == 1:
            _check_vanilla(eng, orc)
        else:
            assert _check_vanilla(eng, orc, exact_costs=False) <= U_TOL


# ------------------------------------------------------------------ config 4: AutoRally-NN K=16384 T=150 --------------
def autorally_cfg_survey_literal(K=16384, T=150):
    """SURVEY.md §8d config 4 with its inputs taken LITERALLY: lambda = 1, x0 = [0, 0, 0, 0, 4, 0, 0], u in [-1, 1]^2 (the
    other configuration of this file starts ON the generated track at (-12, 5), lambda = 20, and keeps the reference's
    throttle range [-0.99, 0.65]).  On the generated standard map the world origin lies 5 m beside the track's centre line
    (map value |15 - 10| + 13/30 = 5.4 against a crash threshold of 0.65), so every rollout starts off the track."""
    cfg = autorally_cfg(K=K, T=T, lambda_=1.0)
    cfg["x0"] = np.array([0.0, 0.0, 0.0, 0.0, 4.0, 0.0, 0.0], np.float32)
    cfg["ranges"] = [[-1.0, 1.0], [-1.0, 1.0]]
    return cfg


@pytest.mark.parametrize("variant", [0, 1], ids=["mfma-pipeline", "mfma-fused"])
@pytest.mark.parametrize("literal", [False, True], ids=["on-track", "survey-literal"])
def test_autorally_16384x150_vs_oracle(gpu, variant, literal):
    """NeuralNetModel<7,2,3> on the MFMA forward + ARStandardCost, SURVEY.md §8d config 4 — on the track (lambda = 20,
    x0 = (-12, 5)) and with the survey's literal inputs (lambda = 1, x0 at the world origin, u in [-1, 1]^2)"""
    cfg = autorally_cfg_survey_literal() if literal else autorally_cfg(K=16384, T=150)
    eng, orc = make_engine(cfg, kernel_variant=variant), make_oracle(cfg)
    eps = host_noise(1, cfg["K"], cfg["T"], 2)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    _check_vanilla(eng, orc)
    costs = orc.costs()
    if literal:
        # every rollout starts off the track and pays the crash cost from the first step on; the ranking that is left comes
        # from the speed / slip terms — the parity bar is the same, the test says what the configuration exercises
        print("\nsurvey-literal config 4: cost min %.1f median %.1f max %.1f, eta %.3f of K = %d" %
              (costs.min(), np.median(costs), costs.max(), orc.stats()["normalizer"][0], cfg["K"]))
        assert np.isfinite(costs).all()
    else:
        assert (costs < 1e4).sum() > 1000, "config should keep a good share of rollouts on the track"


def test_autorally_16384x150_philox_vs_oracle(gpu):
    cfg = autorally_cfg(K=16384, T=150)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    eng.setSeed(7)
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, po.philox_normal(7, 0, cfg["K"], cfg["T"], 2)[None])
    _check_vanilla(eng, orc)


# ------------------------------------------------------------------ config 3: DI Tube K=8192 T=150 --------------------
@pytest.mark.parametrize("kw", [{}, {"kernel_variant": 1}, {"block_x": 64, "block_y": 1}],
                         ids=["folded-pipeline", "fused", "pipeline-64x1x2"])
def test_di_tube_8192x150_vs_oracle(gpu, kw):
    """Tube-MPPI, CORL2020 parameters (examples/double_integrator_CORL2020.cu:30-39, 316-352) at BASELINE size: two
    systems per launch; three calls with the actual state drifting so the nominal-state logic is exercised"""
    cfg = di_cfg(K=8192, T=150)
    eng, orc = make_engine(cfg, **kw), make_oracle(cfg)
    x = cfg["x0"].copy()
    for i in range(3):
        eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=70 + i)
        eng.injectNoise(eps)
        eng.computeControl(x, 1)
        orc.tube_compute_control(x, 1, eps)
        st, so = eng.getStats(), orc.stats()
        if i == 0:
            assert ulp_diff(eng.getSampledCostSeq(), orc.costs()).max() == 0
            _check_system(st.real_sys, so, 0)
            _check_system(st.nominal_sys, so, 1)
        else:  # the mean now carries the ~1e-7 difference of the previous u*: costs agree to fp32 accuracy, not bitwise
            np.testing.assert_allclose(eng.getSampledCostSeq(), orc.costs(), rtol=1e-5)
            assert abs(st.real_sys.baseline - so["baseline"][0]) <= 1e-5 * abs(so["baseline"][0])
            assert abs(st.nominal_sys.baseline - so["baseline"][1]) <= 1e-5 * abs(so["baseline"][1])
        assert st.nominal_state_used == so["nominal_state_used"]
        assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL
        assert np.abs(eng.getNominalControlSeq() - orc.nominal_control()).max() <= U_TOL
        assert np.abs(eng.getTargetStateSeq() - orc.state_traj()).max() <= X_TOL
        assert np.abs(eng.getNominalStateSeq() - orc.nominal_state_traj()).max() <= X_TOL
        x = x + np.array([0.05, -0.03, 0.2, -0.1], np.float32) * (i + 1)  # the actual state drifts off the nominal


# ------------------------------------------------------------------ config 5: LSTM + colored K=65536 T=200 ------------
@pytest.mark.parametrize("noise", ["injected", "philox"])
def test_lstm_colored_65536x200_vs_oracle(gpu, noise):
    """LSTM bicycle-slip dynamics on MFMA + colored-noise sampler (in-kernel GEMM), one iteration at BASELINE size.
    The oracle needs ~30 s on 8 cores for the 13 M LSTM steps + the O(T^2) inverse DFT of every sample row."""
    cfg = bicycle_lstm_cfg(K=65536, T=200)
    cfg["colored"] = ([1.0, 1.0], 0.97, 0.0)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    if noise == "injected":
        z = host_spectrum(1, cfg["K"], cfg["T"], 2, seed=5)
        eng.injectNoise(z)
    else:
        eng.setSeed(99)
        z = po.philox_spectrum(99, 0, cfg["K"], cfg["T"], 2)[None]
    eng.computeControl(cfg["x0"], 1)
    orc.colored_compute_control(cfg["x0"], 1, z, *cfg["colored"])
    _check_vanilla(eng, orc)


# ------------------------------------------------------------------ 8-way K split on one device -----------------------
def _eight_way(cfg, eps, controls=2):
    """rank r of world 8 owns rollouts [r K/8, (r+1) K/8); the all-gather is done by hand with device copies"""
    W = 8
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    ranks = [make_engine(cfg, rank=r, world_size=W) for r in range(W)]
    Kl = cfg["K"] // W
    for r, c in enumerate(ranks):
        assert c.num_rollouts_local == Kl
        if eps is not None:
            c.injectNoise(eps[:, r * Kl:(r + 1) * Kl])
        c.uploadState(cfg["x0"])
        c.iterationLocal()
        c.synchronize()
    bufs = [c.exchangeBuffers() for c in ranks]
    n = bufs[0][2]
    for dst in range(W):
        for src in range(W):
            assert hip.hipMemcpy(bufs[dst][1] + 4 * n * src, bufs[src][0], 4 * n, 3) == 0  # 3 = device to device
    out = []
    for c in ranks:
        c.iterationMerge()
        c.synchronize()
        out.append((c.getOptimalControlSeq()[0], c.getStats().real_sys))
    return out


@pytest.mark.parametrize("mk,K,T,Cd", [(cartpole_cfg, 16384, 100, 1), (autorally_cfg, 16384, 150, 2)],
                         ids=["cartpole", "autorally"])
@pytest.mark.parametrize("noise", ["injected", "philox"])
def test_eight_way_split_matches_unsharded_oracle(gpu, mk, K, T, Cd, noise):
    """SURVEY.md §8e at BASELINE size: the K rollouts split over 8 handles (global rollout index decides the special
    trajectories and the Philox counters) + one record exchange == the oracle's UN-sharded iteration"""
    cfg = mk(K=K, T=T, soft=True) if mk is cartpole_cfg else mk(K=K, T=T)
    orc = make_oracle(cfg)
    if noise == "injected":
        eps = host_noise(1, K, T, Cd, seed=17)
        res = _eight_way(cfg, eps)
    else:
        eps = po.philox_normal(42, 0, K, T, Cd)[None]
        res = _eight_way(cfg, None)
    u_orc = orc.iterate(cfg["x0"], np.zeros((T, Cd), np.float32), eps[0])[0]
    w = orc.weights()[0].astype(np.float64)
    rho = float(orc.costs()[0].min())
    for u, st in res:
        assert np.abs(u - u_orc).max() <= U_TOL, np.abs(u - u_orc).max()
        assert st.baseline == rho
        assert abs(st.normalizer - w.sum()) <= 2e-6 * w.sum()
    # every rank ends with the same bits
    for u, _ in res[1:]:
        assert np.array_equal(u, res[0][0])


def test_full_size_weighted_mean_of_dumped_samples(gpu):
    """u* of the engine == float64 weighted mean of the samples the engine itself dumped (independent of the oracle):
    the block-local softmin + 256-record merge against a direct evaluation of sum_k w_k v_k / sum_k w_k"""
    cfg = cartpole_cfg(K=16384, T=100, soft=True)
    eng = make_engine(cfg, save_samples=True)
    eng.uploadState(cfg["x0"])
    eng.optimize(1)
    costs = eng.getSampledCostSeq()[0].astype(np.float64)
    v = eng.getSampledControls()[0].astype(np.float64)
    w = np.exp(-(costs - costs.min()) / cfg["lambda_"])
    u_direct = (w[:, None, None] * v).sum(0) / w.sum()
    assert np.abs(eng.getOptimalControlSeq()[0] - u_direct).max() <= 2e-6


# ------------------------------------------------------------------ §8(f)-4: elevation-map RACER models K=16384 T=100 --
@pytest.mark.parametrize("model,block_y,variant", [("elevation", 4, 0), ("elevation", 1, 0), ("lstm_steering", 4, 0),
                                                   ("lstm_steering", 4, 1), ("lstm_steering", 1, 1), ("suspension", 4, 0),
                                                   ("suspension", 4, 1), ("suspension", 1, 1), ("uncertainty", 4, 0),
                                                   ("uncertainty", 4, 2), ("uncertainty", 1, 1)],
                         ids=["elev-4lanes-pipeline", "elev-1lane-pipeline", "lstm-4lanes-pipeline", "lstm-4lanes-fused",
                              "lstm-1lane-fused", "suspension-4lanes-pipeline", "suspension-4lanes-fused",
                              "suspension-1lane-fused", "complete-4lanes-fused", "complete-4lanes-pipeline",
                              "complete-1lane-fused"])
def test_racer_elevation_16384x100_vs_oracle(gpu, model, block_y, variant):
    """The elevation-map RACER models (plain, LSTM steering, suspension, the complete model with the mean / uncertainty
    networks) over the synthetic hills at the size DESIGN.md §5 quotes (one block per CU): four lanes per rollout and one
    lane per rollout against the oracle, injected noise"""
    from test_racer_dubins_elevation import elevation_cfg
    from test_racer_dubins_lstm_steering import steering_cfg
    from test_racer_dubins_lstm_unc import uncertainty_cfg
    from test_racer_dubins_suspension import suspension_cfg
    cfg = {"elevation": elevation_cfg, "lstm_steering": steering_cfg, "suspension": suspension_cfg,
           "uncertainty": uncertainty_cfg}[model](K=16384, T=100)
    eng, orc = make_engine(cfg, block_x=64, block_y=block_y, kernel_variant=variant), make_oracle(cfg)
    eps = host_noise(1, cfg["K"], cfg["T"], 2)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    _check_vanilla(eng, orc)


def test_suspension_two_systems_64x4x2_16384x100_vs_oracle(gpu):
    """Tube-MPPI on the suspension model with the (64, 4, 2) block — 512 threads, the instantiation that returned NaN costs for
    injected noise in round 2 while the steering weights sat in per-lane registers (csrc/models/
    racer_dubins_elevation_suspension.hip) — at the full size, injected noise, against the oracle"""
    from test_racer_dubins_suspension import suspension_cfg
    cfg = suspension_cfg(K=16384, T=100, D=2)
    eng, orc = make_engine(cfg, block_x=64, block_y=4), make_oracle(cfg)
    eps = host_noise(1, cfg["K"], cfg["T"], 2, seed=11)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    orc.tube_compute_control(cfg["x0"], 1, eps)
    costs = eng.getSampledCostSeq()
    assert np.isfinite(costs).all()
    assert ulp_diff(costs, orc.costs()).max() == 0
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL
    assert np.abs(eng.getNominalControlSeq() - orc.nominal_control()).max() <= U_TOL
    eng.close()



# ------------------------------------------------------------------ Robust MPPI at the sizes bench.py times it ---------------
def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _robust_pair(cfg, thr, reference_order, nc=9, ns=32, **kw):
    eng = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], cfg["alpha"], cfg["num_iters"],
                                 seed=42, **kw)
    if cfg["dyn"] is not None:
        eng.setDynamicsParams(cfg["dyn"])
    eng.setCostParams(cfg["cost"])
    for name, blob in cfg.get("blobs", {}).items():
        eng.setModelBlob(name, blob)
    if cfg["ranges"] is not None:
        eng.setControlRanges(cfg["ranges"])
    eng.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"])
    eng.setRMPPIParams(thr, nc, ns)
    if reference_order:
        eng.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
    orc = make_oracle(cfg)
    return eng, orc, po.RobustOracle(orc, thr, nc, ns)


def _robust_two_cycles(cfg, thr, reference_order, x_real, require_finite_candidates=False, **kw):
    """Two control cycles of RobustMPPIController on injected noise, engine and oracle side by side (reference:
    controllers/R-MPPI/robust_mppi_controller.cu:635-755 computeControl, :508-633 updateImportanceSamplingControl;
    core/rmppi_kernels.cu:231-356 initEvalKernel, :666-866 rolloutRMPPIKernel).
      cycle 0   the nominal state is not set yet: no candidates; setFeedbackGains; computeControl from identical inputs —
                the TRAJECTORY COSTS OF BOTH SYSTEMS are the oracle's bits, baselines exact, both u* <= 1e-5
      cycle 1   real state moved off the nominal one: 9 x 32 candidate rollouts of time-shifted samples (init-eval kernel),
                best index, nominal state, then computeControl again.
    reference_order: the last stage in the reference's own summation order — then cycle 0's u* is the oracle's bits and
    EVERYTHING of cycle 1 is too (candidate free energies, nominal state, both control sequences, statistics).  Default
    reduction: cycle 1 starts from a u* that differs at the 1e-7 level, so its quantities are held to fp32 accuracy and the
    discrete choice (best candidate) to equality."""
    eng, orc, rob = _robust_pair(cfg, thr, reference_order, **kw)
    S, C, T, K = eng.STATE_DIM, eng.CONTROL_DIM, cfg["T"], cfg["K"]
    g = np.random.default_rng(5).uniform(-0.3, 0.3, (T, S, C)).astype(np.float32)  # bench.py's gains
    x = cfg["x0"].copy()
    for cycle in range(2):
        eps = host_noise(2, K, T, C, seed=900 + cycle)
        eng.injectNoise(eps[1:] if cycle == 0 else eps)
        eng.updateImportanceSamplingControl(x, 1)
        rob.update_importance_sampling(x, 1, eps[0])
        ns_g, best_g, stride_g, fe_g = eng.getRMPPIState()
        ns_o, best_o, stride_o, fe_o = rob.state()
        assert best_g == best_o and stride_g == stride_o, cycle
        if cycle == 1:
            # (a candidate whose 32 costs all lie far above the best one's has exp(-(c - rho)/lambda) = 0 throughout: its free
            # energy is +inf on both sides — the reference computes the same, rmppi_kernels.cu / robust_mppi_controller.cu:
            # 590-612 — and "the same" includes those)
            assert np.array_equal(np.isfinite(fe_g), np.isfinite(fe_o)) and np.isfinite(fe_g).any(), (fe_g, fe_o)
            if require_finite_candidates:
                assert np.isfinite(fe_g).all(), fe_g
            if reference_order:
                assert np.array_equal(_bits(fe_g), _bits(fe_o)), (fe_g, fe_o)
                assert np.array_equal(_bits(ns_g), _bits(ns_o))
            else:
                np.testing.assert_allclose(fe_g, fe_o, rtol=1e-5)
                np.testing.assert_allclose(ns_g, ns_o, rtol=1e-5, atol=1e-6)
        eng.setFeedbackGains(g)
        rob.set_gains(g)
        eng.computeControl(x, 1)
        rob.compute_control(x, 1, eps[1:])
        costs_g, costs_o = eng.getSampledCostSeq(), orc.costs()
        assert costs_g.shape == costs_o.shape == (2, K)
        assert np.isfinite(costs_g).all()
        st, so = eng.getStats(), orc.stats()
        if cycle == 0 or reference_order:
            dc = int(ulp_diff(costs_g, costs_o).max())
            assert dc == 0, "cycle %d: trajectory costs of the two systems differ by up to %d ulp" % (cycle, dc)
            assert st.nominal_sys.baseline == so["baseline"][0] and st.real_sys.baseline == so["baseline"][1]
            for got, z in ((st.nominal_sys.normalizer, 0), (st.real_sys.normalizer, 1)):
                eta = float(so["normalizer"][z])
                assert abs(got - eta) <= (0.0 if reference_order else ETA_RTOL * eta), (cycle, z, got, eta)
        else:
            np.testing.assert_allclose(costs_g, costs_o, rtol=1e-5)
        u_g, un_g = eng.getControlSeq(), eng.getNominalControlSeq()
        du, dun = float(np.abs(u_g - orc.control()).max()), float(np.abs(un_g - orc.nominal_control()).max())
        assert du <= U_TOL and dun <= U_TOL, (cycle, du, dun)
        assert np.abs(eng.getTargetStateSeq() - orc.nominal_state_traj()).max() <= X_TOL
        if reference_order:
            assert np.array_equal(_bits(u_g), _bits(orc.control())), cycle
            assert np.array_equal(_bits(un_g), _bits(orc.nominal_control())), cycle
            assert np.array_equal(_bits(eng.getTargetStateSeq()), _bits(orc.nominal_state_traj())), cycle
        x = x_real.copy()
    eng.close()


@pytest.mark.parametrize("lam", [1.0, 20.0], ids=["lambda1-bench", "lambda20"])
@pytest.mark.parametrize("reference_order", [False, True], ids=["default-reduction", "reference-order"])
def test_robust_autorally_16384x150_vs_oracle(gpu, reference_order, lam):
    """The kernel bench.py's `robust_autorally_nn` leg times — rolloutRMPPIPipelineKernel<NeuralNetModelMFMA<7,2,3>, ARStandardCost,
    DeviceDDP, Gaussian>, 960-thread blocks, chain-masked MFMA rows, 256 blocks = one per CU — AT THE SIZE IT IS TIMED
    (K = 16384, T = 150, threshold 500, the bench's gains), held to oracle_rmppi.hpp.  Round 5 checked this instantiation up to
    K = 4096 only.  lambda = 1 is the bench's value (sharp weights: all but the best candidate's free energies overflow to +inf,
    on both sides); lambda = 20 spreads the weights, every candidate's free energy is finite and compared."""
    cfg = autorally_cfg(K=16384, T=150, lambda_=lam)
    cfg["D"] = 2
    cfg["control_cost_coeff"] = [0.2, 0.1]
    x_real = cfg["x0"] + np.array([0.15, -0.1, 0.05, 0.02, 0.1, 0.02, 0.0], np.float32)
    _robust_two_cycles(cfg, 500.0, reference_order, x_real, require_finite_candidates=lam >= 20.0)


@pytest.mark.parametrize("variant", ["pipeline", "fused"])
def test_robust_complete_racer_4096x100_vs_oracle(gpu, variant):
    """Robust MPPI on the COMPLETE RACER model (RacerDubinsElevationLSTMUncertaintyQuad: steering LSTM + mean and uncertainty
    networks, covariance propagation, elevation / normals maps): the instantiation with the most spilled registers of the whole
    library (round 5's code object: 217-222 spilled VGPRs, 405-438 spilled SGPRs, 672-688 B of scratch per lane), at K = 4096,
    T = 100 — 64 blocks of 64 rollouts x 2 systems, past the K ~ 1000 the round-5 tests stopped at.  Both kernel structures."""
    from test_racer_dubins_lstm_unc import uncertainty_cfg
    cfg = uncertainty_cfg(K=4096, T=100, D=2)
    cfg["control_cost_coeff"] = [0.2, 0.1]
    dx = np.zeros_like(cfg["x0"])
    dx[:7] = [0.3, -0.2, 0.1, 0.05, 0.02, 0.01, 0.0]
    kv = m.MPPI_KERNEL_PIPELINE if variant == "pipeline" else m.MPPI_KERNEL_FUSED
    _robust_two_cycles(cfg, 2000.0, True, cfg["x0"] + dx, kernel_variant=kv)
    _robust_two_cycles(cfg, 2000.0, False, cfg["x0"] + dx, kernel_variant=kv)
