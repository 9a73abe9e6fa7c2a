"""Robust MPPI: DDP feedback k(), init-eval kernel, rolloutRMPPIKernel, controller logic."""
import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import (autorally_cfg, bicycle_lstm_cfg, cartpole_cfg_lr, di_cfg, host_noise, make_engine, make_oracle, racer_cfg,
                    ulp_diff)

U_TOL = 1e-5


def _rm_cfg(model="di", K=1024, T=40, num_iters=1):
    if model == "di":
        cfg = di_cfg(K=K, T=T, tube=True, num_iters=num_iters)
        cfg["control_cost_coeff"] = [0.3, 0.2]  # exercise the likelihood-ratio and feedback cost terms
        cfg["ranges"] = [[-3.0, 3.0], [-3.0, 3.0]]
    elif model == "racer":
        cfg = racer_cfg(K=K, T=T, num_iters=num_iters)
        cfg["D"] = 2
        cfg["control_cost_coeff"] = [0.2, 0.1]
    elif model in ("elevation", "lstm_steering", "suspension", "complete"):
        from test_racer_dubins_elevation import elevation_cfg
        from test_racer_dubins_lstm_steering import steering_cfg
        from test_racer_dubins_lstm_unc import uncertainty_cfg
        from test_racer_dubins_suspension import suspension_cfg
        mk = {"elevation": elevation_cfg, "lstm_steering": steering_cfg, "suspension": suspension_cfg, "complete": uncertainty_cfg}
        cfg = mk[model](K=K, T=T, D=2)
        cfg["num_iters"] = num_iters
        cfg["control_cost_coeff"] = [0.2, 0.1]
    elif model in ("autorally", "lstm"):
        # the NN models: Robust MPPI runs them one lane per rollout and system (LDS forward)
        cfg = autorally_cfg(K=K, T=T, num_iters=num_iters) if model == "autorally" else bicycle_lstm_cfg(K=K, T=T, num_iters=num_iters)
        cfg["D"] = 2
        cfg["control_cost_coeff"] = [0.2, 0.1]
    else:
        cfg = cartpole_cfg_lr(K=K, T=T)
        cfg["D"] = 2
        cfg["num_iters"] = num_iters
        cfg["std_dev"] = [5.0, 4.0]  # different exploration for the nominal and the real system
    return cfg


def _gains(T, S, C, seed=1, scale=0.4):
    return np.random.default_rng(seed).uniform(-scale, scale, (T, S, C)).astype(np.float32)


def _make_pair(cfg, thr=1000.0, nc=9, ns=32, **kw):
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
    orc = make_oracle(cfg)
    rob = po.RobustOracle(orc, thr, nc, ns)
    return eng, orc, rob


# ------------------------------------------------------------------ CPU: oracle pinned on the reference's KATs --------
def test_line_search_weights_strides_candidates_known_answers():
    """reference: tests/controllers/rmppi_test.cu:226-292 (LineSearchWeights_9, ImportanceSampler_Stride_2 / _4,
    InitEvalSelection_Weights)"""
    rob = po.RobustOracle(make_oracle(_rm_cfg(K=576, T=10)), 1000.0, 9, 64)
    w, s2 = rob.line_search(2)
    known = np.array([[1, .75, .5, .25, 0, 0, 0, 0, 0], [0, .25, .5, .75, 1, .75, .5, .25, 0],
                      [0, 0, 0, 0, 0, .25, .5, .75, 1]], np.float32)
    assert np.array_equal(w, known)
    assert list(s2) == [0, 1, 1, 2, 2, 2, 2, 2, 2]
    _, s4 = rob.line_search(4)
    assert list(s4) == [0, 1, 2, 3, 4, 4, 4, 4, 4]
    cand = rob.candidates([-4, 0, 0, 0], [0, 4, 0, 0], [4, 4, 0, 0])
    want = np.zeros((9, 4), np.float32)
    want[:, 0] = [-4, -3, -2, -1, 0, 1, 2, 3, 4]
    want[:, 1] = [0, 1, 2, 3, 4, 4, 4, 4, 4]
    assert np.array_equal(cand, want)


def test_best_candidate_selection():
    """reference: tests/controllers/rmppi_test.cu:357-421 (GetCandidateBaseline, ComputeBestCandidate): the LAST
    candidate whose free energy is below the threshold"""
    cfg = _rm_cfg(K=576, T=10)
    lam = cfg["lambda_"]
    rng = np.random.default_rng(3)
    for thr in (1000.0, 30.0, 12.0):
        rob = po.RobustOracle(make_oracle(cfg), thr, 9, 64)
        costs = (rng.uniform(5, 60, (9, 64)) + np.arange(9)[:, None] * 4).astype(np.float32)
        best, fe = rob.best_index(costs)
        base = costs.min()
        want_fe = -lam * np.log(np.exp(-(costs.astype(np.float64) - base) / lam).mean(1)) + base
        np.testing.assert_allclose(fe, want_fe, rtol=2e-6)
        below = np.nonzero(want_fe < thr)[0]
        assert best == (below[-1] if below.size else 0)


def test_ddp_feedback_reference_behaviour_and_sum_mode():
    """reference: feedback_controllers/DDP/ddp.cu:11-45 — even CONTROL_DIM: only the last state's gain row survives;
    odd CONTROL_DIM and accumulate_all_states: the full K (x - x*)"""
    for model, S, C in (("di", 4, 2), ("cartpole", 4, 1)):
        cfg = _rm_cfg(model, K=64, T=8)
        rob = po.RobustOracle(make_oracle(cfg))
        g = _gains(8, S, C)
        x, xs = np.array([1, -2, 0.5, 3], np.float32), np.array([0.5, 1, -1, 2], np.float32)
        e = x - xs
        full = (g[3] * e[:, None]).sum(0)
        rob.set_gains(g, accumulate_all_states=True)
        np.testing.assert_allclose(rob.feedback(x, xs, 3), full, rtol=1e-6, atol=1e-7)
        rob.set_gains(g, accumulate_all_states=False)
        want = g[3, S - 1] * e[S - 1] if C % 2 == 0 else full
        np.testing.assert_allclose(rob.feedback(x, xs, 3), want, rtol=1e-6, atol=1e-7)


def test_rmppi_rollout_zero_gains_identical_systems():
    """with no feedback, identical initial states and identical exploration the real and nominal costs coincide
    (the nominal cost formula collapses to A + LR) — reference invariant of tests/mppi_core/rmppi_kernel_tests.cu"""
    cfg = _rm_cfg("di", K=256, T=30)
    orc = make_oracle(cfg)
    rob = po.RobustOracle(orc)
    rob.set_gains(np.zeros((30, 4, 2), np.float32))
    eps = host_noise(1, 256, 30, 2)[0]
    mean = np.tile((0.1 * np.ones((30, 2), np.float32))[None], (2, 1, 1))
    v = orc.set_gaussian_controls(mean, eps, 1, 0)
    costs, _ = rob.rollout_costs(np.tile(cfg["x0"], (2, 1)), mean, v)
    np.testing.assert_allclose(costs[0], costs[1], rtol=2e-6)


# ------------------------------------------------------------------ GPU parity -----------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("model,acc_all,mode", [("di", False, "injected"), ("di", True, "injected"), ("di", False, "philox"),
                                                ("racer", False, "injected"), ("racer", True, "philox"),
                                                ("cartpole", False, "injected"), ("cartpole", False, "philox"),
                                                ("autorally", False, "injected"), ("autorally", True, "philox"),
                                                ("lstm", False, "injected"),
                                                ("elevation", False, "injected"), ("elevation", True, "philox"),
                                                ("lstm_steering", False, "injected"), ("lstm_steering", False, "philox"),
                                                ("suspension", False, "injected"), ("suspension", True, "philox"),
                                                ("complete", False, "injected"), ("complete", True, "philox")])
def test_rmppi_rollout_costs_bit_exact(gpu, model, acc_all, mode):
    _rollout_costs_bit_exact(model, acc_all, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("model,K,T,mode,variant", [("autorally", 1000, 37, "injected", m.MPPI_KERNEL_FUSED),
                                                     ("autorally", 1000, 37, "philox", m.MPPI_KERNEL_FUSED),
                                                     ("suspension", 1000, 37, "philox", m.MPPI_KERNEL_FUSED),
                                                     ("di", 1000, 37, "injected", m.MPPI_KERNEL_FUSED),
                                                     ("cartpole", 1000, 37, "philox", m.MPPI_KERNEL_FUSED),
                                                     ("di", 333, 150, "philox", m.MPPI_KERNEL_PIPELINE),
                                                     ("cartpole", 300, 3, "injected", m.MPPI_KERNEL_PIPELINE),
                                                     ("autorally", 330, 150, "philox", m.MPPI_KERNEL_PIPELINE),
                                                     ("autorally", 320, 3, "injected", m.MPPI_KERNEL_PIPELINE),
                                                     ("autorally", 300, 1, "philox", m.MPPI_KERNEL_PIPELINE),
                                                     ("lstm", 1000, 37, "philox", m.MPPI_KERNEL_PIPELINE)])
def test_rmppi_rollout_costs_bit_exact_both_kernels(gpu, model, K, T, mode, variant):
    """models with replicated-lane dynamics run the role-pipelined kernel (rmppi_pipeline_kernel.hpp) by default; the fused
    rolloutRMPPIKernel stays available (kernel_variant) and both give the oracle's bits — also at the benchmark horizon, at
    horizons shorter than a sampler trip and the rings, and with blocks of one partly filled wave"""
    _rollout_costs_bit_exact(model, False, mode, K=K, T=T, kernel_variant=variant)


@pytest.mark.gpu
def test_rmppi_pipelined_kernel_independent_noise(gpu):
    """use_same_noise_for_all_distributions off: the sampler waves draw one Philox stream per system; fused == pipelined"""
    cfg = _rm_cfg("autorally", K=512, T=20)
    got = []
    for variant in (m.MPPI_KERNEL_FUSED, m.MPPI_KERNEL_PIPELINE):
        eng, orc, rob = _make_pair(cfg, thr=40.0, save_samples=True, kernel_variant=variant)
        g = _gains(cfg["T"], eng.STATE_DIM, eng.CONTROL_DIM)
        eng.setFeedbackGains(g, False)
        eng.setIndependentNoise(True)
        x0 = np.stack([cfg["x0"], cfg["x0"] + np.float32(0.05)])
        got.append((eng.rolloutCosts(x0, 2).copy(), eng.getSampledControls().copy()))
        eng.close()
    assert np.isfinite(got[0][0]).all()
    assert ulp_diff(got[0][0], got[1][0]).max() == 0
    assert ulp_diff(got[0][1], got[1][1]).max() == 0
    assert np.abs(got[0][1][0] - got[0][1][1]).max() > 1e-3  # the two systems really drew different noise


@pytest.mark.gpu
@pytest.mark.parametrize("model,T", [("autorally", 37), ("autorally", 150), ("lstm", 20), ("suspension", 21), ("complete", 12),
                                     ("di", 150), ("cartpole", 41)])
def test_rmppi_candidate_evaluation_both_kernels(gpu, model, T):
    """updateImportanceSamplingControl's candidate rollouts (9 x 32, time-shifted samples) on the fused init-eval kernel and
    as blocks of role waves: the same candidate free energies, best index and nominal state, bit for bit"""
    cfg = _rm_cfg(model, K=1024, T=T)
    got = []
    for variant in (m.MPPI_KERNEL_FUSED, m.MPPI_KERNEL_PIPELINE):
        eng, orc, rob = _make_pair(cfg, thr={"autorally": 500.0}.get(model, 2000.0), kernel_variant=variant)
        S, C = eng.STATE_DIM, eng.CONTROL_DIM
        x = cfg["x0"].copy()
        rec = []
        for i in range(3):
            eng.updateImportanceSamplingControl(x, 1 + i)  # strides 1, 2, 3: odd and even shifts of the sample rows
            ns_, best, stride, fe = eng.getRMPPIState()
            rec.append((ns_.copy(), np.array([best, stride]), fe.copy()))
            eng.setFeedbackGains(_gains(T, S, C, seed=3 + i, scale=0.3))
            eng.computeControl(x, 1 + i)
            x = x + np.float32(0.02)
        got.append(rec)
        eng.close()
    for a, b in zip(*got):
        for p, q in zip(a, b):
            assert np.array_equal(p, q)
    assert np.isfinite(got[0][-1][2]).all() and np.abs(got[0][-1][2]).max() > 0


def _rollout_costs_bit_exact(model, acc_all, mode, K=1000, T=37, **kw):
    cfg = _rm_cfg(model, K=K, T=T)  # default: ragged last block, odd horizon
    eng, orc, rob = _make_pair(cfg, thr=40.0, save_samples=True, **kw)
    S, C, T, K = eng.STATE_DIM, eng.CONTROL_DIM, cfg["T"], cfg["K"]
    g = _gains(T, S, C)
    eng.setFeedbackGains(g, acc_all)
    rob.set_gains(g, acc_all)
    mean = (0.3 * np.sin(np.arange(T * C, dtype=np.float32) * 0.2)).reshape(T, C)
    eng.updateImportanceSampler(mean)
    if mode == "injected":
        eps = host_noise(1, K, T, C)[0]
        eng.injectNoise(eps)
    else:
        eps = po.philox_normal(42, 0, K, T, C)
    dx = np.zeros(S, np.float32)
    dx[:min(S, 7)] = np.array([0.3, -0.2, 0.1, 0.05, 0.02, 0.01, 0.0], np.float32)[:S]
    x0 = np.stack([cfg["x0"], cfg["x0"] + dx])
    got = eng.rolloutCosts(x0, 2)
    means = np.tile(mean, (2, 1, 1))
    v = orc.set_gaussian_controls(means, eps, 2, 0)
    want, v_fb = rob.rollout_costs(x0, means, v)
    assert np.isfinite(got).all()
    assert ulp_diff(got, want).max() == 0
    assert ulp_diff(eng.getSampledControls(), v_fb).max() == 0  # feedback-filled clamped controls written back
    assert np.abs(want[0] - want[1]).max() > 1e-3  # the two systems really differ


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["di", "cartpole", "autorally"])  # autorally: the role-pipelined kernel
def test_rmppi_time_specific_std_dev(gpu, model):
    """time_specific_std_dev on a Robust handle: setGaussianControls, the likelihood-ratio cost AND the feedback cost read
    sigma[d][t][c] (gaussian.cu:21-43, :488-493, :579-583) — the feedback cost once kept the scalar sigma (round-2 advice)"""
    cfg = _rm_cfg(model, K=500, T=29)
    eng, orc, rob = _make_pair(cfg, thr=40.0, save_samples=True)
    S, C, T, K = eng.STATE_DIM, eng.CONTROL_DIM, cfg["T"], cfg["K"]
    sd = (0.4 + 1.2 * np.random.default_rng(5).random((2, T, C))).astype(np.float32)
    eng.setTimeSpecificStdDev(sd)
    orc.set_time_specific_std_dev(sd)
    g = _gains(T, S, C)
    eng.setFeedbackGains(g)
    rob.set_gains(g)
    mean = (0.3 * np.sin(np.arange(T * C, dtype=np.float32) * 0.2)).reshape(T, C)
    eng.updateImportanceSampler(mean)
    eps = host_noise(1, K, T, C)[0]
    eng.injectNoise(eps)
    dx = np.zeros(S, np.float32)
    dx[:4] = [0.3, -0.2, 0.1, 0.05]
    x0 = np.stack([cfg["x0"], cfg["x0"] + dx])
    got = eng.rolloutCosts(x0, 1)
    means = np.tile(mean, (2, 1, 1))
    v = orc.set_gaussian_controls(means, eps, 1, 0)
    want, v_fb = rob.rollout_costs(x0, means, v)
    assert ulp_diff(got, want).max() == 0
    assert ulp_diff(eng.getSampledControls(), v_fb).max() == 0
    # the table matters for the feedback term: with the scalar sigma the real system's costs are different numbers
    orc.set_time_specific_std_dev(None)
    v2 = orc.set_gaussian_controls(means, eps, 1, 0)
    other, _ = rob.rollout_costs(x0, means, v2)
    assert ulp_diff(got, other).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("model,T,block_x", [("autorally", 150, 0), ("di", 37, 32), ("di", 200, 0)])
def test_rmppi_32_rollout_blocks(gpu, model, T, block_x):
    """horizons whose sample rows for 64 rollouts x 2 systems overflow the 160 KiB of LDS run with (32, 1, 2) blocks
    (chosen automatically, or requested): same costs, bit for bit"""
    cfg = _rm_cfg(model, K=320, T=T)
    eng, orc, rob = _make_pair(cfg, thr=40.0, save_samples=True, block_x=block_x)
    S, C, K = eng.STATE_DIM, eng.CONTROL_DIM, cfg["K"]
    g = _gains(T, S, C)
    eng.setFeedbackGains(g)
    rob.set_gains(g)
    mean = (0.3 * np.sin(np.arange(T * C, dtype=np.float32) * 0.2)).reshape(T, C)
    eng.updateImportanceSampler(mean)
    eps = host_noise(1, K, T, C)[0]
    eng.injectNoise(eps)
    x0 = np.stack([cfg["x0"], cfg["x0"] + np.array([0.3, -0.2, 0.1, 0.05, 0.02, 0.01, 0.0], np.float32)[:S]])
    got = eng.rolloutCosts(x0, 2)
    means = np.tile(mean, (2, 1, 1))
    v = orc.set_gaussian_controls(means, eps, 2, 0)
    want, v_fb = rob.rollout_costs(x0, means, v)
    assert ulp_diff(got, want).max() == 0
    assert ulp_diff(eng.getSampledControls(), v_fb).max() == 0
    # and a whole control computation through the merged records of the 32-rollout blocks
    eng, orc, rob = _make_pair(cfg, thr=40.0, block_x=block_x)
    eps2 = host_noise(2, K, T, C, seed=77)
    eng.injectNoise(eps2[1:])
    eng.updateImportanceSamplingControl(x0[1], 1)
    rob.update_importance_sampling(x0[1], 1, eps2[0])
    eng.setFeedbackGains(g)
    rob.set_gains(g)
    eng.computeControl(x0[1], 1)
    rob.compute_control(x0[1], 1, eps2[1:])
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL
    assert np.abs(eng.getNominalControlSeq() - orc.nominal_control()).max() <= U_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["di", "cartpole", "autorally"])
def test_rmppi_closed_loop_parity(gpu, model):
    """updateImportanceSamplingControl (candidates, init-eval kernel, best index, slide) + computeControl over several
    steps with a disturbed real state, against the oracle"""
    cfg = _rm_cfg(model, K=1024, T=40, num_iters=2)
    nc, ns = 9, 32
    eng, orc, rob = _make_pair(cfg, thr={"di": 25.0, "autorally": 500.0}.get(model, 2000.0), nc=nc, ns=ns)
    S, C, T, K = eng.STATE_DIM, eng.CONTROL_DIM, cfg["T"], cfg["K"]
    x = cfg["x0"].copy()
    used = set()
    for i in range(5):
        g = _gains(T, S, C, seed=10 + i, scale=0.3)
        eps = host_noise(3, K, T, C, seed=50 + i)
        first = i == 0
        eng.injectNoise(eps[1:] if first else eps)  # the first call does not evaluate candidates (nominal not set yet)
        eng.updateImportanceSamplingControl(x, 2)
        rob.update_importance_sampling(x, 2, eps[0])
        ns_g, best_g, stride_g, fe_g = eng.getRMPPIState()
        ns_o, best_o, stride_o, fe_o = rob.state()
        assert best_g == best_o and stride_g == stride_o
        # the candidates are blends of states that went through the optimised controls (equal to U_TOL, not bitwise)
        np.testing.assert_allclose(ns_g, ns_o, rtol=1e-5, atol=1e-6)
        if not first:
            np.testing.assert_allclose(fe_g, fe_o, rtol=1e-5)
            used.add(best_g)
        eng.setFeedbackGains(g)
        rob.set_gains(g)
        eng.computeControl(x, 1)
        rob.compute_control(x, 1, eps[1:])
        assert np.abs(eng.getControlSeq() - orc.control()).max() <= U_TOL
        assert np.abs(eng.getNominalControlSeq() - orc.nominal_control()).max() <= U_TOL
        assert np.abs(eng.getTargetStateSeq() - orc.nominal_state_traj()).max() <= 1e-4
        st, so = eng.getStats(), orc.stats()
        assert abs(st.nominal_sys.baseline - so["baseline"][0]) <= 1e-5 * abs(so["baseline"][0]) + 1e-6
        assert abs(st.real_sys.baseline - so["baseline"][1]) <= 1e-5 * abs(so["baseline"][1]) + 1e-6
        # the real system drifts away from the nominal one
        x, _ = orc.model_step(x, orc.control()[0])
        x = x + np.array([0.05, -0.03, 0.1, -0.05, 0.02, 0.01, 0.0], np.float32)[:S]
    assert len(used) >= 1


@pytest.mark.gpu
def test_rmppi_error_paths(gpu):
    cfg = _rm_cfg("di", K=512, T=20)
    eng, _, _ = _make_pair(cfg)
    with pytest.raises(m.MPPIError) as e:
        eng.computeControl(cfg["x0"], 1)  # no gains yet
    assert e.value.status == 7 and "gains" in str(e.value)
    for bad, msg in ((1, "greater or equal to 3"), (4, "must be odd"), (99, "cannot exceed")):
        with pytest.raises(m.MPPIError) as e:
            eng.setRMPPIParams(1000.0, bad, 32)
        assert e.value.status == 1 and msg in str(e.value)
    with pytest.raises(m.MPPIError) as e:  # the Robust kernels have one shape: (64 rollouts, 1 lane, 2 systems)
        m.RobustMPPIController("autorally_nn", 512, 20, 0.02, 1.0, block_x=64, block_y=4)
    assert e.value.status == 5
    v = m.VanillaMPPIController("cartpole", 128, 10, 0.02, 1.0)
    with pytest.raises(m.MPPIError) as e:
        v._check(v._lib.mppi_set_feedback_gains(v._h, np.zeros(40, np.float32), 0))
    assert e.value.status == 7


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["injected", "philox"])
def test_rmppi_runs_networks_of_other_shapes_on_the_one_lane_form(gpu, mode):
    """Robust MPPI runs the elevation-map RACER models on their four-lanes-per-rollout form, which is compiled for the
    reference's network shapes; a steering network of another shape ("lstm_structure") used to be refused — since round 3 both
    Robust kernels then run on the model itself, one lane per rollout and system (ModelT::withRmppiDynamics): costs of both
    systems and the written-back controls against the oracle, bit for bit"""
    cfg = _rm_cfg("lstm_steering", K=512, T=12)
    Hn = 6
    rng = np.random.default_rng(3)
    blobs = {k: v for k, v in cfg["blobs"].items() if k.startswith("elevation")}
    blobs["lstm_structure"] = np.array([Hn, Hn + 4, 12, 1], np.float32)
    blobs["lstm_weights"] = rng.uniform(-0.4, 0.4, 4 * Hn * Hn + 4 * Hn * 4 + 6 * Hn).astype(np.float32)
    blobs["lstm_output_weights"] = rng.uniform(-0.4, 0.4, (Hn + 4) * 12 + 12 + 12 + 1).astype(np.float32)
    cfg["blobs"] = dict(sorted(blobs.items(), key=lambda kv: 0 if kv[0] == "lstm_structure" else 1))
    eng, orc, rob = _make_pair(cfg, thr=40.0, save_samples=True)
    S, C, T, K = eng.STATE_DIM, eng.CONTROL_DIM, cfg["T"], cfg["K"]
    g = _gains(T, S, C)
    eng.setFeedbackGains(g, False)
    rob.set_gains(g, False)
    mean = (0.3 * np.sin(np.arange(T * C, dtype=np.float32) * 0.2)).reshape(T, C)
    eng.updateImportanceSampler(mean)
    if mode == "injected":
        eps = host_noise(1, K, T, C)[0]
        eng.injectNoise(eps)
    else:
        eps = po.philox_normal(42, 0, K, T, C)
    dx = np.zeros(S, np.float32)
    dx[:7] = [0.3, -0.2, 0.1, 0.05, 0.02, 0.01, 0.0]
    x0 = np.stack([cfg["x0"], cfg["x0"] + dx])
    got = eng.rolloutCosts(x0, 2)
    means = np.tile(mean, (2, 1, 1))
    v = orc.set_gaussian_controls(means, eps, 2, 0)
    want, v_fb = rob.rollout_costs(x0, means, v)
    assert np.isfinite(got).all()
    assert ulp_diff(got, want).max() == 0
    assert ulp_diff(eng.getSampledControls(), v_fb).max() == 0
    # and a whole computeControl (candidate evaluation + rollout + post-processing) runs on that form
    eng2, orc2, rob2 = _make_pair(cfg, thr=2000.0)
    e3 = host_noise(2, K, T, C, seed=9)
    eng2.injectNoise(e3[1:])
    eng2.updateImportanceSamplingControl(cfg["x0"], 1)
    rob2.update_importance_sampling(cfg["x0"], 1, e3[0])
    eng2.setFeedbackGains(g, False)
    rob2.set_gains(g, False)
    eng2.computeControl(cfg["x0"], 1)
    rob2.compute_control(cfg["x0"], 1, e3[1:])
    assert np.abs(eng2.getControlSeq() - orc2.control()).max() <= U_TOL
    eng.close()
    eng2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model,world", [("di", 2), ("autorally", 2), ("di", 4)])
def test_rmppi_ranks_match_unsharded(gpu, model, world):
    """Robust MPPI with the rollouts sharded over 2 / 4 ranks (handles on this GPU, P2P mailbox exchange) AND the candidate
    evaluation sharded by candidate over the same ranks (9 candidates: 5 + 4, or 3 + 3 + 3 + 0; the costs travel through the
    mailbox's aux channel and every rank picks the best candidate from all 9 x 32 of them): the same nominal / real control
    sequences and the same best candidate as one handle, rank for rank the same bits.  AutoRally runs the role-pipelined
    kernels.  The calls block until the merged result is there, so each rank is driven from its own thread — as one process
    per GPU would."""
    import threading
    import gc
    gc.collect()  # every in-process rank needs a hardware queue of its own: no stream of an earlier test may stay alive
    cfg = _rm_cfg(model, K=2048, T=40, num_iters=2)
    thr = {"di": 25.0}.get(model, 500.0)
    S = C = None

    def drive(eng, out, barrier=None):
        x = cfg["x0"].copy()
        for i in range(3):
            eng.updateImportanceSamplingControl(x, 1)
            eng.setFeedbackGains(_gains(cfg["T"], eng.STATE_DIM, eng.CONTROL_DIM, seed=20 + i, scale=0.3))
            if barrier is not None:
                barrier.wait()
            eng.computeControl(x, 1)
            out.append((eng.getControlSeq().copy(), eng.getNominalControlSeq().copy(), eng.getRMPPIState()[1]))
            x = x + np.float32(0.01)

    full, _, _ = _make_pair(cfg, thr=thr)
    ref = []
    drive(full, ref)
    full.close()
    ranks = [_make_pair(cfg, thr=thr, rank=r, world_size=world)[0] for r in range(world)]
    m.MPPIController.p2pConnectLocal(ranks)
    outs = [[] for _ in range(world)]
    # a rank that fails must not leave the others waiting for ever (a blocked non-daemon thread would keep the whole pytest
    # process from exiting): the barrier times out, the threads are daemons
    barrier = threading.Barrier(world, timeout=60)
    ts = [threading.Thread(target=drive, args=(ranks[r], outs[r], barrier), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not any(t.is_alive() for t in ts)
    for c in ranks:
        c.close()
    assert all(len(o) == 3 for o in outs)
    for q in range(1, world):
        for a, b, r in zip(outs[0], outs[q], ref):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] == r[2]
            assert np.abs(a[0] - r[0]).max() <= 5e-6 and np.abs(a[1] - r[1]).max() <= 5e-6
