"""The last stage of an iteration in the REFERENCE'S OWN ARITHMETIC ORDER (mppi_set_reduction_mode,
mppi-generic_amd/csrc/exact_reduce_kernels.hpp) against the oracle — BIT FOR BIT:

  rho   first-occurring minimum                                   core/mppi_common.cu:885-900
  w_k   exp(-(S_k - rho)/lambda) against the GLOBAL rho           core/mppi_common.cu:958-966
  eta   float(sum of double(w_k)) in index order                  core/mppi_common.cu:1055-1063
  F     computeFreeEnergy's serial fp32 sums                      core/mppi_common.cu:1065-1081
  u*    weight = w_k / eta per rollout, cells of sum_strides consecutive rollouts serially, then the cells serially
                                                                  core/mppi_common.cu:1115-1160

(the oracle's weighted reduction is pinned bitwise on the reference's own kernel test, tests/test_oracle_kat.py), and on top
of it FREE-RUNNING closed loops — never re-synchronised — for the Tube and the Robust controller (the Vanilla ones at the BASELINE
sizes are in test_closed_loop_parity.py): BASELINE.md §3's "L-inf(u*) <= 1e-5 after 100 closed-loop iterations".
"""
import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import autorally_cfg, cartpole_cfg, di_cfg, host_noise, make_engine, make_oracle, ulp_diff

pytestmark = pytest.mark.gpu

U_TOL = 1e-5


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("K,lam", [(10000, 0.5), (1000, 0.25), (16384, 20.0), (20000, 1.0), (37, 3.0)])
def test_weights_and_statistics_reference_order(gpu, K, lam):
    """reference: tests/mppi_core/normexp_kernel_tests.cu:126-256; K = 20000 spans three LDS tiles, K = 37 is ragged"""
    rng = np.random.default_rng(K)
    costs = rng.uniform(5, 60, K).astype(np.float32)
    costs[K // 3] = costs.min()  # the minimum twice: the FIRST occurrence is the baseline (same value either way)
    w, st = m.compute_weights_reference_order(costs, lam)
    base = po.baseline(costs)
    w_o = po.norm_exp(costs, np.float32(1.0 / lam), base)
    assert st[0] == base
    assert np.array_equal(_bits(w), _bits(w_o))
    assert st[1] == po.normalizer(w_o)
    fe = po.free_energy(w_o, base, lam)
    assert np.array_equal(_bits(st[2:5]), _bits(fe)), (st[2:5], fe)


def test_weights_reference_order_sum_is_order_sensitive(gpu):
    """weights spanning 30 binades: the double sum in index order differs from other orders in its last bits often enough
    that a wrong order would show over 200 random vectors; the kernel must equal the oracle's serial sum every time"""
    rng = np.random.default_rng(5)
    for trial in range(200):
        costs = (rng.uniform(0, 18, 4096) ** 1.3).astype(np.float32)
        w, st = m.compute_weights_reference_order(costs, 0.8)
        w_o = po.norm_exp(costs, np.float32(1.0 / 0.8), po.baseline(costs))
        assert st[1] == po.normalizer(w_o), trial


@pytest.mark.parametrize("fma", [False, True], ids=["mul_add", "fma"])
@pytest.mark.parametrize("K,T,C,stride", [(1024, 100, 4, 64), (1000, 37, 2, 32), (16384, 150, 2, 32), (70, 5, 1, 32),
                                          (4096, 100, 1, 1)])
def test_weighted_reduction_reference_order(gpu, K, T, C, stride, fma):
    """reference: tests/mppi_core/weightedreduction_kernel_tests.cu:135-173 — here bitwise, ragged last cell included"""
    rng = np.random.default_rng(K + T)
    w = np.exp(-rng.normal(5.0, 1.2, K)).astype(np.float32)
    w[0] = 1.0
    v = rng.normal(5.0, 1.2, (K, T, C)).astype(np.float32)
    eta = po.normalizer(w)
    po.set_reduction_fma(fma)
    try:
        u_o = po.weighted_reduction(w, v, eta, stride)
    finally:
        po.set_reduction_fma(False)
    u_g = m.weighted_reduction_reference_order(w, v, eta, stride, fma)
    assert np.array_equal(_bits(u_g), _bits(u_o))
    if not fma and K >= 1024 and stride > 1:  # the two flavours are different functions (otherwise the parametrisation tests nothing)
        assert not np.array_equal(_bits(u_g), _bits(m.weighted_reduction_reference_order(w, v, eta, stride, True)))


@pytest.mark.parametrize("variant", [m.MPPI_KERNEL_PIPELINE, m.MPPI_KERNEL_FUSED], ids=["pipeline", "fused"])
@pytest.mark.parametrize("soft", [False, True], ids=["lambda0.25", "lambda200"])
def test_iterations_bit_identical(gpu, variant, soft):
    """three optimisation iterations in one computeControl: with the reduction in the reference's order the mean of every
    iteration is the oracle's bit for bit, so costs stay 0 ulp in iterations 2 and 3 as well (the fused reduction: only in
    the first), and the smoothed sequence, the state trajectory and the statistics are the oracle's bits"""
    cfg = cartpole_cfg(K=2048, T=100, num_iters=3, soft=soft)
    eps = host_noise(3, cfg["K"], cfg["T"], 1)
    eng, orc = make_engine(cfg, kernel_variant=variant), make_oracle(cfg)
    eng.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    assert int(ulp_diff(eng.getSampledCostSeq(), orc.costs()).max()) == 0
    st, so = eng.getStats().real_sys, orc.stats()
    assert st.baseline == so["baseline"][0] and st.normalizer == so["normalizer"][0]
    assert st.free_energy_mean == so["free_energy"][0] and st.free_energy_variance == so["free_energy_var"][0]
    assert np.array_equal(_bits(eng.getControlSeq()), _bits(orc.control()))
    assert np.array_equal(_bits(eng.getTargetStateSeq()), _bits(orc.state_traj()))
    eng.close()


def test_mode_switch_and_sum_strides(gpu):
    """the mode can be switched on a live handle; sum_strides (GaussianParams, gaussian.cuh:30) changes the cells"""
    cfg = cartpole_cfg(K=1000, T=50, soft=True)
    eps = host_noise(1, cfg["K"], cfg["T"], 1)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    eng.injectNoise(eps)
    eng.computeControl(cfg["x0"], 1)
    u_fused = eng.getControlSeq().copy()
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    u_o32 = orc.control().copy()
    eng.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
    eng.updateImportanceSampler(np.zeros_like(u_fused))
    eng.computeControl(cfg["x0"], 1)
    assert np.array_equal(_bits(eng.getControlSeq()), _bits(u_o32))
    assert np.abs(u_fused - u_o32).max() <= U_TOL
    # cells of 7 rollouts (ragged: 1000 = 142 * 7 + 6)
    eng.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"], sum_strides=7)
    orc.set_sampler(cfg["std_dev"], cfg["control_cost_coeff"], sum_strides=7)
    orc.set_nominal_control(np.zeros_like(u_fused))
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    eng.updateImportanceSampler(np.zeros_like(u_fused))
    eng.computeControl(cfg["x0"], 1)
    assert np.array_equal(_bits(eng.getControlSeq()), _bits(orc.control()))
    # fma flavour (nvcc's default contraction of `inter += weight * v`)
    eng.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER_FMA)
    po.set_reduction_fma(True)
    try:
        orc.set_nominal_control(np.zeros_like(u_fused))
        orc.vanilla_compute_control(cfg["x0"], 1, eps)
    finally:
        po.set_reduction_fma(False)
    eng.updateImportanceSampler(np.zeros_like(u_fused))
    eng.computeControl(cfg["x0"], 1)
    assert np.array_equal(_bits(eng.getControlSeq()), _bits(orc.control()))
    eng.setReductionMode(m.MPPI_REDUCTION_FUSED)
    eng.updateImportanceSampler(np.zeros_like(u_fused))
    eng.computeControl(cfg["x0"], 1)
    assert np.array_equal(_bits(eng.getControlSeq()), _bits(u_fused))
    with pytest.raises(m.MPPIError) as e:
        eng.setReductionMode(7)
    assert e.value.status == 1
    eng.close()
    # a K-sharded handle refuses: the reference order runs over ALL rollouts
    sh = m.VanillaMPPIController("cartpole", 1024, 20, 0.02, 1.0, rank=0, world_size=2)
    with pytest.raises(m.MPPIError) as e:
        sh.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
    assert e.value.status == 10
    sh.close()


@pytest.mark.parametrize("tsallis", [False, True], ids=["exponential", "tsallis"])
def test_colored_mppi_reference_order(gpu, tsallis):
    """ColoredMPPI (colored-noise sampler; with and without Tsallis weights, core/mppi_common.cu:968-985) in the reference-order
    mode: four closed-loop steps, control sequence and state trajectory bit for bit"""
    from common import host_spectrum
    from test_colored_noise import _colored_cartpole
    cfg = _colored_cartpole(K=2048, T=60)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    eng.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
    exps, decay, fmin = cfg["colored"]
    if tsallis:
        eng.setColoredMPPIParams(gamma=400.0, r_exp=1.7)
        orc.set_colored_mppi_params(400.0, 1.7, None, False, 1)
    x = cfg["x0"].copy()
    for i in range(4):
        z = host_spectrum(1, cfg["K"], cfg["T"], 1, seed=60 + i)
        eng.injectNoise(z)
        eng.computeControl(x, 1)
        orc.colored_compute_control(x, 1, z, exps, decay, fmin)
        assert np.array_equal(_bits(eng.getControlSeq()), _bits(orc.control())), i
        assert np.array_equal(_bits(eng.getTargetStateSeq()), _bits(orc.state_traj())), i
        st, so = eng.getStats().real_sys, orc.stats()
        assert st.baseline == so["baseline"][0] and st.normalizer == so["normalizer"][0], i
        x, _ = orc.model_step(x, orc.control()[0])
        eng.slideControlSequence(1)
        orc.vanilla_slide(1)
    eng.close()


def test_tube_free_running_closed_loop_100_steps(gpu):
    """config 3 (double integrator, Tube-MPPI, K = 8192, T = 150): 100 control iterations with Tube's slide
    (tube_mppi_controller.cu:312-323: the nominal state takes a model step), the actual state disturbed every step, NEVER
    re-synchronised — both control sequences, both state trajectories and the statistics of both systems bit for bit"""
    cfg = di_cfg(K=8192, T=150, tube=True)
    cfg["ranges"] = [[-3.0, 3.0], [-3.0, 3.0]]
    eng, orc = make_engine(cfg), make_oracle(cfg)
    eng.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
    eng.setSeed(11)
    x_e, x_o = cfg["x0"].copy(), cfg["x0"].copy()
    rng = np.random.default_rng(0)
    used = set()
    for i in range(100):
        eps = po.philox_normal(11, i, cfg["K"], cfg["T"], 2)[None]
        eng.computeControl(x_e, 1)
        orc.tube_compute_control(x_o, 1, eps)
        u_e, u_o = eng.getControlSeq(), orc.control()
        assert np.abs(u_e - u_o).max() <= U_TOL, i
        assert np.array_equal(_bits(u_e), _bits(u_o)), i
        assert np.array_equal(_bits(eng.getNominalControlSeq()), _bits(orc.nominal_control())), i
        assert np.array_equal(_bits(eng.getTargetStateSeq()), _bits(orc.state_traj())), i
        assert np.array_equal(_bits(eng.getNominalStateSeq()), _bits(orc.nominal_state_traj())), i
        st, so = eng.getStats(), orc.stats()
        assert st.real_sys.baseline == so["baseline"][0] and st.nominal_sys.baseline == so["baseline"][1], i
        assert st.real_sys.normalizer == so["normalizer"][0] and st.nominal_sys.normalizer == so["normalizer"][1], i
        assert st.nominal_state_used == so["nominal_state_used"], i
        used.add(st.nominal_state_used)
        # the plant: the model itself + a disturbance that occasionally is large (the nominal system then takes over)
        kick = rng.normal(0, 0.02, 4).astype(np.float32)
        if i % 17 == 16:
            kick += np.array([0.8, -0.6, 0.5, 0.5], np.float32)
        x_e, _ = eng.modelStep(x_e, u_e[0])
        x_o, _ = orc.model_step(x_o, u_o[0])
        assert np.array_equal(_bits(x_e), _bits(x_o)), i
        x_e, x_o = x_e + kick, x_o + kick
        eng.slideControlSequence(1)
        orc.tube_slide(1)
    eng.close()


def _robust_pair(cfg, thr, nc=9, ns=32):
    eng = m.RobustMPPIController(cfg["model"], cfg["K"], cfg["T"], cfg["dt"], cfg["lambda_"], cfg["alpha"], cfg["num_iters"],
                                 seed=42)
    if cfg["dyn"] is not None:
        eng.setDynamicsParams(cfg["dyn"])
    eng.setCostParams(cfg["cost"])
    for name, blob in cfg.get("blobs", {}).items():
        eng.setModelBlob(name, blob)
    if cfg["ranges"] is not None:
        eng.setControlRanges(cfg["ranges"])
    eng.setSamplingParams(cfg["std_dev"], cfg["control_cost_coeff"])
    eng.setRMPPIParams(thr, nc, ns)
    eng.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
    orc = make_oracle(cfg)
    return eng, orc, po.RobustOracle(orc, thr, nc, ns)


@pytest.mark.parametrize("model,K,T,steps", [("di", 8192, 150, 100), ("autorally", 4096, 100, 40)])
def test_robust_free_running_closed_loop(gpu, model, K, T, steps):
    """Robust MPPI, free-running: updateImportanceSamplingControl (candidates, init-eval kernel, best index, slide) +
    computeControl per step with a disturbed real state, never re-synchronised — real and nominal control, nominal state
    trajectory, candidate free energies, best index and the statistics of both systems bit for bit on every step"""
    if model == "di":
        cfg = di_cfg(K=K, T=T, tube=True)
        cfg["control_cost_coeff"] = [0.3, 0.2]
        cfg["ranges"] = [[-3.0, 3.0], [-3.0, 3.0]]
        thr = 25.0
    else:
        cfg = autorally_cfg(K=K, T=T)
        cfg["D"] = 2
        cfg["control_cost_coeff"] = [0.2, 0.1]
        thr = 500.0
    eng, orc, rob = _robust_pair(cfg, thr)
    S, C = eng.STATE_DIM, eng.CONTROL_DIM
    g = np.random.default_rng(1).uniform(-0.3, 0.3, (T, S, C)).astype(np.float32)
    eng.setFeedbackGains(g)
    rob.set_gains(g)
    x_e, x_o = cfg["x0"].copy(), cfg["x0"].copy()
    rng = np.random.default_rng(2)
    used = set()
    for i in range(steps):
        eps = host_noise(2, K, T, C, seed=300 + i)
        first = i == 0
        eng.injectNoise(eps[1:] if first else eps)  # the first call does not evaluate candidates (nominal not set yet)
        eng.updateImportanceSamplingControl(x_e, 1)
        rob.update_importance_sampling(x_o, 1, eps[0])
        ns_g, best_g, stride_g, fe_g = eng.getRMPPIState()
        ns_o, best_o, stride_o, fe_o = rob.state()
        assert best_g == best_o and stride_g == stride_o, i
        assert np.array_equal(_bits(ns_g), _bits(ns_o)), i
        if not first:
            assert np.array_equal(_bits(fe_g), _bits(fe_o)), i
            used.add(best_g)
        eng.computeControl(x_e, 1)
        rob.compute_control(x_o, 1, eps[1:])
        u_e, u_o = eng.getControlSeq(), orc.control()
        assert np.abs(u_e - u_o).max() <= U_TOL, i
        assert np.array_equal(_bits(u_e), _bits(u_o)), i
        assert np.array_equal(_bits(eng.getNominalControlSeq()), _bits(orc.nominal_control())), i
        assert np.array_equal(_bits(eng.getTargetStateSeq()), _bits(orc.nominal_state_traj())), i
        st, so = eng.getStats(), orc.stats()
        assert st.nominal_sys.baseline == so["baseline"][0] and st.real_sys.baseline == so["baseline"][1], i
        assert st.nominal_sys.normalizer == so["normalizer"][0] and st.real_sys.normalizer == so["normalizer"][1], i
        x_e, _ = eng.modelStep(x_e, u_e[0])
        x_o, _ = orc.model_step(x_o, u_o[0])
        assert np.array_equal(_bits(x_e), _bits(x_o)), i
        kick = np.zeros(S, np.float32)
        kick[:4] = rng.normal(0, 0.02, 4)
        x_e, x_o = x_e + kick, x_o + kick
    eng.close()
