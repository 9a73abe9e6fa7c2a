"""100 closed-loop control iterations at the BASELINE sizes ("after 1 iteration and after 100 closed-loop
iterations", bar 1e-5).

A closed loop feeds every u* into the next state and the next mean, so a free-running engine and the oracle are two
chaotic systems started 1e-7 apart: their distance after n steps says how sensitive the PLANT is, not how accurate an
iteration is.  The two effects are therefore separated (round-2 review, "weak" 1):

  re-synchronised run   both sides start EVERY step from the oracle's state and the oracle's mean (and therefore the same
                        control history); asserted over all 100 steps:  trajectory costs 0 ulp, rho exact,
                        eta <= 1e-6 relative, u* <= 1e-5, state trajectory <= 1e-4
  free-running run,     a handle in MPPI_REDUCTION_REFERENCE_ORDER (the reference's own summation order for the last stage,
  reference order       exact_reduce_kernels.hpp) keeps its own state, mean and history for all 100 steps and is never
                        re-synchronised: ASSERTED on every step — u* <= 1e-5 against the oracle's loop (BASELINE.md §3's
                        literal criterion; in fact bit-identical: trajectory costs, rho, eta, u*, the plant state)
  free-running run,     the same with the default fused reduction (block-local softmin records + merge): u* of ONE
  default reduction     iteration is ~6e-8 from the reference's order, and a plant at the limit of grip amplifies that; its
                        distance to the oracle's loop and to the reference-order handle is REPORTED per step (printed, and
                        written to gpurun_out/closed_loop_drift.json), and only required to stay finite and to start below
                        the bar

Noise: the product's own in-kernel Philox stream (generation g = control iteration g) on the engine and the same stream
evaluated by the oracle, or (Cartpole, lambda = 200) noise injected from the host generator.
"""
import json
import os

import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import autorally_cfg, cartpole_cfg, host_noise, make_engine, make_oracle, ulp_diff

pytestmark = pytest.mark.gpu

U_TOL = 1e-5
STEPS = 100
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(name, drift):
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "closed_loop_drift.json")
    try:
        with open(path) as f:
            doc = json.load(f)
    except Exception:
        doc = {}
    doc[name] = drift
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)


def _closed_loop(cfg, name, seed=42, steps=STEPS, inject=False):
    K, T = cfg["K"], cfg["T"]
    C = len(cfg["std_dev"])
    sync, free, exact, orc = make_engine(cfg), make_engine(cfg), make_engine(cfg), make_oracle(cfg)
    exact.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
    sync.setSeed(seed)
    free.setSeed(seed)
    exact.setSeed(seed)
    x = cfg["x0"].copy()
    x_free = cfg["x0"].copy()
    x_exact = cfg["x0"].copy()
    worst_sync, drift_u, drift_x, exact_u, exact_bits, fused_vs_exact = 0.0, [], [], [], 0, []
    for i in range(steps):
        if inject:
            eps = host_noise(1, K, T, C, seed=1000 + i)
            sync.injectNoise(eps)
            free.injectNoise(eps)
            exact.injectNoise(eps)
        else:
            eps = po.philox_normal(seed, i, K, T, C)[None]
        sync.computeControl(x, 1)
        free.computeControl(x_free, 1)
        exact.computeControl(x_exact, 1)
        orc.vanilla_compute_control(x, 1, eps)
        u_o = orc.control().copy()
        # ---- free-running in the reference's summation order: never re-synchronised, asserted on every step
        u_e = exact.getControlSeq()
        de = float(np.abs(u_e - u_o).max())
        assert de <= U_TOL, "step %d: free-running reference-order u* differs by %g" % (i, de)
        assert np.abs(x_exact - x).max() <= 1e-5 * max(1.0, float(np.abs(x).max())), i
        exact_u.append(de)
        se, so_ = exact.getStats().real_sys, orc.stats()
        bit_equal = (np.array_equal(u_e, u_o) and np.array_equal(x_exact, x) and se.baseline == so_["baseline"][0] and
                     se.normalizer == so_["normalizer"][0] and
                     int(ulp_diff(exact.getSampledCostSeq(), orc.costs()).max()) == 0)
        exact_bits += int(bit_equal)
        # ---- re-synchronised: same state, same mean, same history -> one iteration's error, 100 times
        dc = int(ulp_diff(sync.getSampledCostSeq(), orc.costs()).max())
        assert dc == 0, "step %d: trajectory costs differ by %d ulp" % (i, dc)
        st, so = sync.getStats().real_sys, orc.stats()
        assert st.baseline == so["baseline"][0], i
        assert abs(st.normalizer - so["normalizer"][0]) <= 1e-6 * so["normalizer"][0], i
        du = float(np.abs(sync.getControlSeq() - u_o).max())
        assert du <= U_TOL, "step %d: u* differs by %g" % (i, du)
        assert np.abs(sync.getTargetStateSeq() - orc.state_traj()).max() <= 1e-4, i
        worst_sync = max(worst_sync, du)
        # ---- free-running: reported
        u_f = free.getControlSeq()
        assert np.isfinite(u_f).all()
        drift_u.append(float(np.abs(u_f - u_o).max()))
        drift_x.append(float(np.abs(x_free - x).max()))
        fused_vs_exact.append(float(np.abs(u_f - u_e).max()))
        # ---- advance: the plant is the model itself (examples/cartpole_example.cu:63-85)
        x_free, _ = free.modelStep(x_free, u_f[0])
        free.slideControlSequence(1)
        x_exact, _ = exact.modelStep(x_exact, u_e[0])
        exact.slideControlSequence(1)
        sync.updateImportanceSampler(u_o)  # the oracle's mean; the slide below derives the same history from it
        sync.slideControlSequence(1)
        x, _ = orc.model_step(x, u_o[0])
        orc.vanilla_slide(1)
    sync.close()  # leave no stream behind: the multi-rank tests need the hardware queues
    free.close()
    exact.close()
    assert drift_u[0] <= U_TOL
    rep = {"steps": steps, "K": K, "T": T, "resynchronised_worst_u": worst_sync,
           "free_running_reference_order_u_linf_max": max(exact_u), "free_running_reference_order_bit_identical_steps": exact_bits,
           "fused_vs_reference_order_u_linf": {"step1": fused_vs_exact[0], "step10": fused_vs_exact[min(9, steps - 1)],
                                               "step50": fused_vs_exact[min(49, steps - 1)], "max": max(fused_vs_exact)},
           "free_running_u_linf": {"step1": drift_u[0], "step10": drift_u[min(9, steps - 1)],
                                   "step50": drift_u[min(49, steps - 1)], "last": drift_u[-1], "max": max(drift_u)},
           "free_running_state_linf_last": drift_x[-1], "state_norm_last": float(np.abs(x).max())}
    print("\nclosed loop %s: %s" % (name, json.dumps(rep)))
    _report(name, rep)
    return rep


@pytest.mark.parametrize("soft", [False, True], ids=["lambda0.25", "lambda200"])
def test_cartpole_16384x100_closed_loop_100_steps(gpu, soft):
    """BASELINE headline configuration, 100 control iterations with slide (examples/cartpole_example.cu:63-85)"""
    rep = _closed_loop(cartpole_cfg(K=16384, T=100, soft=soft), "cartpole_16384x100_" + ("lambda200" if soft else "lambda0.25"),
                       inject=soft)
    assert rep["resynchronised_worst_u"] <= U_TOL
    assert rep["free_running_reference_order_u_linf_max"] <= U_TOL
    assert rep["free_running_reference_order_bit_identical_steps"] == STEPS


def test_autorally_16384x150_closed_loop_100_steps(gpu):
    """config 4 (NeuralNetModel on the MFMA forward + ARStandardCost), 100 control iterations"""
    rep = _closed_loop(autorally_cfg(K=16384, T=150), "autorally_16384x150", seed=7)
    assert rep["resynchronised_worst_u"] <= U_TOL
    # BASELINE.md §3, literally: L-inf(u*) <= 1e-5 after 100 FREE-RUNNING closed-loop iterations (asserted per step inside)
    assert rep["free_running_reference_order_u_linf_max"] <= U_TOL
    assert rep["free_running_reference_order_bit_identical_steps"] == STEPS
