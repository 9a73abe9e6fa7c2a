"""One launch per iteration: a Vanilla handle's pipelined rollout kernel merges the PREVIOUS iteration's per-block records in
its sampler waves (rollout_pipeline_kernel.hpp, STREAM_MERGE; the arithmetic is merge_wave.hpp, the same functions
combineKernel runs), so the merge launch between two iterations goes away and only the last iteration's records are merged by
combineKernel.  The reference has no such stage to compare with (it launches rollout, normExp and weightedReduction per
iteration, controllers/MPPI/mppi_controller.cu:128-236); what the test pins is that the streamed form is the SAME FUNCTION as
the two-launch form — every u*, every statistic, every trajectory cost, bit for bit — and that it really is the form that ran.
"""
import os

import numpy as np
import pytest

from common import cartpole_cfg, cartpole_cfg_lr, di_cfg, make_engine, make_oracle

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _pair(cfg, **kw):
    """(streamed handle, two-launch handle) of the same configuration; the switch is read at mppi_create"""
    streamed = make_engine(cfg, **kw)
    os.environ["MPPI_AMD_NO_STREAM_MERGE"] = "1"
    try:
        plain = make_engine(cfg, **kw)
    finally:
        del os.environ["MPPI_AMD_NO_STREAM_MERGE"]
    return streamed, plain


def _stats(e):
    s = e.getStats().real_sys
    return np.array([s.baseline, s.normalizer, s.free_energy_mean, s.free_energy_variance, s.free_energy_modified_variance],
                    np.float32)


@pytest.mark.parametrize("cfg", [
    cartpole_cfg(K=16384, T=100, soft=True, num_iters=4),
    cartpole_cfg(K=16384, T=100, soft=False, num_iters=3),
    cartpole_cfg(K=2048, T=52, soft=True, num_iters=5),
    cartpole_cfg(K=1000, T=100, soft=True, num_iters=3),   # ragged last block
    cartpole_cfg(K=64, T=8, soft=True, num_iters=6),       # one block, one dynamics trip
    dict(cartpole_cfg_lr(K=4096, T=100), num_iters=4),     # alpha, control cost, terminal cost: the LR term reads the merged mean
    di_cfg(K=4096, T=60, tube=False, num_iters=4),         # two controls per step
], ids=["cartpole-baseline-soft", "cartpole-baseline-sharp", "cartpole-2048x52", "cartpole-ragged", "cartpole-tiny",
        "cartpole-lr", "di"])
def test_streamed_merge_is_the_two_launch_iteration(gpu, cfg):
    a, b = _pair(cfg)
    x = cfg["x0"].copy()
    for step in range(6):
        a.computeControl(x, 1)
        b.computeControl(x, 1)
        assert np.array_equal(_bits(a.getControlSeq()), _bits(b.getControlSeq())), step
        assert np.array_equal(_bits(_stats(a)), _bits(_stats(b))), step
        assert np.array_equal(_bits(a.getSampledCostSeq()), _bits(b.getSampledCostSeq())), step
        x, _ = a.modelStep(x, a.getControlSeq()[0])
        a.slideControlSequence(1)
        b.slideControlSequence(1)
    n = cfg["num_iters"]
    ra, ga = a.launchCounts()
    rb, gb = b.launchCounts()
    assert (ra, rb) == (6 * n, 6 * n)
    assert gb == 6 * n          # the two-launch form merges after every rollout launch
    assert ga == 6, (ga, n)     # the streamed form: once per computeControl (the last iteration's records)
    a.close()
    b.close()


def test_streamed_merge_against_the_oracle(gpu):
    """the streamed iterations against the CPU restatement (fused Philox draw, so the oracle draws the same stream)"""
    import pyoracle as po
    cfg = cartpole_cfg(K=2048, T=100, soft=True, num_iters=3)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    eps = np.stack([po.philox_normal(42, g, cfg["K"], cfg["T"], 1) for g in range(3)])
    eng.computeControl(cfg["x0"], 1)
    orc.vanilla_compute_control(cfg["x0"], 1, eps)
    assert eng.launchCounts() == (3, 1)
    assert np.abs(eng.getControlSeq() - orc.control()).max() <= 1e-5
    eng.close()


def test_streamed_merge_optimize_and_settings_between_launches(gpu):
    """mppi_optimize (no readback between iterations) and a switch of the reduction mode with records still pending"""
    import mppi_generic_amd as m
    cfg = cartpole_cfg(K=4096, T=100, soft=True, num_iters=1)
    a, b = _pair(cfg)
    for e in (a, b):
        e.uploadState(cfg["x0"])
        e.optimize(7)
    assert np.array_equal(_bits(a.getControlSeq()), _bits(b.getControlSeq()))
    assert np.array_equal(_bits(_stats(a)), _bits(_stats(b)))
    # the reference-order reduction does not stream: the handle must fall back to two launches and stay equal to its twin
    for e in (a, b):
        e.setReductionMode(m.MPPI_REDUCTION_REFERENCE_ORDER)
        e.optimize(3)
        e.setReductionMode(m.MPPI_REDUCTION_FUSED)
        e.optimize(2)
    assert np.array_equal(_bits(a.getControlSeq()), _bits(b.getControlSeq()))
    a.close()
    b.close()


@pytest.mark.parametrize("what", ["T_not_multiple_of_4", "colored"])
def test_configurations_the_streamed_merge_leaves_alone(gpu, what):
    """T*C % 4 != 0, colored noise (and with it the Tsallis weights): two launches per iteration, results as before"""
    if what == "T_not_multiple_of_4":
        cfg = cartpole_cfg(K=1024, T=37, soft=True, num_iters=3)
    else:
        cfg = cartpole_cfg(K=1024, T=64, soft=True, num_iters=3)
        if what == "colored":
            cfg["colored"] = ([1.0], 0.97, 0.0)
    e = make_engine(cfg)
    e.computeControl(cfg["x0"], 1)
    r, g = e.launchCounts()
    assert r == 3 and g >= 3
    e.close()
