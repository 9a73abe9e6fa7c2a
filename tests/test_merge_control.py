"""The merge inside the control phase (finalize_kernel.hpp: mergeControlKernel): on a split hand-over of a model with the plain
one-wave finalize form, mppi_compute_control merges the last rollout launch's block records in the launch that smooths,
constrains and hands the control sequence over — read from the TRANSPOSED copy of the records the rollout kernels leave
(rollout_kernel.hpp: RolloutArgs::records_t_d) — instead of launching combineKernel first.  The reference has no such stage
(normExp + weightedReduction + smoothing are host-sequenced launches, controllers/MPPI/mppi_controller.cu:196-241); what is
pinned here is that the fused launch is the SAME FUNCTION as combineKernel + finalizeKernel's control phase: control sequence,
statistics, trajectories and the device-resident mean, bit for bit, and against the CPU oracle within the usual bound.
"""
import os

import numpy as np
import pytest

from common import cartpole_cfg, cartpole_cfg_lr, di_cfg, make_engine, make_oracle

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _pair(cfg, **kw):
    """(handle whose control phase merges, handle with combineKernel + control phase); the switch is read at mppi_create"""
    fused = make_engine(cfg, **kw)
    os.environ["MPPI_AMD_NO_MERGE_CONTROL"] = "1"
    try:
        plain = make_engine(cfg, **kw)
    finally:
        del os.environ["MPPI_AMD_NO_MERGE_CONTROL"]
    return fused, plain


def _stats(e):
    s = e.getStats().real_sys
    return np.array([s.baseline, s.normalizer, s.free_energy_mean, s.free_energy_variance, s.free_energy_modified_variance],
                    np.float32)


@pytest.mark.parametrize("cfg", [
    cartpole_cfg(K=16384, T=100, soft=True, num_iters=1),   # the compute_control block of bench.py
    cartpole_cfg(K=16384, T=100, soft=False, num_iters=3),
    cartpole_cfg(K=1000, T=100, soft=True, num_iters=2),    # ragged last block, 16 records
    cartpole_cfg(K=64, T=8, soft=True, num_iters=2),        # one record, two column quads: most waves idle
    cartpole_cfg(K=8192, T=200, soft=True, num_iters=1),    # 50 quads: a third round of loads inside the kernel
    dict(cartpole_cfg_lr(K=4096, T=100), num_iters=2),
    di_cfg(K=4096, T=60, tube=False, num_iters=2),          # two controls per step
], ids=["cartpole-baseline", "cartpole-baseline-sharp", "cartpole-ragged", "cartpole-tiny", "cartpole-T200", "cartpole-lr", "di"])
def test_merging_control_phase_equals_merge_then_control_phase(gpu, cfg):
    a, b = _pair(cfg)
    x = cfg["x0"].copy()
    for step in range(5):
        a.computeControl(x, 1)
        b.computeControl(x, 1)
        assert np.array_equal(_bits(a.getControlSeq()), _bits(b.getControlSeq())), step
        assert np.array_equal(_bits(_stats(a)), _bits(_stats(b))), step
        assert np.array_equal(_bits(a.getTargetStateSeq()), _bits(b.getTargetStateSeq())), step
        x, _ = a.modelStep(x, a.getControlSeq()[0])
        a.slideControlSequence(1)
        b.slideControlSequence(1)
    # the device-resident mean and statistics (what mppi_optimize and the operators continue from) are the merged ones
    for e in (a, b):
        e.optimize(2)
    assert np.array_equal(_bits(a.getControlSeq()), _bits(b.getControlSeq()))
    assert np.array_equal(_bits(_stats(a)), _bits(_stats(b)))
    a.close()
    b.close()


def test_merging_control_phase_against_the_oracle(gpu):
    import pyoracle as po
    cfg = cartpole_cfg(K=2048, T=100, soft=True, num_iters=1)
    eng, orc = make_engine(cfg), make_oracle(cfg)
    x = cfg["x0"].copy()
    for call in range(3):
        eps = np.stack([po.philox_normal(42, call, cfg["K"], cfg["T"], 1)])
        eng.computeControl(x, 1)
        orc.vanilla_compute_control(x, 1, eps)
        assert np.abs(eng.getControlSeq() - orc.control()).max() <= 1e-5, call
        eng.slideControlSequence(1)
        orc.vanilla_slide(1)
    eng.close()


def test_handles_the_merging_control_phase_leaves_alone(gpu):
    """no split hand-over (MPPI_AMD_SPLIT_FINALIZE=0), T*C not a multiple of 4: combineKernel + finalize as before"""
    cfg = cartpole_cfg(K=2048, T=100, soft=True, num_iters=2)
    os.environ["MPPI_AMD_SPLIT_FINALIZE"] = "0"
    try:
        unsplit = make_engine(cfg)
    finally:
        del os.environ["MPPI_AMD_SPLIT_FINALIZE"]
    fused = make_engine(cfg)
    for e in (unsplit, fused):
        e.computeControl(cfg["x0"], 1)
    assert np.array_equal(_bits(unsplit.getControlSeq()), _bits(fused.getControlSeq()))
    assert np.array_equal(_bits(_stats(unsplit)), _bits(_stats(fused)))
    unsplit.close()
    fused.close()
    cfg = cartpole_cfg(K=1024, T=37, soft=True, num_iters=2)
    a, b = _pair(cfg)
    for e in (a, b):
        e.computeControl(cfg["x0"], 1)
    assert np.array_equal(_bits(a.getControlSeq()), _bits(b.getControlSeq()))
    a.close()
    b.close()
