"""Accuracy of include/mppi_amd/det_math.h against float64 libm (CPU side; the GPU side is checked bit-for-bit against
this same host build in tests/test_gpu_ops.py::test_det_math_device_equals_host_bitwise)."""
import numpy as np
import pytest

import pyoracle as po


def _ulp_err(got, want64):
    want = want64.astype(np.float32)
    ulp = np.abs(np.spacing(want)).astype(np.float64)
    ulp = np.maximum(ulp, 1.4e-45)
    return np.abs(got.astype(np.float64) - want64) / ulp


@pytest.mark.parametrize("func,ref,lo,hi,tol", [
    (0, np.sin, -1000, 1000, 2.0), (1, np.cos, -1000, 1000, 2.0), (2, np.exp, -87, 88, 2.0),
    (4, np.tanh, -12, 12, 7.0), (5, np.arctan, -1e4, 1e4, 3.0),
])
def test_accuracy(func, ref, lo, hi, tol):
    rng = np.random.default_rng(func)
    x = rng.uniform(lo, hi, 1_000_000).astype(np.float32)
    got = po.det_eval(func, x)
    assert _ulp_err(got, ref(x.astype(np.float64))).max() <= tol


def test_asin_accuracy_and_edges():
    rng = np.random.default_rng(13)
    x = rng.uniform(-1, 1, 1_000_000).astype(np.float32)
    assert _ulp_err(po.det_eval(13, x), np.arcsin(x.astype(np.float64))).max() <= 2.5
    e = po.det_eval(13, np.array([0.0, 1.0, -1.0, 1.0000001, np.nan, 5e-5, 0.5], np.float32))
    assert e[0] == 0 and e[1] == np.float32(np.pi / 2) and e[2] == -e[1] and np.isnan(e[3]) and np.isnan(e[4])
    assert e[5] == np.float32(5e-5) and abs(e[6] - np.pi / 6) < 1e-7


def test_log_accuracy_and_edges():
    rng = np.random.default_rng(9)
    x = np.exp(rng.uniform(np.log(1e-38), np.log(1e38), 1_000_000)).astype(np.float32)
    assert _ulp_err(po.det_eval(3, x), np.log(x.astype(np.float64))).max() <= 1.5
    e = po.det_eval(3, np.array([1.0, 0.0, -1.0, np.inf, 1e-45], np.float32))
    assert e[0] == 0 and e[1] == -np.inf and np.isnan(e[2]) and e[3] == np.inf and abs(e[4] - np.log(1.4e-45)) < 1e-3


def test_exp_edges_and_tanh_sigmoid():
    e = po.det_eval(2, np.array([0.0, -104.0, -200.0, 89.0, np.nan, -87.5, -100.0], np.float32))
    assert e[0] == 1 and e[1] == 0 and e[2] == 0 and e[3] == np.inf and np.isnan(e[4])
    assert abs(e[5] / np.exp(-87.5) - 1) < 1e-6 and abs(e[6] - np.exp(-100.0)) <= 1.5e-45  # subnormal range: 1 ulp
    t = po.det_eval(4, np.array([0.0, 20.0, -20.0, 0.3], np.float32))
    assert t[0] == 0 and abs(t[1] - 1) < 3e-7 and t[2] == -t[1] and abs(t[3] - np.tanh(0.3)) < 1e-7  # saturates 4 ulp below 1
    s = po.det_eval(7, np.array([0.0, 2.0], np.float32))  # device-flavour sigmoid (activation_functions.cuh:49-59)
    assert s[0] == 0.5 and abs(s[1] - 1 / (1 + np.exp(-2.0))) < 1e-7


def test_normalize_angle_matches_fmodf_formula():
    """reference: utils/angle_utils.cuh:21-27 — fmodf(a + pi, 2pi) -/+ pi in float; fmod is exact, so equality is bitwise"""
    rng = np.random.default_rng(1)
    a = rng.uniform(-1e4, 1e4, 500000).astype(np.float32)
    pi = np.float32(np.pi)
    r = np.fmod((a + pi).astype(np.float32), np.float32(2) * pi).astype(np.float32)
    want = np.where(r <= 0, r + pi, r - pi).astype(np.float32)
    assert np.array_equal(po.det_eval(6, a), want)


def test_libm_flavour_deviation():
    """common-mode check: oracle and engine share det_math.h, so their 0-ulp agreement says nothing about the distance to an
    implementation on other transcendentals (the reference's CUDA path: __sinf / __cosf / tanhf / expf).  The same oracle
    rebuilt on glibc's sinf / cosf / expf / logf / tanhf / atanf (make -C oracle libm) stays within the north-star bar:
    u* L-inf <= 1e-5, trajectory costs <= 1e-4 relative (the reference's own GPU-vs-CPU tolerance).  At the BASELINE sizes
    tools/libm_flavour_study.py measures u* <= 4.1e-7 and costs <= 1.3e-5."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "libm_flavour_study", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "libm_flavour_study.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys
    argv = sys.argv
    sys.argv = ["libm_flavour_study.py", "--small"]
    try:
        res = mod.main()
    finally:
        sys.argv = argv
    for name, rel_cost, du in res:
        assert du <= 1e-5, (name, du)
        assert rel_cost <= 1e-4, (name, rel_cost)
