"""Kernel-level operators and error behaviour through the C ABI (reference tests: tests/mppi_core/
normexp_kernel_tests.cu, weightedreduction_kernel_tests.cu, tests/controllers/controller_kernel_testing.cu)."""
import os

import numpy as np
import pytest

import mppi_generic_amd as m
import pyoracle as po
from common import cartpole_cfg, host_noise, make_engine, make_oracle, merge_records_numpy, ulp_diff

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("func,lo,hi", [(0, -50, 50), (1, -50, 50), (2, -100, 88), (3, 1e-30, 1e30), (4, -12, 12),
                                        (5, -100, 100), (6, -1000, 1000), (7, -30, 30), (8, 0, 1e20), (9, -1e3, 1e3),
                                        (10, -12, 12), (11, -20, 20), (4, -1e-3, 1e-3), (10, -1e-3, 1e-3), (12, -20, 20), (13, -1.01, 1.01)])
def test_det_math_device_equals_host_bitwise(gpu, func, lo, hi):
    """func 4 / 7 / 10 / 11: tanh and sigmoid run a hand-written division core on the device (det::div_benign, packed in
    tanh2) that must reproduce the host's IEEE division bit for bit — 2 M samples each, plus tiny and subnormal inputs"""
    rng = np.random.default_rng(func)
    n = 2_000_000 if func in (4, 7, 10, 11) else 200000
    x = rng.uniform(lo, hi, n).astype(np.float32)
    if func in (4, 10):
        tiny = np.exp(rng.uniform(np.log(1e-44), np.log(1e-3), 100000)).astype(np.float32)
        x[1000:101000] = tiny * np.where(rng.uniform(size=100000) < 0.5, -1, 1).astype(np.float32)
    if func == 3:
        x = np.exp(rng.uniform(np.log(1e-30), np.log(1e30), 200000)).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, np.float32(np.pi), 1e-40, 0.625, -0.625]
    if func in (3, 8):
        x = np.abs(x)
    a, b = m.det_eval(func, x), po.det_eval(func, x)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), "det func %d differs at x=%r: %r vs %r" % (func, x[~same][:4], a[~same][:4], b[~same][:4])


def test_norm_exp_kernel(gpu):
    """reference: tests/mppi_core/normexp_kernel_tests.cu:126-150"""
    rng = np.random.default_rng(0)
    costs = rng.uniform(0, 100, 10000).astype(np.float32)
    base = float(costs.min())
    w_gpu = m.norm_exp(costs, 1.0 / 0.5, base)
    w_cpu = po.norm_exp(costs, 1.0 / 0.5, base)
    assert ulp_diff(w_gpu, w_cpu).max() == 0


def test_compute_weights_two_pass(gpu):
    """reference: tests/mppi_core/normexp_kernel_tests.cu:180-256 (old host min/sum vs device weight transform)"""
    rng = np.random.default_rng(1)
    costs = rng.uniform(5, 60, 10000).astype(np.float32)
    w, base, eta = m.compute_weights(costs, 2.0)
    assert base == po.baseline(costs)
    w_cpu = po.norm_exp(costs, 2.0, base)
    assert ulp_diff(w, w_cpu).max() == 0
    assert abs(eta - po.normalizer(w_cpu)) <= 1e-6 * eta


def test_weighted_reduction_kernel(gpu):
    """reference: tests/mppi_core/weightedreduction_kernel_tests.cu:135-173"""
    rng = np.random.default_rng(2)
    K, T, C = 1024, 100, 4
    w = np.exp(-rng.normal(5.0, 1.2, K)).astype(np.float32)
    w[0] = 1.0
    v = rng.normal(5.0, 1.2, (K, T, C)).astype(np.float32)
    eta = float(w.astype(np.float64).sum())
    u_gpu = m.weighted_reduction(w, v, eta)
    u_cpu = po.weighted_reduction(w, v, eta, 64)
    assert np.abs(u_gpu - u_cpu).max() <= 1e-5 * np.abs(u_cpu).max()


def test_error_codes(gpu):
    with pytest.raises(m.MPPIError) as e:
        m.VanillaMPPIController("no_such_model", 128, 10, 0.02, 1.0)
    assert e.value.status == 2 and "cartpole" in str(e.value)
    with pytest.raises(m.MPPIError) as e:  # launch shape not instantiated (reference: exit(), mppi_common.cu:1266-1277)
        m.VanillaMPPIController("cartpole", 128, 10, 0.02, 1.0, block_x=48, block_y=3)
    assert e.value.status == 5
    # LDS overflow (reference: runtime_error, mppi_controller.cu:64-76).  Every rollout kernel moves its sample rows to HBM
    # when they do not fit (T = 2000 on the pipeline variant runs) and the post-processing kernels their control sequence
    # (tests/test_long_horizon.py: T = 25 000); what is left is the LDS + barrier variant of the post-processing kernel, which
    # collects its trajectories in LDS — a RACER model with a network of another shape than its four-lane form at T = 4000
    m.VanillaMPPIController("cartpole", 128, 2000, 0.02, 1.0, block_x=64, block_y=1, kernel_variant=2).close()
    from test_racer_dubins_lstm_steering import steering_cfg
    from common import make_engine
    rng = np.random.default_rng(2)
    cfg = steering_cfg(K=128, T=4000)
    Hn = 6
    blobs = {"lstm_structure": np.array([Hn, Hn + 4, 12, 1], np.float32),
             "lstm_weights": rng.uniform(-0.4, 0.4, 4 * Hn * Hn + 4 * Hn * 4 + 6 * Hn).astype(np.float32),
             "lstm_output_weights": rng.uniform(-0.4, 0.4, (Hn + 4) * 12 + 12 + 12 + 1).astype(np.float32)}
    maps = {k: v for k, v in cfg["blobs"].items() if k.startswith("elevation")}
    cfg["blobs"] = {**blobs, **maps}  # the structure first, then the weights
    big = make_engine(cfg)
    with pytest.raises(m.MPPIError) as e:
        big.computeControl(cfg["x0"], 1)
    assert e.value.status == 6
    big.close()
    with pytest.raises(m.MPPIError) as e:
        m.VanillaMPPIController("cartpole", 0, 10, 0.02, 1.0)
    assert e.value.status == 1
    c = m.VanillaMPPIController("cartpole", 128, 10, 0.02, 1.0)
    with pytest.raises(m.MPPIError) as e:
        c.setDynamicsParams(m.DoubleIntegratorParams())  # wrong struct size
    assert e.value.status == 1
    with pytest.raises(m.MPPIError) as e:
        c.getSampledControls()  # created without save_samples
    assert e.value.status == 7


def test_nan_state_reports_error(gpu):
    """reference: base_plant.hpp:515-535 exits on NaN controls; here MPPI_ERR_NAN is returned"""
    cfg = cartpole_cfg(K=256, T=20, soft=True)
    eng = make_engine(cfg)
    with pytest.raises(m.MPPIError) as e:
        eng.computeControl(np.array([np.nan, 0, 0, 0], np.float32), 1)
    assert e.value.status == 8


def test_two_way_sharding_on_one_gpu(gpu):
    """K split over two handles (rank 0/1 of world 2) with the exchange done by hand reproduces the unsharded u* —
    the K-sharding of SURVEY.md §8e, exercised on one device."""
    import ctypes as C
    cfg = cartpole_cfg(K=2048, T=100, soft=True)
    full = make_engine(cfg)
    full.uploadState(cfg["x0"])
    full.optimize(1)
    u_full = full.getOptimalControlSeq()[0]
    st_full = full.getStats().real_sys

    r = [make_engine(cfg, rank=i, world_size=2) for i in range(2)]
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for c in r:
        assert c.num_rollouts_local == 1024
        c.uploadState(cfg["x0"])
        c.iterationLocal()
        c.synchronize()
    bufs = [c.exchangeBuffers() for c in r]
    n = bufs[0][2]
    for dst in range(2):
        for src in range(2):
            # recv[dst][src] <- send[src]   (what the all-gather does); 3 = hipMemcpyDeviceToDevice
            assert hip.hipMemcpy(bufs[dst][1] + 4 * n * src, bufs[src][0], 4 * n, 3) == 0
    for c in r:
        c.iterationMerge()
        c.synchronize()
        u = c.getOptimalControlSeq()[0]
        assert np.abs(u - u_full).max() <= 2e-6, np.abs(u - u_full).max()
        st = c.getStats().real_sys
        assert st.baseline == st_full.baseline
        assert abs(st.normalizer - st_full.normalizer) <= 2e-6 * st_full.normalizer
    # and against the oracle's un-sharded iteration on the oracle's own generator stream
    orc = make_oracle(cfg)
    eps = po.philox_normal(42, 0, cfg["K"], cfg["T"], 1)
    u_orc = orc.iterate(cfg["x0"], np.zeros((cfg["T"], 1), np.float32), eps)[0]
    assert np.abs(u_full - u_orc).max() <= 1e-5


def test_rccl_exchange_path_on_one_gpu(gpu):
    """the library's own RCCL driver (dlopen, ncclGetUniqueId, ncclCommInitRank, ncclAllGather on the handle's stream) with a
    world of ONE rank: force_exchange routes every iteration through local merge -> all-gather -> global merge"""
    import ctypes as C
    cfg = cartpole_cfg(K=2048, T=100, soft=True)
    plain = make_engine(cfg)
    plain.uploadState(cfg["x0"])
    plain.optimize(3)
    lib = m.load_library()
    buf = C.create_string_buffer(128)
    nb = C.c_size_t()
    assert lib.mppi_rccl_unique_id(buf, 128, C.byref(nb)) == 0 and nb.value == 128
    eng = make_engine(cfg, force_exchange=True)
    with pytest.raises(m.MPPIError) as e:  # exchange requested but no communicator yet
        eng.uploadState(cfg["x0"])
        eng.optimize(1)
    assert e.value.status == 7
    eng.setSeed(42)
    eng.commInitRccl(bytes(buf.raw))
    eng.uploadState(cfg["x0"])
    eng.optimize(3)
    assert np.abs(eng.getOptimalControlSeq() - plain.getOptimalControlSeq()).max() <= 2e-6
    assert eng.getStats().real_sys.baseline == plain.getStats().real_sys.baseline


HOST_STAGED_SCRIPT = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
from common import cartpole_cfg, make_engine
from mppi_generic_amd.distributed import HostStagedExchange, ShardedController, hip_runtimes_in_process
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1)
cfg = cartpole_cfg(K=2048, T=100, soft=True)
plain = make_engine(cfg)
plain.uploadState(cfg["x0"])
plain.optimize(3)
eng = make_engine(cfg, force_exchange=True)
hx = HostStagedExchange(eng)
eng.uploadState(cfg["x0"])
hx.iterate(3)
assert np.abs(eng.getOptimalControlSeq() - plain.getOptimalControlSeq()).max() <= 2e-6
# the zero-copy variant refuses to run across two HIP runtimes (this image's torch wheel bundles its own ROCm)
if torch.cuda.is_available() and len(hip_runtimes_in_process()) > 1:
    side = torch.cuda.Stream()
    eng2 = make_engine(cfg, force_exchange=True, stream=side.cuda_stream)
    try:
        ShardedController(eng2, side)
        raise SystemExit("ShardedController accepted two HIP runtimes")
    except RuntimeError:
        pass
dist.destroy_process_group()
print("HOST_STAGED_OK")
sys.stdout.flush()
os._exit(0)
"""


def test_host_staged_exchange_on_one_gpu(gpu):
    """the external driver with host-staged records (mppi-generic_amd/distributed.py HostStagedExchange) over a gloo group
    of one rank: local merge -> D2H -> all-gather -> H2D -> global merge reproduces the plain result.  Runs in its own
    process: torch brings a second ROCm runtime into the process, and whether that coexists with an already loaded
    system RCCL depends on the import order of everything that ran before."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "REPO = %r\n" % repo + HOST_STAGED_SCRIPT], capture_output=True, text=True,
                       timeout=600)
    assert "HOST_STAGED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_full_size_properties(gpu):
    """BASELINE size (K=16384, T=100): size-independent properties instead of a slow oracle run —
    (1) baseline == min of the costs, (2) normaliser == sum of exp weights, (3) u* == direct weighted mean of the
    dumped samples, (4) rollout 0 is noise-free, (5) the result does not depend on the block shape."""
    cfg = cartpole_cfg(K=16384, T=100, soft=True)
    eng = make_engine(cfg, save_samples=True)
    eng.uploadState(cfg["x0"])
    eng.optimize(1)
    costs = eng.getSampledCostSeq()[0]
    v = eng.getSampledControls()[0]
    st = eng.getStats().real_sys
    assert st.baseline == costs.min()
    w = np.exp(-(costs.astype(np.float64) - costs.min()) / cfg["lambda_"])
    assert abs(st.normalizer - w.sum()) <= 1e-5 * w.sum()
    assert np.all(v[0] == 0.0)  # k = 0 takes the (zero) mean exactly
    assert np.abs(v).max() <= 5.0  # clamped
    # second handle, different block shape, same seed -> same costs
    eng2 = make_engine(cfg, block_x=32)
    eng2.uploadState(cfg["x0"])
    eng2.optimize(1)
    assert np.array_equal(eng2.getSampledCostSeq()[0], costs)


def test_choose_appropriate_kernel(gpu):
    """reference: VanillaMPPI::chooseAppropriateKernel (mppi_controller.cu:44-143) — both structures timed, the faster kept,
    results unchanged (the structures are bit-identical) and the noise stream not advanced by the trial launches"""
    cfg = cartpole_cfg(K=16384, T=100, soft=True)
    ref = make_engine(cfg)
    ref.uploadState(cfg["x0"])
    ref.optimize(2)
    eng = make_engine(cfg)
    eng.uploadState(cfg["x0"])
    v, fused_ms, pipe_ms = eng.chooseAppropriateKernel(5)
    assert v in (1, 2) and np.isfinite(fused_ms) and np.isfinite(pipe_ms)
    assert (v == 2) == (pipe_ms < fused_ms)
    eng.optimize(2)
    assert np.array_equal(eng.getOptimalControlSeq(), ref.getOptimalControlSeq())
    # a configuration with one structure only: block shape (64, 4) of the LDS contract variant
    one = make_engine(cartpole_cfg(K=512, T=30), block_x=64, block_y=4)
    one.uploadState(cfg["x0"])
    v, fused_ms, pipe_ms = one.chooseAppropriateKernel(2)
    assert v == 1 and np.isinf(pipe_ms)


def test_rocrand_host_noise_source(gpu):
    """MPPI_NOISE_ROCRAND_HOST: rocrand_generate_normal (PHILOX4_32_10) fills an eps buffer in HBM before every rollout launch
    — the reference's structure (curandGenerateNormal, gaussian.cu:380-394).  Statistical checks only, like the reference's
    own sampler tests: the stream is rocRAND's, not the in-kernel Philox order."""
    cfg = cartpole_cfg(K=8192, T=64, soft=True)
    eng = make_engine(cfg, noise_source=2, save_samples=True)
    eng.uploadState(cfg["x0"])
    eng.optimize(1)
    v0 = eng.getSampledControls()[0].copy()
    # sigma = 5, clamp to [-5, 5]: look at the unclamped interior through the quantiles of the standard normal
    body = v0[1:int(0.99 * cfg["K"])]  # without the noise-free rollout 0 and the zero-mean tail (mean is 0 here anyway)
    frac_inside = np.mean(np.abs(body) < 4.999)
    assert abs(frac_inside - 0.6827) < 0.01  # |eps| < 1  <=>  |v| < sigma
    assert abs(np.median(body)) < 0.05
    assert np.all(v0[0] == 0.0)
    eng.updateImportanceSampler(np.zeros((cfg["T"], 1), np.float32))
    eng.uploadState(cfg["x0"])
    eng.optimize(1)
    v1 = eng.getSampledControls()[0]
    assert np.mean(v1[1:100] == v0[1:100]) < 0.3  # the next generation draws a new stretch of the stream (clamped values tie)
    # two ranks of a sharded problem draw disjoint stretches
    r = [make_engine(cfg, noise_source=2, save_samples=True, rank=i, world_size=2) for i in range(2)]
    for c in r:
        c.uploadState(cfg["x0"])
        c.iterationLocal()
        c.synchronize()
    a, b = r[0].getSampledControls()[0], r[1].getSampledControls()[0]
    assert np.mean(a[1:100] == b[1:100]) < 0.3
    # and the optimisation behaves like the Philox mode on the same problem (same baseline to a few percent)
    ph = make_engine(cfg)
    ph.uploadState(cfg["x0"])
    ph.optimize(1)
    assert abs(eng.getStats().real_sys.baseline / ph.getStats().real_sys.baseline - 1.0) < 0.05
