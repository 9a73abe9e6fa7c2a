"""Out-of-tree model registration (mppi_register_model / mppi_load_plugin; include/mppi_amd/engine/model_registry.hpp).

The reference's user instantiates the controller templates with their own Dynamics / Cost in their own translation unit
(src/controllers/cartpole/cartpole_mppi.cu:30-42).  Here examples/my_model/pendulum_model.hip is compiled on its own with
hipcc into a library of its own and loaded into the engine at run time; nothing of libmppi_amd.so is rebuilt."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import mppi_generic_amd as m
from common import SEED, host_noise

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "examples", "my_model", "pendulum_model.hip")
OUT_DIR = os.path.join(REPO, "examples", "_build")
PLUGIN = os.path.join(OUT_DIR, "libpendulum_model.so")


class PendulumParams(C.Structure):
    _fields_ = [("mass", C.c_float), ("length", C.c_float), ("damping", C.c_float), ("gravity", C.c_float)]


class PendulumCostParams(C.Structure):
    _fields_ = [("control_cost_coeff", C.c_float * 1), ("discount", C.c_float), ("angle_coeff", C.c_float),
                ("velocity_coeff", C.c_float), ("terminal_coeff", C.c_float), ("goal_angle", C.c_float)]


SRC_REF_STYLE = os.path.join(REPO, "examples", "my_model", "pendulum_model_reference_style.hip")
PLUGIN_REF_STYLE = os.path.join(OUT_DIR, "libpendulum_model_reference_style.so")


def build_plugin(src=SRC, out=PLUGIN):
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(REPO, "examples", "my_model", "pendulum_reference_style.cuh")))
    for d, _, files in os.walk(os.path.join(REPO, "include")):
        for f in files:
            newest = max(newest, os.path.getmtime(os.path.join(d, f)))
    if os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-I" + os.path.join(REPO, "include"), src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


@pytest.fixture(scope="module")
def plugin():
    lib = m.load_library()
    assert lib.mppi_load_plugin(build_plugin().encode()) == 0, lib.mppi_last_error(None)
    return lib


def test_plugin_builds_alone_and_registers(plugin):
    """the user's translation unit compiles against include/ only, and its static initialiser registers the model"""
    assert "user_pendulum" in plugin.mppi_list_models().decode().split("\n")
    # the plugin's own kernels live in the plugin, not in libmppi_amd.so
    syms = subprocess.run(["nm", "-D", "--defined-only", PLUGIN], capture_output=True, text=True).stdout
    assert "PendulumDynamics" in syms and "mppi_create" not in syms


def test_register_model_argument_checks(plugin):
    fn = C.cast(plugin.mppi_device_count, C.c_void_p)  # any non-null function pointer
    assert plugin.mppi_register_model(None, 0, fn, 0) == 1
    assert plugin.mppi_register_model(b"x", 7, fn, 0) == 1
    assert plugin.mppi_register_model(b"x", 0, fn, 12345) == 1  # header / library skew: the ABI fingerprint differs
    assert b"different mppi_amd/engine/model_instance.hpp" in plugin.mppi_last_error(None)
    assert plugin.mppi_load_plugin(b"/nonexistent/libnothing.so") == 1


def test_barrier_bearing_model_forced_onto_the_pipeline_is_refused_at_registration(plugin):
    """tests/probes/pendulum_forced_pipeline.hip: the reference-style pendulum — a step() with the reference's two
    __syncthreads() (dynamics/dynamics.cu:138,140), no MPPI_BARRIER_FREE_STEP declaration — registered with PIPELINE = true.
    Launched, its dynamics wave would wait at a barrier the sampler and cost waves never reach (a GPU hang, no status).  The
    registration is refused instead: mppi_load_plugin fails with MPPI_ERR_INVALID_ARG and says why, the model never enters the
    table, and mppi_create of its name is an unknown-model error.  Runs without a GPU (nothing is launched)."""
    src = os.path.join(REPO, "tests", "probes", "pendulum_forced_pipeline.hip")
    out = build_plugin(src, os.path.join(OUT_DIR, "libpendulum_forced_pipeline.so"))
    rc = plugin.mppi_load_plugin(out.encode())
    assert rc == 1, rc  # MPPI_ERR_INVALID_ARG
    msg = plugin.mppi_last_error(None).decode()
    assert "MPPI_BARRIER_FREE_STEP" in msg and "user_pendulum_forced_pipeline" in msg, msg
    assert "user_pendulum_forced_pipeline" not in plugin.mppi_list_models().decode().split("\n")
    # the C entry itself: ROLE_SEPARATED without BARRIER_FREE_DECLARED is refused, with it or without ROLE_SEPARATED it is not
    fn = C.cast(plugin.mppi_device_count, C.c_void_p)
    assert plugin.mppi_register_model_checked(b"probe_forced", 0, fn, 0, 1) == 1
    assert b"MPPI_BARRIER_FREE_STEP" in plugin.mppi_last_error(None)
    assert plugin.mppi_register_model_checked(b"probe_forced", 0, fn, 12345, 3) == 1  # passes the flag check, fails the fingerprint
    assert b"different mppi_amd/engine/model_instance.hpp" in plugin.mppi_last_error(None)


@pytest.mark.gpu
def test_barrier_bearing_pipeline_model_registered_unchecked_is_refused_at_create(gpu, plugin):
    """the second line of defence: the probe library also registers the forced instantiation through the UNCHECKED
    mppi_register_model (which cannot know what the factory builds) — it enters the table, and mppi_create turns it down
    (MPPI_ERR_INVALID_ARG, naming the classes without MPPI_BARRIER_FREE_STEP) before anything is launched"""
    src = os.path.join(REPO, "tests", "probes", "pendulum_forced_pipeline.hip")
    out = build_plugin(src, os.path.join(OUT_DIR, "libpendulum_forced_pipeline.so"))
    plugin.mppi_load_plugin(out.encode())  # (reports the checked registration's refusal; the unchecked one went through)
    assert "user_pendulum_forced_pipeline_unchecked" in plugin.mppi_list_models().decode().split("\n")
    with pytest.raises(m.MPPIError) as e:
        m.VanillaMPPIController("user_pendulum_forced_pipeline_unchecked", 256, 20, 0.02, 1.0)
    assert e.value.status == 1 and "MPPI_BARRIER_FREE_STEP" in str(e.value) and "Dynamics" in str(e.value), str(e.value)


def test_in_tree_models_declare_barrier_free_steps_and_reference_style_ones_do_not():
    """the trait the engine decides on (plugin/parallel_utils.hpp: barrier_free_step): compiled as static_asserts — host only"""
    probe = os.path.join(OUT_DIR, "barrier_trait_probe.hip")
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(probe, "w") as f:
        f.write("""#include <mppi/dynamics/cartpole/cartpole_dynamics.cuh>
#include <mppi/cost_functions/cartpole/cartpole_quadratic_cost.cuh>
#include <mppi/dynamics/double_integrator/di_dynamics.cuh>
#include <mppi/cost_functions/double_integrator/double_integrator_circle_cost.cuh>
#include <mppi/controllers/MPPI/mppi_controller.cuh>
#include "my_model/pendulum_reference_style.cuh"
using G = mppi::sampling_distributions::GaussianDistribution<CartpoleDynamicsParams>;
static_assert(mppi::barrier_free_step<CartpoleDynamics>::value && mppi::barrier_free_step<CartpoleQuadraticCost>::value, "");
static_assert(mppi::barrier_free_step<DoubleIntegratorDynamics>::value && mppi::barrier_free_step<DoubleIntegratorCircleCost>::value, "");
static_assert(mppi::barrier_free_step<G>::value, "");
static_assert(!mppi::barrier_free_step<RefPendulumDynamics>::value && !mppi::barrier_free_step<RefPendulumCost>::value,
              "a class that says nothing is taken to have barriers: the CRTP bases must not declare it for it");
using RG = mppi::sampling_distributions::GaussianDistribution<RefPendulumParams>;
static_assert(mppi_amd::templated::rolePipelineAllowed<CartpoleDynamics, CartpoleQuadraticCost, G>(), "");
static_assert(!mppi_amd::templated::rolePipelineAllowed<RefPendulumDynamics, RefPendulumCost, RG>(), "");
static_assert(!mppi_amd::templated::rolePipelineAllowed<CartpoleDynamics, RefPendulumCost, G>(), "one undeclared plugin is enough");
int main() { return 0; }
""")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(REPO, "include"),
                        "-I" + os.path.join(REPO, "examples"), probe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def _numpy_rollout_costs(v, x0, dt, p, c):
    """float64 Euler rollout of the pendulum + cost, same loop structure as the rollout kernel (cost of the state AFTER the
    step, averaged over T; reference: tests/include/kernel_tests/core/rollout_kernel_test.cu:505-541)"""
    K, T, _ = v.shape
    th = np.full(K, x0[0], np.float64)
    om = np.full(K, x0[1], np.float64)
    total = np.zeros(K)
    inertia = p.mass * p.length ** 2
    for t in range(T):
        u = v[:, t, 0].astype(np.float64)
        acc = (u - p.damping * om - p.mass * p.gravity * p.length * np.sin(th)) / inertia
        th, om = th + om * dt, om + acc * dt
        total += c.angle_coeff * (1.0 - np.cos(th - c.goal_angle)) + c.velocity_coeff * om ** 2
    return total / T


def _make(K, T, dt, lam, **kw):
    eng = m.VanillaMPPIController("user_pendulum", K, T, dt, lam, 0.0, 1, seed=SEED, **kw)
    p = PendulumParams(1.0, 1.0, 0.1, 9.81)
    c = PendulumCostParams((C.c_float * 1)(0.0), 1.0, 10.0, 0.1, 0.0, np.pi)
    eng.setDynamicsParams(p)
    eng.setCostParams(c)
    eng.setControlRanges([[-4.0, 4.0]])
    return eng, p, c


@pytest.mark.gpu
def test_user_model_runs_and_matches_float64_rollout(gpu, plugin):
    K, T, dt = 2048, 60, 0.02
    eng, p, c = _make(K, T, dt, 1.0, save_samples=True)
    eng.setSamplingParams([2.0], [0.0])
    x0 = np.array([0.3, 0.0], np.float32)
    eng.injectNoise(host_noise(1, K, T, 1))
    eng.uploadState(x0)
    eng.optimize(1)  # one optimisation iteration, no smoothing: u* is the plain weighted mean
    costs = eng.getSampledCostSeq()[0]
    v = eng.getSampledControls()[0]
    assert np.abs(v).max() <= 4.0 and np.all(v[0] == 0.0)
    want = _numpy_rollout_costs(v, x0, dt, p, c)
    np.testing.assert_allclose(costs, want, rtol=1e-4)  # the reference's own GPU-vs-CPU bar (rollout_kernel_tests.cu:200-261)
    w = np.exp(-(costs.astype(np.float64) - costs.min()) / 1.0)
    u_direct = (w[:, None, None] * v.astype(np.float64)).sum(0) / w.sum()
    assert np.abs(eng.getOptimalControlSeq()[0] - u_direct).max() <= 1e-5
    # both kernel structures give the same bits for the user's model too
    eng2, _, _ = _make(K, T, dt, 1.0, kernel_variant=1)
    eng2.setSamplingParams([2.0], [0.0])
    eng2.injectNoise(host_noise(1, K, T, 1))
    eng2.uploadState(x0)
    eng2.optimize(1)
    assert np.array_equal(eng2.getSampledCostSeq()[0], costs)


@pytest.mark.gpu
def test_user_model_swings_up_in_closed_loop(gpu, plugin):
    eng, _, _ = _make(4096, 100, 0.02, 0.1)
    eng.setControlRanges([[-7.0, 7.0]])  # below m g l = 9.81: the pendulum has to pump energy to get up
    eng.setSamplingParams([3.0], [0.0])
    x = np.array([0.0, 0.0], np.float32)
    for _ in range(600):
        eng.computeControl(x, 1)
        u = eng.getControlSeq()[0]
        x, _ = eng.modelStep(x, u)
        eng.slideControlSequence(1)
    err = np.arctan2(np.sin(x[0] - np.pi), np.cos(x[0] - np.pi))
    assert abs(err) < 0.4 and abs(x[1]) < 2.0, x


# ------------------------------------------------------------------ a model file as a MPPI-Generic user has it ----------------
@pytest.fixture(scope="module")
def plugin_reference_style(plugin):
    assert plugin.mppi_load_plugin(build_plugin(SRC_REF_STYLE, PLUGIN_REF_STYLE).encode()) == 0, plugin.mppi_last_error(None)
    return plugin


def test_reference_style_model_source_is_what_integration_md_says(plugin_reference_style):
    """INTEGRATION.md §1 as an executable statement: a model written with the reference's include paths, its own step() with
    __syncthreads(), threadIdx.y-strided loops and the platform's sinf / cosf builds ALONE against include/ — the only edit
    against a CUDA source is cudaStream_t -> hipStream_t"""
    # (the classes live in pendulum_reference_style.cuh since round 6: the templated example includes them too)
    src = open(os.path.join(os.path.dirname(SRC_REF_STYLE), "pendulum_reference_style.cuh")).read()
    code = src[src.index("#include"):]  # below the header comment
    assert "MPPI_BARRIER_FREE_STEP" not in code and "mppi_amd" not in code  # no engine-specific line in the model's classes
    assert "<mppi/dynamics/dynamics.cuh>" in code and "<mppi/cost_functions/cost.cuh>" in code
    assert "__syncthreads()" in code and "sinf(" in code and "cosf(" in code and "threadIdx.y" in code and "blockDim.y" in code
    assert "mppi::det::" not in code and "lane_sync" not in code and "cuda" not in code.replace("cudaStream_t stream = nullptr", "")
    assert "user_pendulum_reference_style" in plugin_reference_style.mppi_list_models().decode().split("\n")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 1), (16, 4)], ids=["64x1", "16x4-reference-shape"])
def test_reference_style_model_meets_the_references_cost_tolerance(gpu, plugin_reference_style, shape):
    """the unmodified device code against a float64 rollout at the reference's own GPU-vs-CPU bar (1e-4 relative,
    tests/mppi_core/rollout_kernel_tests.cu:200-261) — on one lane per rollout and on the reference's (x = rollout,
    y = intra-rollout lane) block shape, where the model's own __syncthreads() separate the phases of a step as upstream —
    and against the mppi::det:: version of the same model (what changes is the last bits, not the controller)"""
    K, T, dt = 2048, 60, 0.02
    x0 = np.array([0.3, 0.0], np.float32)
    eps = host_noise(1, K, T, 1)

    def run(model, **kw):
        eng = m.VanillaMPPIController(model, K, T, dt, 1.0, 0.0, 1, seed=SEED, save_samples=True, **kw)
        p = PendulumParams(1.0, 1.0, 0.1, 9.81)
        c = PendulumCostParams((C.c_float * 1)(0.0), 1.0, 10.0, 0.1, 0.0, np.pi)
        eng.setDynamicsParams(p)
        eng.setCostParams(c)
        eng.setControlRanges([[-4.0, 4.0]])
        eng.setSamplingParams([2.0], [0.0])
        eng.injectNoise(eps)
        eng.uploadState(x0)
        eng.optimize(1)
        out = eng.getSampledCostSeq()[0].copy(), eng.getSampledControls()[0].copy(), eng.getOptimalControlSeq()[0].copy()
        eng.close()
        return out + (p, c)

    costs, v, u, p, c = run("user_pendulum_reference_style", block_x=shape[0], block_y=shape[1])
    want = _numpy_rollout_costs(v, x0, dt, p, c)
    np.testing.assert_allclose(costs, want, rtol=1e-4)
    costs_det, v_det, u_det, _, _ = run("user_pendulum")
    assert np.array_equal(v, v_det)                      # same sampler, same clamp: same samples
    np.testing.assert_allclose(costs, costs_det, rtol=2e-5)
    assert np.abs(u - u_det).max() <= 1e-4               # lambda = 1: a few 1e-6 of cost move the weights by as much
