/**
 * The pendulum of pendulum_model.hip written the way a MPPI-Generic user's model file looks BEFORE any adaptation to this
 * engine — executable documentation of INTEGRATION.md §1 ("what a maintainer changes in a model file"):
 *
 *   kept as it is    the reference's include paths (<mppi/dynamics/dynamics.cuh>, <mppi/cost_functions/cost.cuh>), the CRTP
 *                    bases and the params structs with their index enums, a step() written like the reference's own
 *                    (dynamics/dynamics.cu:130-142: computeStateDeriv / __syncthreads() / updateState / __syncthreads() /
 *                    stateToOutput), loops strided over threadIdx.y / blockDim.y, the platform's sinf / cosf (ocml here, the
 *                    CUDA math library there) instead of mppi::det::
 *   changed          cudaStream_t -> hipStream_t (one token per constructor; there is no CUDA shim in this repository), and the
 *                    Eigen host overloads (computeDynamics(const Eigen::Ref<...>&...), stateFromMap) are simply not written:
 *                    nothing on this path evaluates a model on the host
 *   consequence of   a model that calls __syncthreads() itself (instead of mppi::lane_sync(), which folds away when a rollout
 *   keeping the      owns one lane) must run on the FUSED rollout kernel, where every thread of the block reaches every call as
 *   barriers         in the reference: PIPELINE = false below.  With mppi::lane_sync() the same source also runs role-pipelined.
 *   consequence of   trajectory costs agree with a float64 restatement to the reference's own GPU-vs-CPU bar (1e-4 relative,
 *   keeping sinf     tests/mppi_core/rollout_kernel_tests.cu:200-261) but are no longer bit-identical to a CPU oracle: that is what
 *                    mppi::det:: buys (include/mppi_amd/det_math.h)
 *
 * The classes live in pendulum_reference_style.cuh (shared with the templated example).
 * tests/test_plugin_model.py builds this file alone and runs it against the float64 rollout and against pendulum_model.hip.
 */
#include "pendulum_reference_style.cuh"
#include "mppi_amd/engine/model_registry.hpp"

using namespace mppi::engine;
using RefPendulumSampler = mppi::sampling_distributions::GaussianDistribution<RefPendulumParams>;
/* (64, 1, 1): one lane per rollout; (16, 4, 1): the reference's x = rollout / y = intra-rollout lane shape, state in LDS and the
 * model's own barriers doing what they do upstream.  PIPELINE = false: see the header comment. */
using RefPendulumModel = ModelT<RefPendulumDynamics, RefPendulumCost, RefPendulumSampler, Shapes<Shape<64, 1, 1>, Shape<16, 4, 1>>,
                                /*FIN_BY=*/1, void, Shapes<>, /*PIPELINE=*/false>;
MPPI_REGISTER_MODEL("user_pendulum_reference_style", MPPI_SAMPLER_GAUSSIAN, RefPendulumModel, 64, 1)
