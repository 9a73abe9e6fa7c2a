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
 * tests/test_plugin_model.py builds this file alone and runs it against the float64 rollout and against pendulum_model.hip.
 */
#include <mppi/dynamics/dynamics.cuh>
#include <mppi/cost_functions/cost.cuh>
#include <mppi/sampling_distributions/gaussian/gaussian.cuh>
#include "mppi_amd/engine/model_registry.hpp"

struct RefPendulumParams : public DynamicsParams
{
  enum class StateIndex : int
  {
    THETA = 0,
    THETA_DOT,
    NUM_STATES
  };
  enum class ControlIndex : int
  {
    TORQUE = 0,
    NUM_CONTROLS
  };
  enum class OutputIndex : int
  {
    THETA = 0,
    THETA_DOT,
    NUM_OUTPUTS
  };
  float mass = 1.0f;
  float length = 1.0f;
  float damping = 0.1f;
  float gravity = 9.81f;
};

using namespace MPPI_internal;

class RefPendulumDynamics : public Dynamics<RefPendulumDynamics, RefPendulumParams>
{
public:
  using PARENT_CLASS = Dynamics<RefPendulumDynamics, RefPendulumParams>;
  RefPendulumDynamics(hipStream_t stream = nullptr) : PARENT_CLASS(stream)  // reference: cudaStream_t stream = nullptr
  {
  }

  __device__ void computeDynamics(float* state, float* control, float* state_der, float* theta_s = nullptr)
  {
    // reference style: the lanes of a rollout (threadIdx.y) split the state derivative between them
    const float inertia = this->params_.mass * this->params_.length * this->params_.length;
    for (int i = threadIdx.y; i < STATE_DIM; i += blockDim.y)
    {
      if (i == S_INDEX(THETA))
        state_der[i] = state[S_INDEX(THETA_DOT)];
      else
        state_der[i] = (control[C_INDEX(TORQUE)] - this->params_.damping * state[S_INDEX(THETA_DOT)] -
                        this->params_.mass * this->params_.gravity * this->params_.length * sinf(state[S_INDEX(THETA)])) /
                       inertia;
    }
  }

  /** the reference's Dynamics::step (dynamics/dynamics.cu:130-142), block barriers and all */
  __device__ void step(float* state, float* next_state, float* state_der, float* control, float* output, float* theta_s,
                       const float t, const float dt)
  {
    computeStateDeriv(state, control, state_der, theta_s);
    __syncthreads();
    updateState(state, next_state, state_der, dt);
    __syncthreads();
    stateToOutput(next_state, output);
  }
};

struct RefPendulumCostParams : public CostParams<1>
{
  float angle_coeff = 10.0f;
  float velocity_coeff = 0.1f;
  float terminal_coeff = 0.0f;
  float goal_angle = 3.14159265f;
};

class RefPendulumCost : public Cost<RefPendulumCost, RefPendulumCostParams, RefPendulumParams>
{
public:
  RefPendulumCost(hipStream_t stream = nullptr)
  {
    bindToStream(stream);
  }
  __device__ float computeStateCost(float* y, int timestep = 0, float* theta_c = nullptr, int* crash_status = nullptr)
  {
    return params_.angle_coeff * (1.0f - cosf(y[O_IND_CLASS(RefPendulumParams, THETA)] - params_.goal_angle)) +
           params_.velocity_coeff * SQ(y[O_IND_CLASS(RefPendulumParams, THETA_DOT)]);
  }
  __device__ float terminalCost(float* y, float* theta_c)
  {
    return computeStateCost(y) * params_.terminal_coeff;
  }
};

using namespace mppi::engine;
using RefPendulumSampler = mppi::sampling_distributions::GaussianDistribution<RefPendulumParams>;
/* (64, 1, 1): one lane per rollout; (16, 4, 1): the reference's x = rollout / y = intra-rollout lane shape, state in LDS and the
 * model's own barriers doing what they do upstream.  PIPELINE = false: see the header comment. */
using RefPendulumModel = ModelT<RefPendulumDynamics, RefPendulumCost, RefPendulumSampler, Shapes<Shape<64, 1, 1>, Shape<16, 4, 1>>,
                                /*FIN_BY=*/1, void, Shapes<>, /*PIPELINE=*/false>;
MPPI_REGISTER_MODEL("user_pendulum_reference_style", MPPI_SAMPLER_GAUSSIAN, RefPendulumModel, 64, 1)
