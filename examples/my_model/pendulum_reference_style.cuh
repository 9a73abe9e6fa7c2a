/**
 * The classes of pendulum_model_reference_style.hip (see the notes there): a pendulum model and cost written the way a
 * MPPI-Generic user's files look BEFORE any adaptation — reference include paths, CRTP bases, a step() with the reference's two
 * __syncthreads() (dynamics/dynamics.cu:130-142), loops strided over threadIdx.y, ocml sinf / cosf.  Neither class declares
 * MPPI_BARRIER_FREE_STEP, so the engine takes them to have block barriers (they do) and keeps them on the fused kernel.
 * Included by: pendulum_model_reference_style.hip (name-keyed registration), ../templated_pendulum_reference_style.hip (the
 * reference's templated controller classes), and tests/probes/pendulum_forced_pipeline.hip (the refusal test).
 */
#ifndef EXAMPLES_MY_MODEL_PENDULUM_REFERENCE_STYLE_CUH_
#define EXAMPLES_MY_MODEL_PENDULUM_REFERENCE_STYLE_CUH_

#include <mppi/dynamics/dynamics.cuh>
#include <mppi/cost_functions/cost.cuh>
#include <mppi/sampling_distributions/gaussian/gaussian.cuh>

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

#endif
