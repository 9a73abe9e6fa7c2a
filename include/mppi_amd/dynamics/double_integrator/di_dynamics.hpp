/**
 * DoubleIntegratorDynamics plugin (reference: include/mppi/dynamics/double_integrator/di_dynamics.cuh:9-75,
 * di_dynamics.cu:46-53).
 */
#ifndef MPPI_AMD_DI_DYNAMICS_HPP_
#define MPPI_AMD_DI_DYNAMICS_HPP_

#include "mppi_amd/plugin/dynamics.hpp"

struct DoubleIntegratorParams : public DynamicsParams
{
  enum class StateIndex : int
  {
    POS_X = 0,
    POS_Y,
    VEL_X,
    VEL_Y,
    NUM_STATES
  };
  enum class ControlIndex : int
  {
    ACCEL_X = 0,
    ACCEL_Y,
    NUM_CONTROLS
  };
  enum class OutputIndex : int
  {
    POS_X = 0,
    POS_Y,
    VEL_X,
    VEL_Y,
    NUM_OUTPUTS
  };
  float system_noise = 1;
  DoubleIntegratorParams(float noise) : system_noise(noise){};
  DoubleIntegratorParams() = default;
};

using namespace MPPI_internal;

class DoubleIntegratorDynamics : public Dynamics<DoubleIntegratorDynamics, DoubleIntegratorParams>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  using PARENT_CLASS = Dynamics<DoubleIntegratorDynamics, DoubleIntegratorParams>;
  DoubleIntegratorDynamics(float system_noise = 1, hipStream_t stream = nullptr) : PARENT_CLASS(stream)
  {
    this->params_ = DoubleIntegratorParams(system_noise);
  }
  static const char* getDynamicsModelName()
  {
    return "2D Double Integrator Model";
  }

  /** d/dt [p_x, p_y, v_x, v_y] = [v_x, v_y, a_x, a_y] with the control as the acceleration (di_dynamics.cu:46-53) */
  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta = nullptr)
  {
    constexpr int HALF = 2;  // positions, then velocities
#pragma unroll
    for (int axis = 0; axis < HALF; axis++)
    {
      state_der[axis] = state[HALF + axis];
      state_der[HALF + axis] = control[axis];
    }
  }
};

#endif
