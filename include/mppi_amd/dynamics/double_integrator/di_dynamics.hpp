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
  using PARENT_CLASS = Dynamics<DoubleIntegratorDynamics, DoubleIntegratorParams>;
  DoubleIntegratorDynamics(float system_noise = 1, hipStream_t stream = nullptr) : PARENT_CLASS(stream)
  {
    this->params_ = DoubleIntegratorParams(system_noise);
  }
  static const char* getDynamicsModelName()
  {
    return "2D Double Integrator Model";
  }

  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta = nullptr)
  {
    state_der[0] = state[2];    // xdot;
    state_der[1] = state[3];    // ydot;
    state_der[2] = control[0];  // x_force;
    state_der[3] = control[1];  // y_force
  }
};

#endif
