/**
 * DoubleIntegratorCircleCost plugin (reference: include/mppi/cost_functions/double_integrator/
 * double_integrator_circle_cost.cuh:8-39, double_integrator_circle_cost.cu:8-32 and :65-68).
 * powf(discount, t) is exact 1 for the default discount == 1 and goes through det::pow_pos otherwise.
 */
#ifndef MPPI_AMD_DI_CIRCLE_COST_HPP_
#define MPPI_AMD_DI_CIRCLE_COST_HPP_

#include "mppi_amd/plugin/cost.hpp"
#include "mppi_amd/dynamics/double_integrator/di_dynamics.hpp"

struct DoubleIntegratorCircleCostParams : public CostParams<2>
{
  float velocity_cost = 1;
  float crash_cost = 1000;
  float velocity_desired = 2;
  float inner_path_radius2 = 1.875 * 1.875;
  float outer_path_radius2 = 2.125 * 2.125;
  float angular_momentum_desired = 2 * velocity_desired;  // Enforces the system travels counter clockwise

  DoubleIntegratorCircleCostParams()
  {
    control_cost_coeff[0] = 0.01;
    control_cost_coeff[1] = 0.01;
    discount = 1.0;
  }
};

class DoubleIntegratorCircleCost
  : public Cost<DoubleIntegratorCircleCost, DoubleIntegratorCircleCostParams, DoubleIntegratorParams>
{
public:
  DoubleIntegratorCircleCost(hipStream_t stream = nullptr)
  {
    bindToStream(stream);
  }

  __device__ inline float computeStateCost(float* s, int timestep = 0, float* theta_c = nullptr,
                                           int* crash_status = nullptr)
  {
    float radial_position = s[0] * s[0] + s[1] * s[1];
    float current_velocity = mppi::det::sqrt(s[2] * s[2] + s[3] * s[3]);
    float current_angular_momentum = s[0] * s[3] - s[1] * s[2];

    float cost = 0;
    if ((radial_position < params_.inner_path_radius2) || (radial_position > params_.outer_path_radius2))
    {
      const float disc =
          this->params_.discount == 1.0f ? 1.0f : mppi::det::pow_pos(this->params_.discount, (float)timestep);
      cost += disc * params_.crash_cost;
    }

    cost += params_.velocity_cost * fabsf(current_velocity - params_.velocity_desired);
    cost += params_.velocity_cost * fabsf(current_angular_momentum - params_.angular_momentum_desired);
    return cost;
  }

  __device__ inline float terminalCost(float* state, float* theta_c)
  {
    return 0;
  }
};

#endif
