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
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  /** a pure parameter block on the device: role loops may run it straight off the kernel's argument block (engine/kernarg_view.hpp) */
  static constexpr bool MPPI_KERNARG_VIEWABLE = true;
  DoubleIntegratorCircleCost(hipStream_t stream = nullptr)
  {
    bindToStream(stream);
  }

  /**
   * Stay on the annulus inner_path_radius2 <= |p|^2 <= outer_path_radius2 (else discount^t * crash_cost), hold the
   * desired speed and the desired (counter-clockwise) angular momentum p x v — the three terms are added in the
   * reference's order (double_integrator_circle_cost.cu:8-32).
   */
  __device__ inline float computeStateCost(float* s, int timestep = 0, float* theta_c = nullptr,
                                           int* crash_status = nullptr)
  {
    const float px = s[0], py = s[1], vx = s[2], vy = s[3];
    const float r2 = px * px + py * py;
    const float speed = mppi::det::sqrt(vx * vx + vy * vy);
    const float ang_mom = px * vy - py * vx;
    const bool off_track = (r2 < params_.inner_path_radius2) || (r2 > params_.outer_path_radius2);

    float cost = 0;
    if (off_track)
    {
      const float gamma_t = params_.discount == 1.0f ? 1.0f : mppi::det::pow_pos(params_.discount, (float)timestep);
      cost += gamma_t * params_.crash_cost;
    }
    cost += params_.velocity_cost * fabsf(speed - params_.velocity_desired);
    cost += params_.velocity_cost * fabsf(ang_mom - params_.angular_momentum_desired);
    return cost;
  }

  __device__ inline float terminalCost(float* state, float* theta_c)
  {
    return 0;
  }
};

#endif
