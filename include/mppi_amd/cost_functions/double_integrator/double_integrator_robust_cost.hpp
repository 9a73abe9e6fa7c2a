/**
 * DoubleIntegratorRobustCost plugin (reference: include/mppi/cost_functions/double_integrator/
 * double_integrator_robust_cost.cuh:6-26, double_integrator_robust_cost.cu:10-41 — the DEVICE overload).
 *
 * Three of the six controller set-ups of the reference's examples/double_integrator_CORL2020.cu run on this cost
 * (:250-261 Vanilla, :428-438 Tube, :630-645 Robust MPPI, each with crash_cost = 100).
 *
 * Quirk decided: the reference's host and device overloads use DIFFERENT constants — the device
 * one (.cu:18-19) ramps to 0.5 * crash_cost at half the track width, the host one (.cu:55-56) to 0.1 * crash_cost at
 * three quarters.  What the controllers optimise is the device overload (every rollout kernel calls it), so this
 * plugin — evaluated by the rollout kernels AND by the host-side paths of this library — is the device flavour.
 * powf(x, 2) is x * x (exact in any libm for the integer exponent 2, and what nvcc emits for the literal).
 */
#ifndef MPPI_AMD_DI_ROBUST_COST_HPP_
#define MPPI_AMD_DI_ROBUST_COST_HPP_

#include "mppi_amd/cost_functions/double_integrator/double_integrator_circle_cost.hpp"
#include "mppi_amd/plugin/math_utils.hpp"

class DoubleIntegratorRobustCost
  : public Cost<DoubleIntegratorRobustCost, DoubleIntegratorCircleCostParams, DoubleIntegratorParams>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  /** a pure parameter block on the device: role loops may run it straight off the kernel's argument block (engine/kernarg_view.hpp) */
  static constexpr bool MPPI_KERNARG_VIEWABLE = true;
  DoubleIntegratorRobustCost(hipStream_t stream = nullptr)
  {
    bindToStream(stream);
  }

  /** the bound RMPPI's value-function threshold is chosen against (reference: .cuh:20-23) */
  float getLipshitzConstantCost()
  {
    return params_.crash_cost;
  }

  /**
   * A track cost that is continuous up to the edge: 0 on the centre line, half the crash cost half way to the
   * boundary, the crash cost from the boundary on; plus the SQUARED speed and angular-momentum errors
   * (double_integrator_robust_cost.cu:10-41).
   */
  __device__ inline float computeStateCost(float* s, int timestep = 0, float* theta_c = nullptr,
                                           int* crash_status = nullptr)
  {
    const float px = s[0], py = s[1], vx = s[2], vy = s[3];
    const float r2 = px * px + py * py;
    const float speed = mppi::det::sqrt(vx * vx + vy * vy);
    const float ang_mom = px * vy - py * vx;
    const float nd = mppi::math::normDistFromCenter(mppi::det::sqrt(r2), mppi::det::sqrt(params_.inner_path_radius2),
                                                    mppi::det::sqrt(params_.outer_path_radius2));
    const float knee = 0.5f;
    const float knee_cost = 0.5f * params_.crash_cost;

    float cost = 0;
    if (nd <= knee)
    {
      cost += mppi::math::linInterp(nd, 0, knee, 0, knee_cost);
    }
    if (nd > knee && nd <= 1.0f)
    {
      cost += mppi::math::linInterp(nd, knee, 1, knee_cost, params_.crash_cost);
    }
    if (nd > 1.0f)
    {
      cost += params_.crash_cost;
    }
    const float dv = speed - params_.velocity_desired;
    const float dl = ang_mom - params_.angular_momentum_desired;
    cost += params_.velocity_cost * (dv * dv);
    cost += params_.velocity_cost * (dl * dl);
    return cost;
  }

  __device__ inline float terminalCost(float* state, float* theta_c)
  {
    return 0;
  }
};

#endif
