/**
 * QuadraticCost plugin — sum_i coeff_i (y_i - goal_i)^2 over the OUTPUT vector of any Dynamics
 * (reference: include/mppi/cost_functions/quadratic_cost/quadratic_cost.cuh:11-128, quadratic_cost.cu:39-60 device
 * computeStateCost / terminalCost).  One goal (the reference's SIM_TIME_HORIZON = 1 instantiation: getIndex() clamps
 * every timestep to the single stored goal).  powf(x, 2) of the reference is x * x (what the CUDA compiler emits for a
 * constant exponent 2).
 *
 * SKIP_ZERO_COEFF = true: an output with coefficient 0 contributes exactly 0 even when it is not finite.  The elevation-map
 * RACER models mark outputs they do not model with NaN (racer_dubins_elevation.cu:131-139), which the reference's sum
 * turns into a NaN cost for every rollout (NaN * 0); this form is what lets QuadraticCost weigh the outputs those models
 * DO produce.  For finite outputs both forms give the same value (x * 0 = 0).
 */
#ifndef MPPI_AMD_QUADRATIC_COST_HPP_
#define MPPI_AMD_QUADRATIC_COST_HPP_

#include "mppi_amd/plugin/cost.hpp"

template <class DYN_T>
struct QuadraticCostParams : public CostParams<DYN_T::CONTROL_DIM>
{
  float s_goal[DYN_T::OUTPUT_DIM] = { 0 };
  float s_coeffs[DYN_T::OUTPUT_DIM] = { 0 };
  int current_time = 0;

  QuadraticCostParams()
  {
    for (int i = 0; i < DYN_T::CONTROL_DIM; i++)
    {
      this->control_cost_coeff[i] = 0;
    }
    for (int i = 0; i < DYN_T::OUTPUT_DIM; i++)
    {
      this->s_coeffs[i] = 1;
    }
  }
};

template <class DYN_T, bool SKIP_ZERO_COEFF = false>
class QuadraticCost
  : public Cost<QuadraticCost<DYN_T, SKIP_ZERO_COEFF>, QuadraticCostParams<DYN_T>, typename DYN_T::DYN_PARAMS_T>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  /** a pure parameter block on the device: role loops may run it straight off the kernel's argument block (engine/kernarg_view.hpp) */
  static constexpr bool MPPI_KERNARG_VIEWABLE = true;
  static constexpr float MAX_COST_VALUE = 1e16;
  QuadraticCost(hipStream_t stream = nullptr)
  {
    this->bindToStream(stream);
  }

  __device__ inline float computeStateCost(float* s, int timestep = 0, float* theta_c = nullptr,
                                           int* crash_status = nullptr)
  {
    float cost = 0;
    const float* desired_state = this->params_.s_goal;
#pragma unroll
    for (int i = 0; i < DYN_T::OUTPUT_DIM; i++)
    {
      const float term = ((s[i] - desired_state[i]) * (s[i] - desired_state[i])) * this->params_.s_coeffs[i];
      cost += (SKIP_ZERO_COEFF && this->params_.s_coeffs[i] == 0.0f) ? 0.0f : term;
    }
    return cost;
  }

  __device__ inline float terminalCost(float* s, float* theta_c)
  {
    return 0.0f;
  }
};

#endif
