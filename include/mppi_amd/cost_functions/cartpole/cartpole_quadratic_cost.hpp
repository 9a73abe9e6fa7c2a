/**
 * CartpoleQuadraticCost plugin (reference: include/mppi/cost_functions/cartpole/cartpole_quadratic_cost.cuh:10-52,
 * cartpole_quadratic_cost.cu:20-43).
 */
#ifndef MPPI_AMD_CARTPOLE_QUADRATIC_COST_HPP_
#define MPPI_AMD_CARTPOLE_QUADRATIC_COST_HPP_

#include "mppi_amd/plugin/cost.hpp"
#include "mppi_amd/dynamics/cartpole/cartpole_dynamics.hpp"

struct CartpoleQuadraticCostParams : public CostParams<1>
{
  float cart_position_coeff = 1000;
  float cart_velocity_coeff = 100;
  float pole_angle_coeff = 2000;
  float pole_angular_velocity_coeff = 100;
  float terminal_cost_coeff = 0;
  float desired_terminal_state[4] = { 0, 0, (float)M_PI, 0 };

  CartpoleQuadraticCostParams()
  {
    this->control_cost_coeff[0] = 10.0;
  }
};

class CartpoleQuadraticCost : public Cost<CartpoleQuadraticCost, CartpoleQuadraticCostParams, CartpoleDynamicsParams>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  /** a pure parameter block on the device: role loops may run it straight off the kernel's argument block (engine/kernarg_view.hpp) */
  static constexpr bool MPPI_KERNARG_VIEWABLE = true;
  CartpoleQuadraticCost(hipStream_t stream = 0)
  {
    bindToStream(stream);
  }

  /**
   * sum_i coeff_i (x_i - goal_i)^2 over [cart position, cart velocity, pole angle, pole angular velocity], each term
   * evaluated as (d * d) * coeff and added left to right — the rounding sequence of the reference's expression
   * (cartpole_quadratic_cost.cu:20-29)
   */
  __device__ inline float weightedSquaredError(const float* x) const
  {
    const float w[4] = { params_.cart_position_coeff, params_.cart_velocity_coeff, params_.pole_angle_coeff,
                         params_.pole_angular_velocity_coeff };
    float d = x[0] - params_.desired_terminal_state[0];
    float total = d * d * w[0];
#pragma unroll
    for (int i = 1; i < 4; i++)
    {
      d = x[i] - params_.desired_terminal_state[i];
      total = total + d * d * w[i];
    }
    return total;
  }

  __device__ inline float computeStateCost(float* state, int timestep = 0, float* theta_c = nullptr,
                                           int* crash_status = nullptr)
  {
    return weightedSquaredError(state);
  }

  /** cartpole_quadratic_cost.cu:31-43: the same quadratic, scaled by terminal_cost_coeff */
  __device__ inline float terminalCost(float* state, float* theta_c)
  {
    return weightedSquaredError(state) * params_.terminal_cost_coeff;
  }
};

#endif
