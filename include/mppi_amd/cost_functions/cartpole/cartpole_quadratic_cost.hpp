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
  CartpoleQuadraticCost(hipStream_t stream = 0)
  {
    bindToStream(stream);
  }

  __device__ inline float computeStateCost(float* state, int timestep = 0, float* theta_c = nullptr,
                                           int* crash_status = nullptr)
  {
    return (state[0] - params_.desired_terminal_state[0]) * (state[0] - params_.desired_terminal_state[0]) *
               params_.cart_position_coeff +
           (state[1] - params_.desired_terminal_state[1]) * (state[1] - params_.desired_terminal_state[1]) *
               params_.cart_velocity_coeff +
           (state[2] - params_.desired_terminal_state[2]) * (state[2] - params_.desired_terminal_state[2]) *
               params_.pole_angle_coeff +
           (state[3] - params_.desired_terminal_state[3]) * (state[3] - params_.desired_terminal_state[3]) *
               params_.pole_angular_velocity_coeff;
  }

  __device__ inline float terminalCost(float* state, float* theta_c)
  {
    return ((state[0] - params_.desired_terminal_state[0]) * (state[0] - params_.desired_terminal_state[0]) *
                params_.cart_position_coeff +
            (state[1] - params_.desired_terminal_state[1]) * (state[1] - params_.desired_terminal_state[1]) *
                params_.cart_velocity_coeff +
            (state[2] - params_.desired_terminal_state[2]) * (state[2] - params_.desired_terminal_state[2]) *
                params_.pole_angle_coeff +
            (state[3] - params_.desired_terminal_state[3]) * (state[3] - params_.desired_terminal_state[3]) *
                params_.pole_angular_velocity_coeff) *
           params_.terminal_cost_coeff;
  }
};

#endif
