/**
 * oracle_models.hpp — CPU restatement of the concrete Dynamics / Cost plugins on the hot path.  TEST INFRASTRUCTURE ONLY.
 * Device flavour of every formula (see oracle_core.hpp); paths relative to the reference's include/mppi/.
 */
#ifndef MPPI_ORACLE_MODELS_HPP_
#define MPPI_ORACLE_MODELS_HPP_

#include "oracle_core.hpp"
#include "mppi_amd/model_params.h"

#define ORACLE_SQ(a) ((a) * (a)) /* reference: utils/math_utils.h SQ() */

namespace oracle
{
/* ------------------------------------------------------------------ Cartpole -------------------------------------- */
/** reference: dynamics/cartpole/cartpole_dynamics.cu:89-107 (device computeDynamics), cartpole_dynamics.cuh:101 gravity */
struct CartpoleDynamics : Dynamics
{
  mppi_cartpole_dynamics_params p{ 1.0f, 1.0f, 1.0f };
  const float gravity_ = 9.81;
  CartpoleDynamics() : Dynamics(4, 1, 4)
  {
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(p))
      return -1;
    memcpy(&p, pod, n);
    return 0;
  }
  void computeDynamics(const float* state, const float* control, float* state_der, float* theta_s) override
  {
    float theta = det::normalizeAngle(state[2]);
    float sin_theta, cos_theta;
    det::sincos(theta, &sin_theta, &cos_theta); /* reference: __sinf/__cosf */
    float theta_dot = state[3];
    float force = control[0];
    float m_c = p.cart_mass;
    float m_p = p.pole_mass;
    float l_p = p.pole_length;

    state_der[0] = state[1];
    state_der[1] = 1.0f / (m_c + m_p * ORACLE_SQ(sin_theta)) *
                   (force + m_p * sin_theta * (l_p * ORACLE_SQ(theta_dot) + gravity_ * cos_theta));
    state_der[2] = theta_dot;
    state_der[3] = 1.0f / (l_p * (m_c + m_p * ORACLE_SQ(sin_theta))) *
                   (-force * cos_theta - m_p * l_p * ORACLE_SQ(theta_dot) * cos_theta * sin_theta -
                    (m_c + m_p) * gravity_ * sin_theta);
  }
};

/** reference: cost_functions/cartpole/cartpole_quadratic_cost.cu:20-43 */
struct CartpoleQuadraticCost : Cost
{
  mppi_cartpole_cost_params params_{ { 10.0f }, 1.0f, 1000.0f, 100.0f, 2000.0f, 100.0f, 0.0f, { 0.0f, 0.0f, (float)M_PI, 0.0f } };
  CartpoleQuadraticCost() : Cost(1, 4)
  {
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(params_))
      return -1;
    memcpy(&params_, pod, n);
    return 0;
  }
  float quad(const float* state) const
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
  float computeStateCost(const float* s, int t, int* crash) override
  {
    return quad(s);
  }
  float terminalCost(const float* s) override
  {
    return quad(s) * params_.terminal_cost_coeff;
  }
};

/* ------------------------------------------------------------------ Double integrator ------------------------------ */
/** reference: dynamics/double_integrator/di_dynamics.cu:46-53 */
struct DoubleIntegratorDynamics : Dynamics
{
  mppi_di_dynamics_params p{ 1.0f };
  DoubleIntegratorDynamics() : Dynamics(4, 2, 4)
  {
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(p))
      return -1;
    memcpy(&p, pod, n);
    return 0;
  }
  void computeDynamics(const float* state, const float* control, float* state_der, float* theta_s) override
  {
    state_der[0] = state[2];
    state_der[1] = state[3];
    state_der[2] = control[0];
    state_der[3] = control[1];
  }
};

/** reference: cost_functions/double_integrator/double_integrator_circle_cost.cu:8-32 */
struct DoubleIntegratorCircleCost : Cost
{
  mppi_di_circle_cost_params params_{ { 0.01f, 0.01f }, 1.0f, 1.0f, 1000.0f, 2.0f, 1.875f * 1.875f, 2.125f * 2.125f, 4.0f };
  DoubleIntegratorCircleCost() : Cost(2, 4)
  {
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(params_))
      return -1;
    memcpy(&params_, pod, n);
    return 0;
  }
  /** powf(discount, timestep): exact 1 for the default discount == 1; otherwise det::pow_pos (shared with the engine) */
  static float discountPow(float discount, int timestep)
  {
    return discount == 1.0f ? 1.0f : det::pow_pos(discount, (float)timestep);
  }
  float computeStateCost(const float* s, int timestep, int* crash) override
  {
    float radial_position = s[0] * s[0] + s[1] * s[1];
    float current_velocity = det::sqrt(s[2] * s[2] + s[3] * s[3]);
    float current_angular_momentum = s[0] * s[3] - s[1] * s[2];
    float cost = 0;
    if ((radial_position < params_.inner_path_radius2) || (radial_position > params_.outer_path_radius2))
    {
      cost += discountPow(params_.discount, timestep) * params_.crash_cost;
    }
    cost += params_.velocity_cost * fabsf(current_velocity - params_.velocity_desired);
    cost += params_.velocity_cost * fabsf(current_angular_momentum - params_.angular_momentum_desired);
    return cost;
  }
  float terminalCost(const float* s) override
  {
    return 0;
  }
};

inline bool makeModel(const std::string& name, std::unique_ptr<Dynamics>& dyn, std::unique_ptr<Cost>& cost)
{
  if (name == "cartpole")
  {
    dyn.reset(new CartpoleDynamics());
    cost.reset(new CartpoleQuadraticCost());
    return true;
  }
  if (name == "double_integrator")
  {
    dyn.reset(new DoubleIntegratorDynamics());
    cost.reset(new DoubleIntegratorCircleCost());
    return true;
  }
  return false;
}

}  // namespace oracle

#endif  // MPPI_ORACLE_MODELS_HPP_
