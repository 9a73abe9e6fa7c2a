/**
 * CartpoleDynamics plugin (reference: include/mppi/dynamics/cartpole/cartpole_dynamics.cuh:6-105,
 * cartpole_dynamics.cu:89-107 — the device computeDynamics, expression for expression).
 * The reference's __sinf/__cosf become the bit-reproducible det::sincos (see det_math.h for why).
 */
#ifndef MPPI_AMD_CARTPOLE_DYNAMICS_HPP_
#define MPPI_AMD_CARTPOLE_DYNAMICS_HPP_

#include "mppi_amd/plugin/dynamics.hpp"

struct CartpoleDynamicsParams : public DynamicsParams
{
  enum class StateIndex : int
  {
    POS_X = 0,
    VEL_X,
    THETA,
    THETA_DOT,
    NUM_STATES
  };
  enum class ControlIndex : int
  {
    FORCE = 0,
    NUM_CONTROLS
  };
  enum class OutputIndex : int
  {
    POS_X = 0,
    VEL_X,
    THETA,
    THETA_DOT,
    NUM_OUTPUTS
  };
  float cart_mass = 1.0f;
  float pole_mass = 1.0f;
  float pole_length = 1.0f;

  CartpoleDynamicsParams() = default;
  CartpoleDynamicsParams(float cart_mass, float pole_mass, float pole_length)
    : cart_mass(cart_mass), pole_mass(pole_mass), pole_length(pole_length){};
};

using namespace MPPI_internal;

class CartpoleDynamics : public Dynamics<CartpoleDynamics, CartpoleDynamicsParams>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  using PARENT_CLASS = Dynamics<CartpoleDynamics, CartpoleDynamicsParams>;
  CartpoleDynamics(float cart_mass = 1.0f, float pole_mass = 1.0f, float pole_length = 1.0f, hipStream_t stream = 0)
    : PARENT_CLASS(stream)
  {
    this->params_ = CartpoleDynamicsParams(cart_mass, pole_mass, pole_length);
  }

  static const char* getDynamicsModelName()
  {
    return "Cartpole Model";
  }
  __host__ __device__ float getCartMass()
  {
    return this->params_.cart_mass;
  };
  __host__ __device__ float getPoleMass()
  {
    return this->params_.pole_mass;
  };
  __host__ __device__ float getPoleLength()
  {
    return this->params_.pole_length;
  };
  __host__ __device__ float getGravity()
  {
    return gravity_;
  }

  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta = nullptr)
  {
    // reference: angle_utils::normalizeAngle(state[2]) (cartpole_dynamics.cu:91); the bounded variant returns the same
    // value for |theta| < 1e7 rad and keeps the range test off the rollout's dependent chain
    float theta_n = angle_utils::normalizeAngleBounded(state[2]);
    float sin_theta, cos_theta;
    mppi::det::sincos(theta_n, &sin_theta, &cos_theta);
    float theta_dot = state[3];
    float force = control[0];
    float m_c = this->params_.cart_mass;
    float m_p = this->params_.pole_mass;
    float l_p = this->params_.pole_length;

    state_der[0] = state[1];
    // 1.0f / x with x = m_c + m_p sin^2 and x = l_p (m_c + m_p sin^2): positive, ordinary magnitudes -> det::rcp_benign2
    // returns the correctly rounded IEEE reciprocals (what the reference's `1.0f / (...)` and the CPU oracle compute),
    // both in 8 instructions
    const float den = m_c + m_p * SQ(sin_theta);
    float r_x, r_th;
    mppi::det::rcp_benign2(den, l_p * den, &r_x, &r_th);
    state_der[1] = r_x * (force + m_p * sin_theta * (l_p * SQ(theta_dot) + gravity_ * cos_theta));
    state_der[2] = theta_dot;
    state_der[3] =
        r_th *
        (-force * cos_theta - m_p * l_p * SQ(theta_dot) * cos_theta * sin_theta - (m_c + m_p) * gravity_ * sin_theta);
  }

protected:
  const float gravity_ = 9.81;
};

#endif
