/**
 * RacerDubins plugin — the RACER kinematic "Dubins car" with first-order throttle / brake / steering lags
 * (reference: include/mppi/dynamics/racer_dubins/racer_dubins.cuh:15-155, racer_dubins.cu:138-165 device
 * computeDynamics, :73-98 device updateState; expression for expression).
 *
 * State  [VEL_X, YAW, POS_X, POS_Y, STEER_ANGLE, BRAKE_STATE, STEER_ANGLE_RATE], control [THROTTLE_BRAKE, STEER_CMD],
 * 28 outputs (racer_dubins.cuh:37-66).  The plain RacerDubins uses the base class's step(): stateToOutput copies the
 * seven states into outputs 0..6 and the other outputs keep their initial zeros (setOutputs is only called by the
 * elevation-map subclasses, which are out of this round's scope together with the texture helper).
 * Reference intrinsics: __sincosf -> det::sincos, tan -> det::tan, normalizeAngle -> det::normalizeAngle (det_math.h).
 * The angle wrap of updateState is part of the plugin's own updateState, as in the reference.
 */
#ifndef MPPI_AMD_RACER_DUBINS_HPP_
#define MPPI_AMD_RACER_DUBINS_HPP_

#include "mppi_amd/plugin/dynamics.hpp"

struct RacerDubinsParams : public DynamicsParams
{
  enum class StateIndex : int
  {
    VEL_X = 0,
    YAW,
    POS_X,
    POS_Y,
    STEER_ANGLE,
    BRAKE_STATE,
    STEER_ANGLE_RATE,
    NUM_STATES
  };
  enum class ControlIndex : int
  {
    THROTTLE_BRAKE = 0,
    STEER_CMD,
    NUM_CONTROLS
  };
  enum class OutputIndex : int
  {
    BASELINK_VEL_B_X = 0,
    BASELINK_VEL_B_Y,
    BASELINK_POS_I_X,
    BASELINK_POS_I_Y,
    BASELINK_POS_I_Z,
    YAW,
    ROLL,
    PITCH,
    STEER_ANGLE,
    STEER_ANGLE_RATE,
    WHEEL_FORCE_UP_MAX,
    WHEEL_FORCE_FWD_MAX,
    WHEEL_FORCE_SIDE_MAX,
    ACCEL_X,
    ACCEL_Y,
    OMEGA_Z,
    TOTAL_VELOCITY,
    UNCERTAINTY_POS_X,
    UNCERTAINTY_POS_Y,
    UNCERTAINTY_YAW,
    UNCERTAINTY_VEL_X,
    UNCERTAINTY_POS_X_Y,
    UNCERTAINTY_POS_X_YAW,
    UNCERTAINTY_POS_X_VEL_X,
    UNCERTAINTY_POS_Y_YAW,
    UNCERTAINTY_POS_Y_VEL_X,
    UNCERTAINTY_YAW_VEL_X,
    FILLER_1,
    NUM_OUTPUTS
  };
  float c_t[3] = { 1.3f, 2.6f, 3.9f };
  float c_b[3] = { 2.5f, 3.5f, 4.5f };
  float c_v[3] = { 3.7f, 4.7f, 5.7f };
  float c_0 = 4.9f;
  float steering_constant = .6f;
  float steer_command_angle_scale = 5;
  float steer_angle_scale = -9.1f;
  float max_steer_angle = 0.5f;
  float max_steer_rate = 5;
  float steer_accel_constant = 12.1f;
  float steer_accel_drag_constant = 1.0f;
  float brake_delay_constant = 6.6f;
  float brake_delay_constant_neg = 8.2f;
  float max_brake_rate_neg = 0.9f;
  float max_brake_rate_pos = 0.33f;
  float wheel_base = 0.3f;
  float low_min_throttle = 0.13f;
  float gravity = -9.81f;
  int gear_sign = 1;
};

using namespace MPPI_internal;

class RacerDubins : public Dynamics<RacerDubins, RacerDubinsParams>
{
public:
  using PARENT_CLASS = Dynamics<RacerDubins, RacerDubinsParams>;
  RacerDubins(hipStream_t stream = nullptr) : PARENT_CLASS(stream)
  {
  }
  static const char* getDynamicsModelName()
  {
    return "RACER Dubins Model";
  }

  /** racer_dubins.cu:138-165 */
  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta = nullptr)
  {
    const bool enable_brake = control[C_INDEX(THROTTLE_BRAKE)] < 0;

    state_der[S_INDEX(BRAKE_STATE)] =
        fminf(fmaxf((enable_brake * -control[C_INDEX(THROTTLE_BRAKE)] - state[S_INDEX(BRAKE_STATE)]) *
                        this->params_.brake_delay_constant,
                    -this->params_.max_brake_rate_neg),
              this->params_.max_brake_rate_pos);
    // applying position throttle
    state_der[S_INDEX(VEL_X)] =
        (!enable_brake) * this->params_.c_t[0] * control[0] * this->params_.gear_sign +
        this->params_.c_b[0] * state[S_INDEX(BRAKE_STATE)] * (state[S_INDEX(VEL_X)] >= 0 ? -1 : 1) -
        this->params_.c_v[0] * state[S_INDEX(VEL_X)] + this->params_.c_0;
    state_der[S_INDEX(YAW)] = (state[S_INDEX(VEL_X)] / this->params_.wheel_base) *
                              mppi::det::tan(state[S_INDEX(STEER_ANGLE)] / this->params_.steer_angle_scale);
    float sin_yaw, cos_yaw;
    const float yaw = angle_utils::normalizeAngle(state[S_INDEX(YAW)]);
    mppi::det::sincos(yaw, &sin_yaw, &cos_yaw);
    state_der[S_INDEX(POS_X)] = state[S_INDEX(VEL_X)] * cos_yaw;
    state_der[S_INDEX(POS_Y)] = state[S_INDEX(VEL_X)] * sin_yaw;
    state_der[S_INDEX(STEER_ANGLE)] =
        fmaxf(fminf((control[1] * this->params_.steer_command_angle_scale - state[S_INDEX(STEER_ANGLE)]) *
                        this->params_.steering_constant,
                    this->params_.max_steer_rate),
              -this->params_.max_steer_rate);
  }

  /** racer_dubins.cu:73-98: Euler step of the first six states, yaw wrapped, steering angle and brake state clamped,
   *  the steering rate state is the steering angle's derivative */
  __device__ inline void updateState(float* state, float* next_state, float* state_der, const float dt)
  {
    int i, p_index, step;
    mppi::p1::getParallel1DIndex<mppi::p1::Parallel1Dir::THREAD_Y>(p_index, step);
    for (i = p_index; i < 6; i += step)
    {
      next_state[i] = state[i] + state_der[i] * dt;
      if (i == S_INDEX(YAW))
      {
        next_state[i] = angle_utils::normalizeAngle(next_state[i]);
      }
      if (i == S_INDEX(STEER_ANGLE))
      {
        next_state[i] = fmaxf(fminf(next_state[i], this->params_.max_steer_angle), -this->params_.max_steer_angle);
        next_state[S_INDEX(STEER_ANGLE_RATE)] = state_der[i];
      }
      if (i == S_INDEX(BRAKE_STATE))
      {
        next_state[i] = fminf(fmaxf(next_state[i], 0.0f), 1.0f);
      }
    }
  }
};

#endif
