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
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  using PARENT_CLASS = Dynamics<RacerDubins, RacerDubinsParams>;
  RacerDubins(hipStream_t stream = nullptr) : PARENT_CLASS(stream)
  {
  }
  static const char* getDynamicsModelName()
  {
    return "RACER Dubins Model";
  }

  /**
   * Continuous-time model (reference: racer_dubins.cu:138-165, same arithmetic in the same order):
   *   brake lag      d(brake)/dt = clamp((brake_cmd - brake) k_brake, -r_neg, +r_pos),  brake_cmd = max(-u0, 0)
   *   longitudinal   dv/dt = [u0 >= 0] c_t u0 gear + c_b brake (-sgn v) - c_v v + c_0
   *   yaw            dyaw/dt = v / L * tan(steer / steer_angle_scale)
   *   position       d(x, y)/dt = v (cos yaw, sin yaw),  yaw wrapped to (-pi, pi] first
   *   steering lag   d(steer)/dt = clamp((u1 k_cmd - steer) k_steer, +-max_steer_rate)
   */
  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta = nullptr)
  {
    const RacerDubinsParams& p = this->params_;
    const float v = state[S_INDEX(VEL_X)], brake = state[S_INDEX(BRAKE_STATE)], steer = state[S_INDEX(STEER_ANGLE)];
    const float u_tb = control[C_INDEX(THROTTLE_BRAKE)], u_steer = control[C_INDEX(STEER_CMD)];
    const bool braking = u_tb < 0;

    const float brake_rate = (braking * -u_tb - brake) * p.brake_delay_constant;
    state_der[S_INDEX(BRAKE_STATE)] = fminf(fmaxf(brake_rate, -p.max_brake_rate_neg), p.max_brake_rate_pos);

    const float drive = (!braking) * p.c_t[0] * u_tb * p.gear_sign;
    const float drag = p.c_b[0] * brake * (v >= 0 ? -1 : 1);
    state_der[S_INDEX(VEL_X)] = drive + drag - p.c_v[0] * v + p.c_0;

    state_der[S_INDEX(YAW)] = (v / p.wheel_base) * mppi::det::tan(steer / p.steer_angle_scale);

    float s_yaw, c_yaw;
    mppi::det::sincos(angle_utils::normalizeAngle(state[S_INDEX(YAW)]), &s_yaw, &c_yaw);
    state_der[S_INDEX(POS_X)] = v * c_yaw;
    state_der[S_INDEX(POS_Y)] = v * s_yaw;

    const float steer_rate = (u_steer * p.steer_command_angle_scale - steer) * p.steering_constant;
    state_der[S_INDEX(STEER_ANGLE)] = fmaxf(fminf(steer_rate, p.max_steer_rate), -p.max_steer_rate);
  }

  /**
   * HOST: ColoredMPPI's state leash for the RACER models (reference: RacerDubinsImpl::enforceLeash,
   * dynamics/racer_dubins/racer_dubins.cu:176-240): the position error is rotated into the body frame of the true state,
   * clamped there to +-leash[POS_X] / +-leash[POS_Y] and rotated back; the yaw error is the shortest angular distance
   * and the leashed yaw is wrapped; every other state follows the base rule (dynamics.cuh:448-466).  Plain libm on the
   * host, as the reference.
   */
  void enforceLeash(const float* state_true, const float* state_nominal, const float* leash_values, float* state_output) const
  {
    constexpr int YAW = S_INDEX(YAW), POS_X = S_INDEX(POS_X), POS_Y = S_INDEX(POS_Y);
    for (int i = 0; i < STATE_DIM; i++)
      state_output[i] = state_true[i];
    float dx = state_nominal[POS_X] - state_true[POS_X];
    float dy = state_nominal[POS_Y] - state_true[POS_Y];
    const float c = cosf(state_true[YAW]), s = sinf(state_true[YAW]);
    float dx_body = dx * c + dy * s;
    float dy_body = -dx * s + dy * c;
    const float y_leash = leash_values[POS_Y], x_leash = leash_values[POS_X];
    dx_body = fminf(fmaxf(dx_body, -x_leash), x_leash);
    dy_body = fminf(fmaxf(dy_body, -y_leash), y_leash);
    dx = dx_body * c + -dy_body * s;
    dy = dx_body * s + dy_body * c;
    state_output[POS_X] += dx;
    state_output[POS_Y] += dy;
    for (int i = 0; i < STATE_DIM; i++)
    {
      if (i == POS_X || i == POS_Y)
        continue;
      const float diff = (i == YAW) ? angle_utils::shortestAngularDistance(state_true[i], state_nominal[i]) :
                                      state_nominal[i] - state_true[i];
      if (leash_values[i] < fabsf(diff))
      {
        const float leash_dir = fminf(fmaxf(diff, -leash_values[i]), leash_values[i]);
        state_output[i] = state_true[i] + leash_dir;
        if (i == YAW)
          state_output[i] = angle_utils::normalizeAngle(state_output[i]);
      }
      else
      {
        state_output[i] = state_nominal[i];
      }
    }
  }

  /**
   * Euler step of the six integrated states (reference: racer_dubins.cu:73-98): the yaw is wrapped, the steering angle
   * and the brake state are clamped to their physical ranges, and the seventh state simply records the steering rate
   * that was applied.
   */
  __device__ inline void updateState(float* state, float* next_state, float* state_der, const float dt)
  {
    int first, stride;
    mppi::p1::getParallel1DIndex<mppi::p1::Parallel1Dir::THREAD_Y>(first, stride);
    constexpr int INTEGRATED = 6;
    for (int i = first; i < INTEGRATED; i += stride)
    {
      float xn = state[i] + state_der[i] * dt;
      switch (i)
      {
        case S_INDEX(YAW):
          xn = angle_utils::normalizeAngle(xn);
          break;
        case S_INDEX(STEER_ANGLE):
          xn = fmaxf(fminf(xn, this->params_.max_steer_angle), -this->params_.max_steer_angle);
          next_state[S_INDEX(STEER_ANGLE_RATE)] = state_der[i];
          break;
        case S_INDEX(BRAKE_STATE):
          xn = fminf(fmaxf(xn, 0.0f), 1.0f);
          break;
        default:
          break;
      }
      next_state[i] = xn;
    }
  }
};

#endif
