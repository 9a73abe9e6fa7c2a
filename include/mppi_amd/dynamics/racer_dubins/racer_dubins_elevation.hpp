/**
 * RacerDubinsElevation plugin — the RACER Dubins car on an elevation map: speed-dependent throttle / brake / drag
 * coefficients, gravity along the pitch of the terrain, roll / pitch from a four-wheel "static settling" on the map, and
 * a 4x4 covariance of (v, yaw, x, y) propagated along the rollout (Sigma <- (I + A dt) Sigma (I + A dt)^T + Q dt).
 *
 * Reference: include/mppi/dynamics/racer_dubins/racer_dubins_elevation.cuh:16-60 (parameters, state layout),
 * racer_dubins_elevation.cu:836-874 (device step), :753-798 (device computeParametricAccelDeriv), :336-419
 * (computeUncertaintyJacobian), :421-506 (computeQ), :508-633 (covariance <-> state), :672-738
 * (computeUncertaintyPropagation), :72-237 (setOutputs); racer_dubins.cu:281-305 (device brake / steering lags), :69-96
 * (device updateState), :358-434 (RACER::computeStaticSettling); utils/math_utils.h:457-482 (Euler2DCM_NWU), :375-391
 * (RotatePointByDCM), utils/matrix_mult_utils.cuh:82-192 (gemm1).  DEVICE flavour of every formula (angles wrapped before
 * the trigonometric calls, roll taken into the side force); the reference's fast intrinsics become det_math.h functions
 * (__sinf/__cosf/__sincosf -> det::sin/cos/sincos, __tanf -> det::tan, asinf -> det::asin).
 *
 * How it sits on the hardware: the reference gives every rollout a 48-float SharedBlock in shared memory and strides the
 * sixteen matrix entries over threadIdx.y with block barriers between the five phases.  Here a rollout is ONE lane: the
 * three 4x4 matrices are locals (fully unrolled loops -> VGPRs), there is no barrier and no LDS traffic, and the sums
 * run in gemm1's order (k = 0..3, multiply then add) so the oracle reproduces them bit for bit.  With more than one lane
 * per rollout (BY > 1) every lane computes the same values and stores them — nothing is read back inside step().
 *
 * The elevation map is a TwoDTextureHelper<1> member (utils/texture_helpers/two_d_texture_helper.hpp): four bilinear
 * lookups per step through ordinary loads — a map of a few hundred KB stays in the L2 of every XCD.  The map arrives as
 * the "elevation_map" blob ({height, width}) and its frame as the "elevation_map_transform" blob (origin[3],
 * rotations[9] row-major, resolution[3]) of mppi_set_model_blob(); without a map the car settles flat (roll = pitch = 0).
 */
#ifndef MPPI_AMD_RACER_DUBINS_ELEVATION_HPP_
#define MPPI_AMD_RACER_DUBINS_ELEVATION_HPP_

#include "mppi_amd/dynamics/racer_dubins/racer_dubins.hpp"
#include "mppi_amd/utils/texture_helpers/two_d_texture_helper.hpp"

#ifndef U_INDEX
#define U_IND_CLASS(CLASS, enum_val) E_INDEX(CLASS::UncertaintyIndex, enum_val)
#define U_INDEX(enum_val) U_IND_CLASS(PARENT_CLASS::DYN_PARAMS_T, enum_val)
#endif
/* index shorthands on the (non-dependent) parameter struct: the classes below are templates over the derived class */
#define RDE_S(enum_val) S_IND_CLASS(RacerDubinsElevationParams, enum_val)
#define RDE_C(enum_val) C_IND_CLASS(RacerDubinsElevationParams, enum_val)
#define RDE_O(enum_val) O_IND_CLASS(RacerDubinsElevationParams, enum_val)
#define RDE_U(enum_val) U_IND_CLASS(RacerDubinsElevationParams, enum_val)

/** reference: racer_dubins_elevation.cuh:16-60 */
struct RacerDubinsElevationParams : public RacerDubinsParams
{
  enum class StateIndex : int
  {
    VEL_X = 0,
    YAW,
    POS_X,
    POS_Y,
    STEER_ANGLE,
    BRAKE_STATE,
    ROLL,
    PITCH,
    STEER_ANGLE_RATE,
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
    NUM_STATES
  };
  enum class UncertaintyIndex : int
  {
    VEL_X = 0,
    YAW,
    POS_X,
    POS_Y,
    NUM_UNCERTAINTIES
  };
  float clamp_ax = 5.5f;
  float K_x = 1.0f;      ///< feedback gains of the tracking controller the covariance assumes
  float K_y = 1.0f;
  float K_yaw = 1.0f;
  float K_vel_x = 1.0f;
  float Q_x_acc = 1.0f;  ///< process noise: v from |a_x| ...
  float Q_x_v[3] = { 41.74219f, -0.8187027f, -2.2131343f };  ///< ... and from |v|, per speed regime
  float Q_y_f = 0.1f;                                        ///< x, y from the side force
  float Q_omega_v = 0.001f;                                  ///< yaw from |v|
  float Q_omega_steering = 0.0f;                             ///< yaw from |steering angle|
};

/** reference: RacerDubinsElevationImpl<CLASS_T, PARAMS_T> (racer_dubins_elevation.cuh:62-150); CLASS_T is the model that is
 *  instantiated (RacerDubinsElevation below, RacerDubinsElevationLSTMSteering in its own header) */
template <class CLASS_T>
class RacerDubinsElevationImpl : public MPPI_internal::Dynamics<CLASS_T, RacerDubinsElevationParams>
{
public:
  using PARENT_CLASS = MPPI_internal::Dynamics<CLASS_T, RacerDubinsElevationParams>;
  static constexpr int STATE_DIM = RDE_S(NUM_STATES);
  static constexpr int CONTROL_DIM = RDE_C(NUM_CONTROLS);
  static constexpr int OUTPUT_DIM = RDE_O(NUM_OUTPUTS);
  static const int UNCERTAINTY_DIM = U_IND_CLASS(RacerDubinsElevationParams, NUM_UNCERTAINTIES);
  static constexpr int UD = UNCERTAINTY_DIM;

  /** the elevation map (texture 0), reference: tex_helper_ of racer_dubins_elevation.cuh:88-96 */
  mppi::texture::TwoDTextureHelper<1, 1> tex_helper_;

  RacerDubinsElevationImpl(hipStream_t stream = nullptr) : PARENT_CLASS(stream)
  {
  }
  static const char* getDynamicsModelName()
  {
    return "RACER Dubins w/ Elevation Model";
  }

  /** column-major index of a 4x4 entry (reference: matrix_mult_utils.cuh:27-30) */
  __host__ __device__ static constexpr int cm(const int row, const int col)
  {
    return col * UD + row;
  }

  /** the three speed regimes of the longitudinal coefficients: |v| <= 0.2, <= 3, above */
  __device__ static inline int speedRegime(const float vx)
  {
    const float linear_brake_slope = 0.2f;
    return (fabsf(vx) > linear_brake_slope && fabsf(vx) <= 3.0f) + (fabsf(vx) > 3.0f) * 2;
  }

  /**
   * base initializeDynamics (output <- first states), plus a defined value in the outputs step() never writes
   * (FILLER_1; the reference leaves them to whatever the output buffer held)
   */
  __device__ inline void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    PARENT_CLASS::initializeDynamics(state, control, output, theta_s, t_0, dt);
    int first, stride;
    mppi::p1::getParallel1DIndex<mppi::p1::Parallel1Dir::THREAD_Y>(first, stride);
    for (int i = STATE_DIM + first; i < OUTPUT_DIM; i += stride)
      output[i] = 0.0f;
  }

  /** not used: step() below is the whole model (the reference's step() does not go through computeDynamics either) */
  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta = nullptr)
  {
  }

  /** a[index] of a three-entry parameter table by selects (a run-time index into a kernel-argument array would move the
   *  table into scratch memory) */
  __device__ static inline float pick3(const float (&a)[3], const int index)
  {
    return index == 0 ? a[0] : (index == 1 ? a[1] : a[2]);
  }

  /** racer_dubins.cu:281-293 (brake lag, faster on release) and :295-305 (steering lag) */
  __device__ inline void computeParametricDelayDeriv(const float* state, const float* control, float* state_der) const
  {
    const RacerDubinsElevationParams& p = this->params_;
    const bool enable_brake = control[RDE_C(THROTTLE_BRAKE)] < 0.0f;
    const float brake_error = (enable_brake * -control[RDE_C(THROTTLE_BRAKE)] - state[RDE_S(BRAKE_STATE)]);
    state_der[RDE_S(BRAKE_STATE)] = fminf(fmaxf((brake_error > 0) * brake_error * p.brake_delay_constant +
                                                      (brake_error < 0) * brake_error * p.brake_delay_constant_neg,
                                                  -p.max_brake_rate_neg),
                                            p.max_brake_rate_pos);
  }
  __device__ inline void computeParametricSteerDeriv(const float* state, const float* control, float* state_der) const
  {
    const RacerDubinsElevationParams& p = this->params_;
    state_der[RDE_S(STEER_ANGLE)] =
        fmaxf(fminf((control[RDE_C(STEER_CMD)] * p.steer_command_angle_scale - state[RDE_S(STEER_ANGLE)]) *
                        p.steering_constant,
                    p.max_steer_rate),
              -p.max_steer_rate);
  }

  /** racer_dubins_elevation.cu:753-798: longitudinal acceleration per speed regime, clamp, gravity along the pitch; yaw
   *  rate of the bicycle; position rate */
  __device__ inline void computeParametricAccelDeriv(const float* state, const float* control, float* state_der) const
  {
    const RacerDubinsElevationParams& p = this->params_;
    const float vx = state[RDE_S(VEL_X)];
    const float linear_brake_slope = 0.2f;
    const bool enable_brake = control[RDE_C(THROTTLE_BRAKE)] < 0.0f;
    const int index = speedRegime(vx);
    const float brake_state = fminf(fmaxf(state[RDE_S(BRAKE_STATE)], 0.0f), 0.25f);
    const float c_t = pick3(p.c_t, index), c_b = pick3(p.c_b, index), c_v = pick3(p.c_v, index);
    float throttle = c_t * control[RDE_C(THROTTLE_BRAKE)];
    float brake = c_b * brake_state * (vx >= 0.0f ? -1.0f : 1.0f);
    if (fabsf(vx) <= linear_brake_slope)
    {
      throttle = c_t * fmaxf(control[RDE_C(THROTTLE_BRAKE)] - p.low_min_throttle, 0.0f);
      brake = c_b * brake_state * -vx;
    }
    float ax = (!enable_brake) * throttle * p.gear_sign + brake - c_v * vx + p.c_0;
    ax = fminf(fmaxf(ax, -p.clamp_ax), p.clamp_ax);
    if (fabsf(state[RDE_S(PITCH)]) < 1.57079637050628662109375f)
    {
      ax -= p.gravity * mppi::det::sin(angle_utils::normalizeAngle(state[RDE_S(PITCH)]));
    }
    state_der[RDE_S(VEL_X)] = ax;
    state_der[RDE_S(YAW)] =
        (vx / p.wheel_base) * mppi::det::tan(angle_utils::normalizeAngle(state[RDE_S(STEER_ANGLE)] / p.steer_angle_scale));
    float s_yaw, c_yaw;
    mppi::det::sincos(angle_utils::normalizeAngle(state[RDE_S(YAW)]), &s_yaw, &c_yaw);
    state_der[RDE_S(POS_X)] = vx * c_yaw;
    state_der[RDE_S(POS_Y)] = vx * s_yaw;
  }

  /** racer_dubins_elevation.cu:800-834: Euler step of the six integrated states, as in RacerDubins */
  __device__ inline void updateState(const float* state, float* next_state, const float* state_der, const float dt) const
  {
    const RacerDubinsElevationParams& p = this->params_;
#pragma unroll
    for (int i = 0; i < 6; i++)
    {
      float xn = state[i] + state_der[i] * dt;
      switch (i)
      {
        case RDE_S(YAW):
          xn = angle_utils::normalizeAngle(xn);
          break;
        case RDE_S(STEER_ANGLE):
          xn = fmaxf(fminf(xn, p.max_steer_angle), -p.max_steer_angle);
          next_state[RDE_S(STEER_ANGLE_RATE)] = state_der[RDE_S(STEER_ANGLE)];
          break;
        case RDE_S(BRAKE_STATE):
          xn = fminf(fmaxf(xn, 0.0f), 1.0f);
          break;
        default:
          break;
      }
      next_state[i] = xn;
    }
  }

  /** racer_dubins_elevation.cu:336-419 (device branch): A = df/dx + df/du K of the (v, yaw, x, y) error dynamics */
  __device__ inline void computeUncertaintyJacobian(const float* state, float* A) const
  {
    const RacerDubinsElevationParams& p = this->params_;
    const float vx = state[RDE_S(VEL_X)];
    float sin_yaw, cos_yaw;
    mppi::det::sincos(angle_utils::normalizeAngle(state[RDE_S(YAW)]), &sin_yaw, &cos_yaw);
    const float delta = state[RDE_S(STEER_ANGLE)] / p.steer_angle_scale;
    const float tan_steer_angle = mppi::det::tan(delta);
    const float cos_delta = mppi::det::cos(delta);
    const float cos_2_delta = cos_delta * cos_delta;
    const int index = speedRegime(vx);
    const float brake_state = fminf(fmaxf(state[RDE_S(BRAKE_STATE)], 0.0f), 0.25f);

    A[cm(RDE_U(VEL_X), RDE_U(VEL_X))] = -pick3(p.c_v, index) - p.K_vel_x - (index == 0 ? 1.0f : 0.0f) * p.c_b[0] * brake_state;
    A[cm(RDE_U(VEL_X), RDE_U(YAW))] = 0.0f;
    A[cm(RDE_U(VEL_X), RDE_U(POS_X))] = -p.K_x * cos_yaw;
    A[cm(RDE_U(VEL_X), RDE_U(POS_Y))] = -p.K_x * sin_yaw;

    A[cm(RDE_U(YAW), RDE_U(VEL_X))] = tan_steer_angle / (p.wheel_base);
    A[cm(RDE_U(YAW), RDE_U(YAW))] = -fabsf(vx) * p.K_yaw / (p.wheel_base * cos_2_delta);
    A[cm(RDE_U(YAW), RDE_U(POS_X))] = vx * p.K_y * sin_yaw / (p.wheel_base * cos_2_delta);
    A[cm(RDE_U(YAW), RDE_U(POS_Y))] = -vx * p.K_y * cos_yaw / (p.wheel_base * cos_2_delta);

    A[cm(RDE_U(POS_X), RDE_U(VEL_X))] = cos_yaw;
    A[cm(RDE_U(POS_X), RDE_U(YAW))] = -sin_yaw * vx;
    A[cm(RDE_U(POS_X), RDE_U(POS_X))] = 0.0f;
    A[cm(RDE_U(POS_X), RDE_U(POS_Y))] = 0.0f;

    A[cm(RDE_U(POS_Y), RDE_U(VEL_X))] = sin_yaw;
    A[cm(RDE_U(POS_Y), RDE_U(YAW))] = cos_yaw * vx;
    A[cm(RDE_U(POS_Y), RDE_U(POS_X))] = 0.0f;
    A[cm(RDE_U(POS_Y), RDE_U(POS_Y))] = 0.0f;
  }

  /** racer_dubins_elevation.cu:421-506 (device branch): process noise from |a_x|, |v|, steering and the side force */
  __device__ inline void computeQ(const float* state, const float* state_der, float* Q) const
  {
    const RacerDubinsElevationParams& p = this->params_;
    const float abs_vx = fabsf(state[RDE_S(VEL_X)]);
    const float abs_acc_x = fabsf(state_der[RDE_S(VEL_X)]);
    const float delta = state[RDE_S(STEER_ANGLE)] / p.steer_angle_scale;
    float sin_yaw, cos_yaw;
    mppi::det::sincos(angle_utils::normalizeAngle(state[RDE_S(YAW)]), &sin_yaw, &cos_yaw);
    const float tan_steer_angle = mppi::det::tan(delta);
    const float sin_roll = mppi::det::sin(angle_utils::normalizeAngle(state[RDE_S(ROLL)]));
    const float side_force = (abs_vx * abs_vx) * tan_steer_angle / p.wheel_base + p.gravity * sin_roll;
    const float Q_11 = fabsf(p.Q_y_f * fabsf(side_force) * fmaxf(abs_vx - 2, 0.0f));
    const int index = speedRegime(state[RDE_S(VEL_X)]);
#pragma unroll
    for (int i = 0; i < UD * UD; i++)
      Q[i] = 0.0f;
    Q[cm(RDE_U(VEL_X), RDE_U(VEL_X))] = p.Q_x_acc * abs_acc_x + pick3(p.Q_x_v, index) * abs_vx;
    Q[cm(RDE_U(YAW), RDE_U(YAW))] = abs_vx * (p.Q_omega_steering * fabsf(delta) + p.Q_omega_v);
    Q[cm(RDE_U(POS_X), RDE_U(POS_X))] = Q_11 * sin_yaw * sin_yaw;
    Q[cm(RDE_U(POS_X), RDE_U(POS_Y))] = -Q_11 * sin_yaw * cos_yaw;
    Q[cm(RDE_U(POS_Y), RDE_U(POS_Y))] = Q_11 * cos_yaw * cos_yaw;
    Q[cm(RDE_U(POS_Y), RDE_U(POS_X))] = -Q_11 * sin_yaw * cos_yaw;
  }

  /** racer_dubins_elevation.cu:508-565 */
  __device__ static inline void uncertaintyStateToMatrix(const float* state, float* M)
  {
    M[cm(RDE_U(VEL_X), RDE_U(VEL_X))] = state[RDE_S(UNCERTAINTY_VEL_X)];
    M[cm(RDE_U(YAW), RDE_U(VEL_X))] = state[RDE_S(UNCERTAINTY_YAW_VEL_X)];
    M[cm(RDE_U(POS_X), RDE_U(VEL_X))] = state[RDE_S(UNCERTAINTY_POS_X_VEL_X)];
    M[cm(RDE_U(POS_Y), RDE_U(VEL_X))] = state[RDE_S(UNCERTAINTY_POS_Y_VEL_X)];
    M[cm(RDE_U(VEL_X), RDE_U(YAW))] = state[RDE_S(UNCERTAINTY_YAW_VEL_X)];
    M[cm(RDE_U(YAW), RDE_U(YAW))] = state[RDE_S(UNCERTAINTY_YAW)];
    M[cm(RDE_U(POS_X), RDE_U(YAW))] = state[RDE_S(UNCERTAINTY_POS_X_YAW)];
    M[cm(RDE_U(POS_Y), RDE_U(YAW))] = state[RDE_S(UNCERTAINTY_POS_Y_YAW)];
    M[cm(RDE_U(VEL_X), RDE_U(POS_X))] = state[RDE_S(UNCERTAINTY_POS_X_VEL_X)];
    M[cm(RDE_U(YAW), RDE_U(POS_X))] = state[RDE_S(UNCERTAINTY_POS_X_YAW)];
    M[cm(RDE_U(POS_X), RDE_U(POS_X))] = state[RDE_S(UNCERTAINTY_POS_X)];
    M[cm(RDE_U(POS_Y), RDE_U(POS_X))] = state[RDE_S(UNCERTAINTY_POS_X_Y)];
    M[cm(RDE_U(VEL_X), RDE_U(POS_Y))] = state[RDE_S(UNCERTAINTY_POS_Y_VEL_X)];
    M[cm(RDE_U(YAW), RDE_U(POS_Y))] = state[RDE_S(UNCERTAINTY_POS_Y_YAW)];
    M[cm(RDE_U(POS_X), RDE_U(POS_Y))] = state[RDE_S(UNCERTAINTY_POS_X_Y)];
    M[cm(RDE_U(POS_Y), RDE_U(POS_Y))] = state[RDE_S(UNCERTAINTY_POS_Y)];
  }

  /** racer_dubins_elevation.cu:567-612: the lower triangle goes back into the state */
  __device__ static inline void uncertaintyMatrixToState(const float* M, float* state)
  {
    state[RDE_S(UNCERTAINTY_VEL_X)] = M[cm(RDE_U(VEL_X), RDE_U(VEL_X))];
    state[RDE_S(UNCERTAINTY_YAW_VEL_X)] = M[cm(RDE_U(YAW), RDE_U(VEL_X))];
    state[RDE_S(UNCERTAINTY_POS_X_VEL_X)] = M[cm(RDE_U(POS_X), RDE_U(VEL_X))];
    state[RDE_S(UNCERTAINTY_POS_Y_VEL_X)] = M[cm(RDE_U(POS_Y), RDE_U(VEL_X))];
    state[RDE_S(UNCERTAINTY_YAW)] = M[cm(RDE_U(YAW), RDE_U(YAW))];
    state[RDE_S(UNCERTAINTY_POS_X_YAW)] = M[cm(RDE_U(POS_X), RDE_U(YAW))];
    state[RDE_S(UNCERTAINTY_POS_Y_YAW)] = M[cm(RDE_U(POS_Y), RDE_U(YAW))];
    state[RDE_S(UNCERTAINTY_POS_X)] = M[cm(RDE_U(POS_X), RDE_U(POS_X))];
    state[RDE_S(UNCERTAINTY_POS_X_Y)] = M[cm(RDE_U(POS_Y), RDE_U(POS_X))];
    state[RDE_S(UNCERTAINTY_POS_Y)] = M[cm(RDE_U(POS_Y), RDE_U(POS_Y))];
  }

  /**
   * racer_dubins_elevation.cu:672-738: Sigma' = (I + A dt) Sigma (I + A dt)^T + Q dt.  The two products are gemm1's sums
   * (matrix_mult_utils.cuh:82-192): accumulator from zero, k ascending, one multiply and one add per term.
   */
  __device__ inline void computeUncertaintyPropagation(const float* state, const float* state_der, float* next_state,
                                                       const float dt) const
  {
    float A[UD * UD], Sigma_a[UD * UD], Sigma_b[UD * UD];
    computeUncertaintyJacobian(state, A);
    uncertaintyStateToMatrix(state, Sigma_a);
#pragma unroll
    for (int i = 0; i < UD * UD; i++)
      A[i] = (i % (UD + 1) == 0) + A[i] * dt;
#pragma unroll
    for (int n = 0; n < UD; n++)
#pragma unroll
      for (int m = 0; m < UD; m++)
      {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < UD; k++)
          acc += A[cm(m, k)] * Sigma_a[cm(k, n)];
        Sigma_b[cm(m, n)] = acc;
      }
#pragma unroll
    for (int n = 0; n < UD; n++)
#pragma unroll
      for (int m = 0; m < UD; m++)
      {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < UD; k++)
          acc += Sigma_b[cm(m, k)] * A[cm(n, k)];
        Sigma_a[cm(m, n)] = acc;
      }
    computeQ(state, state_der, Sigma_b);
#pragma unroll
    for (int i = 0; i < UD * UD; i++)
      Sigma_a[i] += Sigma_b[i] * dt;
    uncertaintyMatrixToState(Sigma_a, next_state);
  }

  /**
   * RACER::computeStaticSettling (racer_dubins.cu:358-434): the four wheel contact points of the body at (x, y, yaw) with
   * the CURRENT roll and pitch are looked up in the elevation map; roll and pitch for the next state follow from the
   * height differences across the track width (2 x 0.737 m) and the wheel base (2.981 m).
   */
  __device__ inline void computeStaticSettling(const float yaw, const float x, const float y, float& roll, float& pitch,
                                               float& height) const
  {
    height = 0.0f;
    if (!tex_helper_.checkTextureUse(0))
    {
      roll = 0.0f;
      pitch = 0.0f;
      return;
    }
    // Euler2DCM_NWU, device branch (math_utils.h:457-482)
    float sin_phi, cos_phi, sin_theta, cos_theta, sin_psi, cos_psi;
    mppi::det::sincos(angle_utils::normalizeAngle(roll), &sin_phi, &cos_phi);
    mppi::det::sincos(angle_utils::normalizeAngle(pitch), &sin_theta, &cos_theta);
    mppi::det::sincos(angle_utils::normalizeAngle(yaw), &sin_psi, &cos_psi);
    float M[3][3];
    M[0][0] = cos_theta * cos_psi;
    M[0][1] = sin_phi * sin_theta * cos_psi - cos_phi * sin_psi;
    M[0][2] = cos_phi * sin_theta * cos_psi + sin_phi * sin_psi;
    M[1][0] = cos_theta * sin_psi;
    M[1][1] = sin_phi * sin_theta * sin_psi + cos_phi * cos_psi;
    M[1][2] = cos_phi * sin_theta * sin_psi - sin_phi * cos_psi;
    M[2][0] = -sin_theta;
    M[2][1] = sin_phi * cos_theta;
    M[2][2] = cos_phi * cos_theta;
    const float body_pose[3] = { x, y, 0.0f };
    // front left, front right, rear left, rear right (racer_dubins.cu:363-366)
    const float offsets[4][3] = { { 2.981f, 0.737f, 0.0f }, { 2.981f, -0.737f, 0.0f }, { 0.0f, 0.737f, 0.0f }, { 0.0f, -0.737f, 0.0f } };
    float h[4], world[4][3];
#pragma unroll
    for (int w = 0; w < 4; w++)
    {
#pragma unroll
      for (int r = 0; r < 3; r++)
      {  // RotatePointByDCM -> gemm1<3, 3, 1>: three terms accumulated from zero, then the body pose
        float acc = 0.0f;
        acc += M[r][0] * offsets[w][0];
        acc += M[r][1] * offsets[w][1];
        acc += M[r][2] * offsets[w][2];
        world[w][r] = acc + body_pose[r];
      }
    }
    tex_helper_.template queryTextureAtWorldPoseBatch<4>(0, world, h);  // the sixteen loads of the four wheels in flight together
    const float front_left_height = h[0], front_right_height = h[1], rear_left_height = h[2], rear_right_height = h[3];

    float front_diff = front_left_height - front_right_height;
    front_diff = fmaxf(fminf(front_diff, 0.736f * 2.0f), -0.736f * 2.0f);
    float rear_diff = rear_left_height - rear_right_height;
    rear_diff = fmaxf(fminf(rear_diff, 0.736f * 2.0f), -0.736f * 2.0f);
    const float front_roll = mppi::det::asin(front_diff / (0.737f * 2.0f));
    const float rear_roll = mppi::det::asin(rear_diff / (0.737f * 2.0f));
    roll = (front_roll + rear_roll) / 2.0f;

    float left_diff = rear_left_height - front_left_height;
    left_diff = fmaxf(fminf(left_diff, 2.98f), -2.98f);
    float right_diff = rear_right_height - front_right_height;
    right_diff = fmaxf(fminf(right_diff, 2.98f), -2.98f);
    const float left_pitch = mppi::det::asin((left_diff) / 2.981f);
    const float right_pitch = mppi::det::asin((right_diff) / 2.981f);
    pitch = (left_pitch + right_pitch) / 2.0f;

    height = (rear_left_height + rear_right_height) / 2.0f;

    // 2 pi: a rotation that accidentally uses such a value is the identity
    const float two_pi = 2.0f * 3.14159274101257324219f;
    if (!isfinite(roll) || fabsf(roll) > 3.14159274101257324219f)
      roll = two_pi;
    if (!isfinite(pitch) || fabsf(pitch) > 3.14159274101257324219f)
      pitch = two_pi;
    if (!isfinite(height))
      height = 0.0f;
  }

  /** racer_dubins_elevation.cu:72-237: every output but BASELINK_POS_I_Z (static settling) and FILLER_1 */
  __device__ static inline void setOutputs(const float* state_der, const float* next_state, float* output)
  {
    const float nan = __builtin_nanf("");
    output[RDE_O(BASELINK_VEL_B_X)] = next_state[RDE_S(VEL_X)];
    output[RDE_O(BASELINK_VEL_B_Y)] = 0.0f;
    output[RDE_O(BASELINK_POS_I_X)] = next_state[RDE_S(POS_X)];
    output[RDE_O(BASELINK_POS_I_Y)] = next_state[RDE_S(POS_Y)];
    output[RDE_O(PITCH)] = next_state[RDE_S(PITCH)];
    output[RDE_O(ROLL)] = next_state[RDE_S(ROLL)];
    output[RDE_O(YAW)] = next_state[RDE_S(YAW)];
    output[RDE_O(STEER_ANGLE)] = next_state[RDE_S(STEER_ANGLE)];
    output[RDE_O(STEER_ANGLE_RATE)] = next_state[RDE_S(STEER_ANGLE_RATE)];
    output[RDE_O(WHEEL_FORCE_UP_MAX)] = nan;
    output[RDE_O(WHEEL_FORCE_FWD_MAX)] = nan;
    output[RDE_O(WHEEL_FORCE_SIDE_MAX)] = nan;
    output[RDE_O(ACCEL_X)] = state_der[RDE_S(VEL_X)];
    output[RDE_O(ACCEL_Y)] = 0.0f;
    output[RDE_O(OMEGA_Z)] = state_der[RDE_S(YAW)];
    output[RDE_O(UNCERTAINTY_VEL_X)] = next_state[RDE_S(UNCERTAINTY_VEL_X)];
    output[RDE_O(UNCERTAINTY_YAW_VEL_X)] = next_state[RDE_S(UNCERTAINTY_YAW_VEL_X)];
    output[RDE_O(UNCERTAINTY_POS_X_VEL_X)] = next_state[RDE_S(UNCERTAINTY_POS_X_VEL_X)];
    output[RDE_O(UNCERTAINTY_POS_Y_VEL_X)] = next_state[RDE_S(UNCERTAINTY_POS_Y_VEL_X)];
    output[RDE_O(UNCERTAINTY_YAW)] = next_state[RDE_S(UNCERTAINTY_YAW)];
    output[RDE_O(UNCERTAINTY_POS_X_YAW)] = next_state[RDE_S(UNCERTAINTY_POS_X_YAW)];
    output[RDE_O(UNCERTAINTY_POS_Y_YAW)] = next_state[RDE_S(UNCERTAINTY_POS_Y_YAW)];
    output[RDE_O(UNCERTAINTY_POS_X)] = next_state[RDE_S(UNCERTAINTY_POS_X)];
    output[RDE_O(UNCERTAINTY_POS_X_Y)] = next_state[RDE_S(UNCERTAINTY_POS_X_Y)];
    output[RDE_O(UNCERTAINTY_POS_Y)] = next_state[RDE_S(UNCERTAINTY_POS_Y)];
    output[RDE_O(TOTAL_VELOCITY)] = fabsf(next_state[RDE_S(VEL_X)]);
  }

  /** racer_dubins_elevation.cu:836-874 */
  __device__ inline void step(float* state, float* next_state, float* state_der, float* control, float* output,
                              float* theta_s, const float t, const float dt)
  {
    // every lane of a rollout works on private copies and stores the same results (see the header)
    float x[STATE_DIM], xn[STATE_DIM], xd[6], u[CONTROL_DIM];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      x[i] = state[i];
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      u[i] = control[i];
    computeParametricDelayDeriv(x, u, xd);
    computeParametricSteerDeriv(x, u, xd);
    computeParametricAccelDeriv(x, u, xd);
    updateState(x, xn, xd, dt);
    computeUncertaintyPropagation(x, xd, xn, dt);
    float roll = x[RDE_S(ROLL)], pitch = x[RDE_S(PITCH)], height;
    computeStaticSettling(xn[RDE_S(YAW)], xn[RDE_S(POS_X)], xn[RDE_S(POS_Y)], roll, pitch, height);
    xn[RDE_S(PITCH)] = pitch;
    xn[RDE_S(ROLL)] = roll;
    mppi::lane_sync();  // BY > 1: nobody overwrites a buffer a sibling lane may still be reading
#pragma unroll
    for (int i = 0; i < 6; i++)
      state_der[i] = xd[i];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      next_state[i] = xn[i];
    output[RDE_O(BASELINK_POS_I_Z)] = height;
    setOutputs(xd, xn, output);
  }
};

class RacerDubinsElevation : public RacerDubinsElevationImpl<RacerDubinsElevation>
{
public:
  RacerDubinsElevation(hipStream_t stream = nullptr) : RacerDubinsElevationImpl<RacerDubinsElevation>(stream)
  {
  }
};

#endif
