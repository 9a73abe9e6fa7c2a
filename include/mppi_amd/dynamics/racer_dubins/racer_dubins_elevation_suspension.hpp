/**
 * RacerDubinsElevationSuspension plugin — the elevation-map RACER Dubins car with the LSTM steering column and a
 * spring / damper suspension in place of the static settling: roll, pitch and the height of the centre of gravity are
 * dynamic states driven by the four wheel forces, which follow from the terrain height and the terrain NORMAL under
 * each wheel (a second, four-channel map).
 *
 * Reference: include/mppi/dynamics/racer_dubins/racer_dubins_elevation_suspension_lstm.cuh:17-66 (parameters, the 24-entry
 * state layout), racer_dubins_elevation_suspension_lstm.cu:199-340 (device computeSimpleSuspensionStep), :342-392 (device
 * step), :394-417 (device updateState), :437-525 (setOutputs).  Kept as the reference has them:
 *   - the front wheels' heading is yaw + S_INDEX(STEER_ANGLE) / -9.1 — the state INDEX (4), not the steering angle
 *     (:253-259);
 *   - rear-right sits at y = +0.737, rear-left at y = -0.737 (:261-266; the front pair is the other way round);
 *   - roll and pitch enter the force geometry as angles (small-angle form), un-wrapped;
 *   - initializeDynamics() fills the t = 0 outputs with the ELEVATION model's setOutputs (the parent's using-declaration,
 *     racer_dubins_elevation_lstm_steering.cuh:28), i.e. NaN wheel forces at t = 0 and real ones from the first step on.
 * Defined here where the reference leaves it open: the three derivatives the wheels add to (vertical acceleration, roll
 * and pitch acceleration) are accumulated in the order FL, FR, BL, BR (the reference uses atomicAdd from up to four
 * threadIdx.y lanes, i.e. no fixed order).
 * One lane per rollout; the eight map lookups of a step (4 heights, 4 normals x 4 channels) are issued together
 * (TwoDTextureHelper::queryTextureAtWorldPoseBatch).  Maps: "elevation_map" / "elevation_map_transform" as for the parent,
 * "normals_map" ({height, width, 4}: nx, ny, nz, unused) with "normals_map_transform" (defaults to the elevation map's).
 */
#ifndef MPPI_AMD_RACER_DUBINS_ELEVATION_SUSPENSION_HPP_
#define MPPI_AMD_RACER_DUBINS_ELEVATION_SUSPENSION_HPP_

#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation_lstm_steering.hpp"

/** reference: racer_dubins_elevation_suspension_lstm.cuh:17-66 */
struct RacerDubinsElevationSuspensionParams : public RacerDubinsElevationParams
{
  enum class WheelIndex : int
  {
    FL = 0,
    FR,
    BL,
    BR,
    NUM_WHEELS,
  };
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
    CG_POS_Z,
    CG_VEL_I_Z,
    ROLL_RATE,
    PITCH_RATE,
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
    FILLER_1,
    NUM_STATES
  };
  float spring_k = 14000.0f;                                           ///< [N / m]
  float drag_c = 1000.0f;                                              ///< [N s / m]
  float mass = 1447.0f;                                                ///< [kg]
  float I_xx = 1.0f / 12 * 1447.0f * 2 * (1.5f * 1.5f);                ///< [kg m^2]
  float I_yy = 1.0f / 12 * 1447.0f * ((1.5f * 1.5f) + (3.0f * 3.0f));  ///< [kg m^2]
  float wheel_radius = 0.32f;                                          ///< [m]
  float c_g[3] = { 2.981f * 0.5f, 0.0f, 0.0f };                        ///< centre of gravity in the body frame
};

/** the per-wheel arithmetic of computeSimpleSuspensionStep (racer_dubins_elevation_suspension_lstm.cu:199-340), shared by the
 *  one-lane form (a loop over the wheels) and the four-lane form (a wheel per replica lane) so that both produce the same bits */
template <class PARAMS_T>
struct RacerSuspensionMath
{
  /** wheel i = FL, FR, BL, BR: body-frame contact point (:249-269) */
  __device__ static inline float wheelBodyX(const int i)
  {
    return i < 2 ? 2.981f : 0.0f;
  }
  __device__ static inline float wheelBodyY(const int i)
  {
    return (i == RDE_W(FL) || i == RDE_W(BR)) ? 0.737f : -0.737f;
  }
  /** spring / damper force of wheel i and its forward / sideways components (magnitudes); height, normal: the terrain under it */
  /** heading of wheel i; front wheels: the reference adds the state INDEX of the steering angle over -9.1 (kept, see the
   *  header) */
  __device__ static inline float wheelYaw(const float yaw, const int i)
  {
    return (i < 2) ? yaw + (float)RDE_S(STEER_ANGLE) / -9.1f : yaw;
  }
  __device__ static inline void wheelForce(const PARAMS_T& p, const float* state, const int i, const float height,
                                           const float nx, const float ny, const float nz, const float sin_wheel_yaw,
                                           const float cos_wheel_yaw, float& up, float& fwd, float& side)
  {
    const float roll = state[RDE_S(ROLL)], pitch = state[RDE_S(PITCH)];
    const float cg_x = wheelBodyX(i) - p.c_g[0], cg_y = wheelBodyY(i) - p.c_g[1];
    const float wheel_pos_z = state[RDE_S(CG_POS_Z)] + roll * cg_y - pitch * cg_x - p.wheel_radius;
    const float wheel_vel_z = state[RDE_S(CG_VEL_I_Z)] + state[RDE_S(ROLL_RATE)] * cg_y - state[RDE_S(PITCH_RATE)] * cg_x;
    const float h_dot = -(state[RDE_S(VEL_X)] * cos_wheel_yaw * nx + state[RDE_S(VEL_X)] * sin_wheel_yaw * ny);
    const float wheel_force = -p.spring_k * (wheel_pos_z - height) - p.drag_c * (wheel_vel_z - h_dot);
    const float fwd_wheel_force = wheel_force / nz * (nx * cos_wheel_yaw + ny * sin_wheel_yaw + nz * (-pitch));
    const float side_wheel_force = wheel_force / nz * (-nx * sin_wheel_yaw + ny * cos_wheel_yaw + nz * roll);
    up = wheel_force;
    fwd = fabsf(fwd_wheel_force);
    side = fabsf(side_wheel_force);
  }
  /** vertical, roll and pitch acceleration from the four wheel forces, summed FL, FR, BL, BR; the three force outputs */
  __device__ static inline void bodyAccelerations(const PARAMS_T& p, const float (&up)[4], const float (&fwd)[4],
                                                  const float (&side)[4], float* state_der, float* output)
  {
    float acc_z = 0.0f, acc_roll = 0.0f, acc_pitch = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
      const float cg_x = wheelBodyX(i) - p.c_g[0], cg_y = wheelBodyY(i) - p.c_g[1];
      acc_z += up[i] / p.mass;
      acc_roll += up[i] * cg_y / p.I_xx;
      acc_pitch += -up[i] * cg_x / p.I_yy;
    }
    state_der[RDE_S(CG_VEL_I_Z)] = acc_z;
    state_der[RDE_S(ROLL_RATE)] = acc_roll;
    state_der[RDE_S(PITCH_RATE)] = acc_pitch;
    output[RDE_O(WHEEL_FORCE_UP_MAX)] = fmaxf(up[0], fmaxf(up[1], fmaxf(up[2], up[3])));
    output[RDE_O(WHEEL_FORCE_FWD_MAX)] = fmaxf(fwd[0], fmaxf(fwd[1], fmaxf(fwd[2], fwd[3])));
    output[RDE_O(WHEEL_FORCE_SIDE_MAX)] = fmaxf(side[0], fmaxf(side[1], fmaxf(side[2], side[3])));
  }
  /** racer_dubins_elevation_suspension_lstm.cu:437-525: as the elevation model's, but the body height comes from the centre
   *  of gravity and the wheel-force outputs are those of the suspension step */
  __device__ static inline void setSuspensionOutputs(const PARAMS_T& p, const float* state_der, const float* next_state,
                                                     float* output)
  {
    output[RDE_O(BASELINK_VEL_B_X)] = next_state[RDE_S(VEL_X)];
    output[RDE_O(BASELINK_VEL_B_Y)] = 0.0f;
    output[RDE_O(BASELINK_POS_I_X)] = next_state[RDE_S(POS_X)];
    output[RDE_O(BASELINK_POS_I_Y)] = next_state[RDE_S(POS_Y)];
    output[RDE_O(BASELINK_POS_I_Z)] = next_state[RDE_S(CG_POS_Z)] - next_state[RDE_S(PITCH)] * (-p.c_g[0]);
    output[RDE_O(PITCH)] = next_state[RDE_S(PITCH)];
    output[RDE_O(ROLL)] = next_state[RDE_S(ROLL)];
    output[RDE_O(YAW)] = next_state[RDE_S(YAW)];
    output[RDE_O(STEER_ANGLE)] = next_state[RDE_S(STEER_ANGLE)];
    output[RDE_O(STEER_ANGLE_RATE)] = next_state[RDE_S(STEER_ANGLE_RATE)];
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
};

template <class CLASS_T, class PARAMS_T = RacerDubinsElevationSuspensionParams>
class RacerDubinsElevationSuspensionImpl : public RacerDubinsElevationLSTMSteeringImpl<CLASS_T, PARAMS_T>
{
public:
  using STEERING = RacerDubinsElevationLSTMSteeringImpl<CLASS_T, PARAMS_T>;
  using ELEVATION = typename STEERING::ELEVATION;
  using StepTrig = typename ELEVATION::StepTrig;
  static constexpr int STATE_DIM = ELEVATION::STATE_DIM, CONTROL_DIM = ELEVATION::CONTROL_DIM,
                       OUTPUT_DIM = ELEVATION::OUTPUT_DIM;
  /** every state in front of STEER_ANGLE_RATE takes the explicit Euler step (racer_dubins_elevation_suspension_lstm.cu:401) */
  static constexpr int NUM_EULER_STATES = RDE_S(STEER_ANGLE_RATE);
  static constexpr int XD = RDE_S(STEER_ANGLE_RATE) + 1;
  /** a step of these models is thousands of instructions: one copy of it per kernel (engine/rollout_kernel.hpp; three
   *  inlined copies overflow the instruction cache: 1177 -> 855 us at K = 16384, T = 100; for the smaller LSTM-steering
   *  step the single site is slightly slower, 688 -> 713 us, and is not used) */
  static constexpr bool SINGLE_STEP_SITE = true;
  using ELEVATION::bodyRotation;
  using ELEVATION::computeParametricAccelDeriv;
  using ELEVATION::computeParametricDelayDeriv;
  using ELEVATION::computeUncertaintyPropagation;
  using ELEVATION::stateTrig;
  using ELEVATION::wheelWorldPoint;
  using STEERING::computeLSTMSteering;
  using STEERING::updateState;

  /** terrain normals (texture 0, four channels), reference: normals_tex_helper_ (…suspension_lstm.cuh:131-146) */
  mppi::texture::TwoDTextureHelper<1, 4> normals_tex_helper_;

  RacerDubinsElevationSuspensionImpl(hipStream_t stream = nullptr) : STEERING(stream)
  {
  }
  static const char* getDynamicsModelName()
  {
    return "RACER Dubins LSTM Steering and Suspension Model";
  }

  using MATH = RacerSuspensionMath<PARAMS_T>;
  __device__ static inline float wheelBodyX(const int i)
  {
    return MATH::wheelBodyX(i);
  }
  __device__ static inline float wheelBodyY(const int i)
  {
    return MATH::wheelBodyY(i);
  }

  /**
   * racer_dubins_elevation_suspension_lstm.cu:199-340: spring / damper force of every wheel from its height above the
   * terrain and its vertical speed relative to the terrain, the resulting vertical, roll and pitch accelerations, and the
   * largest upward / forward / sideways wheel force (outputs).  g: trigonometry of the current state.
   */
  __device__ __forceinline__ void computeSimpleSuspensionStep(const float* state, float* state_der, const StepTrig& g,
                                                     float* output) const
  {
    const PARAMS_T& p = this->S().params_;
    state_der[RDE_S(ROLL)] = state[RDE_S(ROLL_RATE)];
    state_der[RDE_S(PITCH)] = state[RDE_S(PITCH_RATE)];
    state_der[RDE_S(CG_POS_Z)] = state[RDE_S(CG_VEL_I_Z)];
    float M[3][3];
    bodyRotation(g, g.sin_yaw, g.cos_yaw, M);  // Euler2DCM_NWU(roll, pitch, yaw), device branch
    float world[4][3];
#pragma unroll
    for (int i = 0; i < 4; i++)
      wheelWorldPoint(M, wheelBodyX(i), wheelBodyY(i), state[RDE_S(POS_X)], state[RDE_S(POS_Y)], world[i]);
    float height[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    float normal[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
      normal[i][0] = 0.0f;
      normal[i][1] = 0.0f;
      normal[i][2] = 1.0f;
      normal[i][3] = 0.0f;
    }
    if (this->S().tex_helper_.checkTextureUse(0))
    {
      this->S().tex_helper_.template queryTextureAtWorldPoseBatch<4>(0, world, height);
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (!isfinite(height[i]))
          height[i] = state[RDE_S(CG_POS_Z)] - p.wheel_radius;
    }
    if (this->S().normals_tex_helper_.checkTextureUse(0))
    {
      this->S().normals_tex_helper_.template queryTextureAtWorldPoseBatch<4>(0, world, &normal[0][0]);
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (!isfinite(normal[i][0]) || !isfinite(normal[i][1]) || !isfinite(normal[i][2]))
        {
          normal[i][0] = 0.0f;
          normal[i][1] = 0.0f;
          normal[i][2] = 1.0f;
          normal[i][3] = 0.0f;
        }
    }
    float up[4], fwd[4], side[4], sin_wheel_yaw[2], cos_wheel_yaw[2];  // front, rear
    mppi::det::sincos(MATH::wheelYaw(state[RDE_S(YAW)], 0), &sin_wheel_yaw[0], &cos_wheel_yaw[0]);
    mppi::det::sincos(MATH::wheelYaw(state[RDE_S(YAW)], 2), &sin_wheel_yaw[1], &cos_wheel_yaw[1]);
#pragma unroll
    for (int i = 0; i < 4; i++)
      MATH::wheelForce(p, state, i, height[i], normal[i][0], normal[i][1], normal[i][2], sin_wheel_yaw[i >> 1],
                       cos_wheel_yaw[i >> 1], up[i], fwd[i], side[i]);
    MATH::bodyAccelerations(p, up, fwd, side, state_der, output);
  }

  __device__ inline void setSuspensionOutputs(const float* state_der, const float* next_state, float* output) const
  {
    MATH::setSuspensionOutputs(this->S().params_, state_der, next_state, output);
  }

  /** racer_dubins_elevation_suspension_lstm.cu:342-392 */
  __device__ __forceinline__ void step(float* state, float* next_state, float* state_der, float* control, float* output,
                              float* theta_s, const float t, const float dt)
  {
    float x[STATE_DIM], xn[STATE_DIM], xd[XD], u[CONTROL_DIM], wheel_out[OUTPUT_DIM];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
    {
      x[i] = state[i];
      xn[i] = state[i];  // FILLER_1 is carried along (the reference never writes it)
    }
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      u[i] = control[i];
    const StepTrig g = stateTrig(x);
    computeParametricDelayDeriv(x, u, xd);
    computeParametricAccelDeriv(x, u, xd, g);
    computeLSTMSteering(x, u, xd, theta_s);
    computeSimpleSuspensionStep(x, xd, g, wheel_out);
    updateState(x, xn, xd, dt);
    computeUncertaintyPropagation(x, xd, xn, dt, g);
    mppi::lane_sync();
#pragma unroll
    for (int i = 0; i < XD; i++)
      state_der[i] = xd[i];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      next_state[i] = xn[i];
    output[RDE_O(WHEEL_FORCE_UP_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_UP_MAX)];
    output[RDE_O(WHEEL_FORCE_FWD_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_FWD_MAX)];
    output[RDE_O(WHEEL_FORCE_SIDE_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_SIDE_MAX)];
    setSuspensionOutputs(xd, xn, output);
  }
};

class RacerDubinsElevationSuspension : public RacerDubinsElevationSuspensionImpl<RacerDubinsElevationSuspension>
{
public:
  using PARAMS_T = RacerDubinsElevationSuspensionParams;
  RacerDubinsElevationSuspension(hipStream_t stream = nullptr)
    : RacerDubinsElevationSuspensionImpl<RacerDubinsElevationSuspension>(stream)
  {
  }
};

/**
 * Four lanes per rollout (REPLICATED_LANES = 4, lane = column + 16 * replica; racer_dubins_elevation.hpp has the contract):
 * replica r evaluates one angle of each of the two trigonometry passes, hidden unit r of the steering LSTM and five neurons
 * of its output network (lstm_quad.hpp), wheel r of the suspension (terrain height, terrain normal, spring / damper force)
 * and row r of the covariance update.  Per value the arithmetic is the one-lane form's (RacerSuspensionMath, the wheel
 * forces summed FL, FR, BL, BR on every replica), so both agree with the oracle bit for bit.  Default network shape only.
 * The pieces of the step are members of their own so that the complete RACER model (racer_dubins_elevation_lstm_unc.hpp)
 * composes its step from them.
 */
template <class CLASS_T, class PARAMS_T = RacerDubinsElevationSuspensionParams, class STEER_NET = mppi::LSTMQuadRows<4, 20, 1>>
class RacerDubinsElevationSuspensionQuadImpl : public RacerDubinsElevationImpl<CLASS_T, PARAMS_T>
{
public:
  using ELEVATION = RacerDubinsElevationImpl<CLASS_T, PARAMS_T>;
  using StepTrig = typename ELEVATION::StepTrig;
  using MATH = RacerSuspensionMath<PARAMS_T>;
  static constexpr int STATE_DIM = ELEVATION::STATE_DIM, CONTROL_DIM = ELEVATION::CONTROL_DIM,
                       OUTPUT_DIM = ELEVATION::OUTPUT_DIM;
  static constexpr int REPLICATED_LANES = 4;
  static constexpr int NUM_EULER_STATES = RDE_S(STEER_ANGLE_RATE);
  static constexpr int XD = RDE_S(STEER_ANGLE_RATE) + 1;
  using ELEVATION::allReplicas;
  using ELEVATION::bodyRotation;
  using ELEVATION::fromReplica;
  using ELEVATION::pick4;
  using ELEVATION::setOutputs;
  using ELEVATION::wheelWorldPoint;

  mppi::texture::TwoDTextureHelper<1, 4> normals_tex_helper_;
  const float* lstm_d_ = nullptr;
  const float* fnn_d_ = nullptr;
  STEER_NET net_ = {};  ///< this lane's share of the steering network and its recurrent state

  RacerDubinsElevationSuspensionQuadImpl(hipStream_t stream = nullptr) : ELEVATION(stream)
  {
  }
  /** parameters, control limits, maps and the steering network of the one-lane model */
  template <class OTHER>
  void copyFrom(const OTHER& other)
  {
    this->params_ = other.params_;
    for (int i = 0; i < CONTROL_DIM; i++)
    {
      this->control_rngs_[i] = other.control_rngs_[i];
      this->control_deadband_[i] = other.control_deadband_[i];
      this->zero_control_[i] = other.zero_control_[i];
    }
    this->tex_helper_ = other.tex_helper_;
    normals_tex_helper_ = other.normals_tex_helper_;
    lstm_d_ = other.lstm_.weights_d_;
    fnn_d_ = other.lstm_.output_nn_.theta_d_;
  }

  __device__ static inline int replica()
  {
    return (int)(threadIdx.x & 63) >> 4;
  }

  __device__ __forceinline__ void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                                     float dt)
  {
    this->stageStepSource(theta_s);  // the read-only members' copy in LDS first: setOutputs below reads through S()
    net_.load(replica(), lstm_d_, fnn_d_);
    output[RDE_O(BASELINK_POS_I_Z)] = 0.0f;
    output[RDE_O(FILLER_1)] = 0.0f;
    setOutputs(state, state, output);
  }

  /** the trigonometry of the current state in two passes: 1) replica 0 yaw, 1 wrapped steering angle, 2 raw steering angle,
   *  3 pitch; 2) replica 0 roll, 1 heading of the front wheels, 2 and 3 heading of the rear wheels (returned for THIS
   *  replica's wheel) */
  __device__ __forceinline__ void quadTrig(const float* x, const int rep, StepTrig& g, float& sin_wheel_yaw,
                                           float& cos_wheel_yaw) const
  {
    const PARAMS_T& p = this->S().params_;
    {
      const float delta = x[RDE_S(STEER_ANGLE)] / p.steer_angle_scale;
      const float raw = pick4(rep, x[RDE_S(YAW)], delta, delta, x[RDE_S(PITCH)]);
      const float wrapped = angle_utils::normalizeAngle(raw);
      float s, c, sa[4], ca[4];
      mppi::det::sincos(rep == 2 ? raw : wrapped, &s, &c);
      allReplicas(s, sa);
      allReplicas(c, ca);
      g.sin_yaw = sa[0];
      g.cos_yaw = ca[0];
      g.tan_steer_n = sa[1] / ca[1];
      g.tan_delta = sa[2] / ca[2];
      g.cos_delta = ca[2];
      g.sin_pitch = sa[3];
      g.cos_pitch = ca[3];
    }
    {
      const float wheel_yaw = MATH::wheelYaw(x[RDE_S(YAW)], rep == 1 ? 0 : 2);
      float s, c;
      mppi::det::sincos(rep == 0 ? angle_utils::normalizeAngle(x[RDE_S(ROLL)]) : wheel_yaw, &s, &c);
      g.sin_roll = fromReplica(s, 0);
      g.cos_roll = fromReplica(c, 0);
      const int src = rep < 2 ? 1 : 2;  // the heading of this replica's wheel
      sin_wheel_yaw = fromReplica(s, src);
      cos_wheel_yaw = fromReplica(c, src);
    }
  }

  /** racer_dubins_elevation_lstm_steering.cu:131-167, the network shared out over the replicas */
  __device__ __forceinline__ void quadSteering(const float* x, const float* u, float* xd)
  {
    const PARAMS_T& p = this->S().params_;
    const float steer = x[RDE_S(STEER_ANGLE)], rate = x[RDE_S(STEER_ANGLE_RATE)];
    const float parametric_accel = (u[RDE_C(STEER_CMD)] * p.steer_command_angle_scale - steer) * p.steering_constant;
    float rate_dot = fmaxf(fminf((parametric_accel - rate) * p.steer_accel_constant - rate * p.steer_accel_drag_constant,
                                 p.max_steer_rate),
                           -p.max_steer_rate);
    const float input[4] = { steer * 0.2f, rate * 0.2f, u[RDE_C(STEER_CMD)], rate_dot * 0.2f };
    float out[1] = { 0.0f };
    net_.forward(this->S().fnn_d_, input, out);
    rate_dot += out[0] * 5.0f;
    xd[RDE_S(STEER_ANGLE_RATE)] = rate_dot;
    xd[RDE_S(STEER_ANGLE)] = rate;
  }

  /** racer_dubins_elevation_suspension_lstm.cu:199-340 with wheel `rep` on this lane; fills the suspension entries of xd and
   *  the three wheel-force outputs */
  __device__ __forceinline__ void quadSuspension(const float* x, const StepTrig& g, const int rep, const float sin_wheel_yaw,
                                                 const float cos_wheel_yaw, float* xd, float* wheel_out) const
  {
    const PARAMS_T& p = this->S().params_;
    xd[RDE_S(ROLL)] = x[RDE_S(ROLL_RATE)];
    xd[RDE_S(PITCH)] = x[RDE_S(PITCH_RATE)];
    xd[RDE_S(CG_POS_Z)] = x[RDE_S(CG_VEL_I_Z)];
    float M[3][3], world[3];
    bodyRotation(g, g.sin_yaw, g.cos_yaw, M);
    wheelWorldPoint(M, MATH::wheelBodyX(rep), MATH::wheelBodyY(rep), x[RDE_S(POS_X)], x[RDE_S(POS_Y)], world);
    float height = 0.0f, normal[4] = { 0.0f, 0.0f, 1.0f, 0.0f };
    if (this->S().tex_helper_.checkTextureUse(0))
    {
      this->S().tex_helper_.queryTextureAtWorldPose(0, world, &height);
      if (!isfinite(height))
        height = x[RDE_S(CG_POS_Z)] - p.wheel_radius;
    }
    if (this->S().normals_tex_helper_.checkTextureUse(0))
    {
      this->S().normals_tex_helper_.queryTextureAtWorldPose(0, world, normal);
      if (!isfinite(normal[0]) || !isfinite(normal[1]) || !isfinite(normal[2]))
      {
        normal[0] = 0.0f;
        normal[1] = 0.0f;
        normal[2] = 1.0f;
        normal[3] = 0.0f;
      }
    }
    float up_own, fwd_own, side_own, up[4], fwd[4], side[4];
    MATH::wheelForce(p, x, rep, height, normal[0], normal[1], normal[2], sin_wheel_yaw, cos_wheel_yaw, up_own, fwd_own,
                     side_own);
    allReplicas(up_own, up);
    allReplicas(fwd_own, fwd);
    allReplicas(side_own, side);
    MATH::bodyAccelerations(p, up, fwd, side, xd, wheel_out);
  }

  __device__ __forceinline__ void step(float* state, float* next_state, float* state_der, float* control, float* output,
                                       float* theta_s, const float t, const float dt)
  {
    const PARAMS_T& p = this->S().params_;
    const int rep = replica();
    float x[STATE_DIM], xn[STATE_DIM], xd[XD], u[CONTROL_DIM], wheel_out[OUTPUT_DIM];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
    {
      x[i] = state[i];
      xn[i] = state[i];  // FILLER_1 is carried along
    }
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      u[i] = control[i];
    StepTrig g;
    float sin_wheel_yaw, cos_wheel_yaw;
    quadTrig(x, rep, g, sin_wheel_yaw, cos_wheel_yaw);
    this->computeParametricDelayDeriv(x, u, xd);
    this->computeParametricAccelDeriv(x, u, xd, g);
    quadSteering(x, u, xd);
    quadSuspension(x, g, rep, sin_wheel_yaw, cos_wheel_yaw, xd, wheel_out);
    // Euler step of the twelve integrated states and of the steering rate
    this->updateState(x, xn, xd, dt);
    xn[RDE_S(STEER_ANGLE_RATE)] = x[RDE_S(STEER_ANGLE_RATE)] + xd[RDE_S(STEER_ANGLE_RATE)] * dt;
    this->covarianceFourLanes(x, xd, g, dt, rep, xn);
#pragma unroll
    for (int i = 0; i < XD; i++)
      state_der[i] = xd[i];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      next_state[i] = xn[i];
    output[RDE_O(WHEEL_FORCE_UP_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_UP_MAX)];
    output[RDE_O(WHEEL_FORCE_FWD_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_FWD_MAX)];
    output[RDE_O(WHEEL_FORCE_SIDE_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_SIDE_MAX)];
    MATH::setSuspensionOutputs(p, xd, xn, output);
  }
};

class RacerDubinsElevationSuspensionQuad : public RacerDubinsElevationSuspensionQuadImpl<RacerDubinsElevationSuspensionQuad>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  /** S() of RacerDubinsElevationImpl: 0 = the object itself.  1 (argument block, s_load) and 2 (copy in LDS, ds_read) remove
   *  most of the spilled-SGPR reads of the step loop and are SLOWER (profiles/r06_step_source_ab.json) — A/B: -DMPPI_STEP_SOURCE_QUAD=1|2 */
  static constexpr int MPPI_STEP_SOURCE = MPPI_STEP_SOURCE_QUAD;
  /** helper waves of the role-pipelined Robust kernel for this model (engine/rmppi_pipeline_kernel.hpp): one sampler and ONE cost
   *  wave per system = 11 waves per block = 3 per SIMD = 168 VGPRs per lane instead of 15 waves / 128 (K = 16384, T = 100, us per
   *  launch: elevation 356 -> 352, LSTM steering 601 -> 489, suspension 1013 -> 615; two cost waves per system = 13 waves change
   *  nothing; with 168 VGPRs this model no longer spills any, and a RMPPI_PIPELINE_FORM reading its parameters from the argument
   *  block — what the complete model's Robust kernel gains another 22 % from — makes no difference here: 618 us —
   *  profiles/r06_robust_racer_ab.json) */
  static constexpr int MPPI_RMPPI_PIPE_SAMPLERS = 1;
  static constexpr int MPPI_RMPPI_PIPE_COSTS = 1;
  using PARAMS_T = RacerDubinsElevationSuspensionParams;
  RacerDubinsElevationSuspensionQuad(const RacerDubinsElevationSuspension& other)
    : RacerDubinsElevationSuspensionQuadImpl<RacerDubinsElevationSuspensionQuad>(other.stream_)
  {
    copyFrom(other);
  }
};

#endif
