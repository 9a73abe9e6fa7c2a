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
#include "mppi_amd/engine/kernarg_view.hpp"
#include "mppi_amd/utils/texture_helpers/two_d_texture_helper.hpp"

#ifndef U_INDEX
#define U_IND_CLASS(CLASS, enum_val) E_INDEX(CLASS::UncertaintyIndex, enum_val)
#define U_INDEX(enum_val) U_IND_CLASS(PARENT_CLASS::DYN_PARAMS_T, enum_val)
#endif
/* index shorthands on PARAMS_T — the parameter struct of the class they are used in (a template parameter in the *Impl
 * classes, a member alias in the concrete ones): the RACER subclasses re-number the states */
#define RDE_S(enum_val) S_IND_CLASS(PARAMS_T, enum_val)
#define RDE_C(enum_val) C_IND_CLASS(PARAMS_T, enum_val)
#define RDE_O(enum_val) O_IND_CLASS(PARAMS_T, enum_val)
#define RDE_U(enum_val) U_IND_CLASS(PARAMS_T, enum_val)
#define RDE_W(enum_val) E_INDEX(PARAMS_T::WheelIndex, enum_val)

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
template <class CLASS_T, class PARAMS_T = RacerDubinsElevationParams>
class RacerDubinsElevationImpl : public MPPI_internal::Dynamics<CLASS_T, PARAMS_T>
{
public:
  using PARENT_CLASS = MPPI_internal::Dynamics<CLASS_T, PARAMS_T>;
  static constexpr int STATE_DIM = RDE_S(NUM_STATES);
  static constexpr int CONTROL_DIM = RDE_C(NUM_CONTROLS);
  static constexpr int OUTPUT_DIM = RDE_O(NUM_OUTPUTS);
  static const int UNCERTAINTY_DIM = U_IND_CLASS(PARAMS_T, NUM_UNCERTAINTIES);
  /** states advanced by the explicit Euler step of updateState(): the first six here, more in the suspension models */
  static constexpr int NUM_EULER_STATES = 6;
  static constexpr int UD = UNCERTAINTY_DIM;

  /** the elevation map (texture 0), reference: tex_helper_ of racer_dubins_elevation.cuh:88-96 */
  mppi::texture::TwoDTextureHelper<1, 1> tex_helper_;

  /**
   * S(): where the per-step device methods READ this object's read-only members (params_, the maps' frames, blob pointers)
   * from.  Three sources exist; the product uses the first — the other two are the round-6 experiment the round-5 review asked
   * for ("get kernel arguments out of spilled SGPRs"), kept behind -DMPPI_STEP_SOURCE_QUAD=1|2 with their measurements.
   *
   * The question.  Written `this->params_.x`, every parameter is a loop-invariant scalar load: the compiler hoists all of them
   * in front of the step loop and keeps them in SGPRs across it.  A step of these models reads a few hundred parameters, a wave
   * has ~100 SGPRs: the code objects carry 300-460 SPILLED SGPRs (to VGPR lanes; one v_readlane_b32 + hazard s_nops per use —
   * 680 + 456 of the 7740 instructions of a pair of steps of the complete model's dynamics wave).  Read through a pointer the
   * optimiser cannot see through, or from LDS, the loads stay inside the method that asked for them: 7740 -> 6680
   * instructions, v_readlane 680 -> 204 (argument block) / 177 (LDS).
   *
   * The answer (MI355X, K = 16384, T = 100, us per rollout launch; profiles/r06_step_source_ab.json):
   *                          this (0)   argument block (1)   LDS copy (2)
   *     elevation, 4 lanes     221.4         254.8               260.9
   *     + steering LSTM        305.4         329.9               328.4
   *     + suspension           393.9         412.3               416.7
   *     complete model         811.0         860.9               865.0
   *     Robust complete model 3398.9        2953.1              2979.4      (the one kernel that spills VGPRs to scratch;
   *                                                                          its RMPPI_PIPELINE_FORM uses source 1: 1565 us)
   * The 14 % fewer instructions run 6-18 % LONGER: the dynamics waves are not short of issue slots — a lone wave per SIMD
   * issues a dependent instruction every ~7 cycles and an independent one every 4 (tools/ubench/exec_width.hip), the step is a
   * long dependent chain, and the v_readlane / s_nop pairs sit in slots that were idle anyway — while every s_load / ds_read
   * puts its memory latency INTO the chain (SMEM results return out of order: each use waits for lgkmcnt(0), LDS traffic
   * included).  Spilled SGPRs are the cheapest place those parameters can live.  Only where VGPRs spill to scratch memory
   * (the Robust kernel of the complete model, 960 threads per block = 128 VGPRs per lane) does taking the parameters out of
   * the register file pay.
   *
   * Why a function and not a pointer member (both tried): members that hold PER-LANE STATE (a network's parameters in this
   * lane's registers, loaded by initializeDynamics) must stay the object's own, so the kernels cannot run the methods on a view
   * of the argument block (the matrix-core and four-lane models lose their weights); and a `step_src_` member set by the
   * kernels makes the by-value argument a private copy that cannot be split into registers where a method indexes a member
   * array with a run-time index — the whole object moved to scratch (12 -> 1072 B in the finalize and model-step kernels).
   * Coupling of sources 1 and 2: the object is the running kernel's argument at byte STEP_SOURCE_KERNARG_OFFSET (0: the dynamics
   * object is the first argument of every kernel of the engine) / the dynamics region starts the kernel's dynamic LDS
   * (theta_s_shared = smem_raw everywhere).  On the host S() is the object.
   */
  static constexpr size_t STEP_SOURCE_KERNARG_OFFSET = 0;
  /** which of the three a class gets: `static constexpr int MPPI_STEP_SOURCE = ...` in the (most derived) class —
   *    0  the object itself (`this->params_`) — the default, and what the product builds;
   *    1  the argument block behind the opaque pointer (engine/kernarg_view.hpp): s_load next to the use;
   *    2  a copy of the object in LDS, at the start of the class's block-shared (Grd) region = the start of the kernel's dynamic
   *       LDS: staged once per block by initializeDynamics (stageStepSource), read with ds_read (uniform address: a broadcast). */
  template <class T, class = void>
  struct step_source_of : std::integral_constant<int, 0>
  {
  };
  template <class T>
  struct step_source_of<T, std::void_t<decltype(T::MPPI_STEP_SOURCE)>> : std::integral_constant<int, T::MPPI_STEP_SOURCE>
  {
  };
  static constexpr int stepSourceBytes()  // (a function: its body is only instantiated once CLASS_T is complete)
  {
    return (int)((sizeof(CLASS_T) + 15) / 16 * 16);
  }
  __host__ __device__ __forceinline__ const CLASS_T& S() const
  {
#if defined(__HIP_DEVICE_COMPILE__) && MPPI_KERNARG_RELOAD
    if constexpr (step_source_of<CLASS_T>::value == 2)
    {
      extern __shared__ __attribute__((aligned(16))) char mppi_step_source_lds[];
      return *reinterpret_cast<const CLASS_T*>(mppi_step_source_lds);
    }
    else if constexpr (step_source_of<CLASS_T>::value == 1)
      return *mppi::kernels::kernargObject<CLASS_T>(mppi::kernels::kernargBase(), STEP_SOURCE_KERNARG_OFFSET);
    else
      return *static_cast<const CLASS_T*>(this);
#else
    return *static_cast<const CLASS_T*>(this);
#endif
  }
  /** MPPI_STEP_SOURCE == 2: every thread of the block copies its share of the kernel's argument block into the class's Grd
   *  region (theta_s), then a block barrier.  FIRST statement of every initializeDynamics of the hierarchy (all threads of a
   *  block reach initializeDynamics: the plugin contract). */
  __device__ __forceinline__ void stageStepSource(float* theta_s) const
  {
#if defined(__HIP_DEVICE_COMPILE__) && MPPI_KERNARG_RELOAD
    if constexpr (step_source_of<CLASS_T>::value == 2)
    {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(mppi::kernels::kernargObject<CLASS_T>(
          (mppi::kernels::kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr(), STEP_SOURCE_KERNARG_OFFSET));
      uint32_t* dst = reinterpret_cast<uint32_t*>(theta_s);
      const int nthreads = (int)(blockDim.x * blockDim.y * blockDim.z);
      const int tid = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
      for (int i = tid; i < (int)(sizeof(CLASS_T) / 4); i += nthreads)
        dst[i] = src[i];
      __syncthreads();  // initializeDynamics itself reads through S() right away (setOutputs); block-uniform, as the contract allows
    }
#endif
  }
  /** the staged copy is this class's block-shared request (the four-lane forms ask for nothing else) */
  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return (MPPI_KERNARG_RELOAD && step_source_of<CLASS_T>::value == 2) ? stepSourceBytes() : 0;
  }

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
  __device__ __forceinline__ void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    stageStepSource(theta_s);  // MPPI_STEP_SOURCE == 2: the read-only members' copy in LDS (see S())
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

  /**
   * HOST: ColoredMPPI's state leash for the RACER models (reference: RacerDubinsImpl::enforceLeash,
   * dynamics/racer_dubins/racer_dubins.cu:176-240): the position error is rotated into the body frame of the true state,
   * clamped there to +-leash[POS_X] / +-leash[POS_Y] and rotated back; the yaw error is the shortest angular distance
   * and the leashed yaw is wrapped; every other state follows the base rule (dynamics.cuh:448-466).  Plain libm on the
   * host, as the reference.
   */
  void enforceLeash(const float* state_true, const float* state_nominal, const float* leash_values, float* state_output) const
  {
    constexpr int YAW = RDE_S(YAW), POS_X = RDE_S(POS_X), POS_Y = RDE_S(POS_Y);
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

  /** a[index] of a three-entry parameter table by selects (a run-time index into a kernel-argument array would move the
   *  table into scratch memory) */
  __device__ static inline float pick3(const float (&a)[3], const int index)
  {
    const float a0 = a[0], a1 = a[1], a2 = a[2];  // three scalar loads, then selects (not a select of three addresses)
    return index == 0 ? a0 : (index == 1 ? a1 : a2);
  }

  /** racer_dubins.cu:281-293 (brake lag, faster on release) and :295-305 (steering lag) */
  __device__ inline void computeParametricDelayDeriv(const float* state, const float* control, float* state_der) const
  {
    const PARAMS_T& p = this->S().params_;
    const bool enable_brake = control[RDE_C(THROTTLE_BRAKE)] < 0.0f;
    const float brake_error = (enable_brake * -control[RDE_C(THROTTLE_BRAKE)] - state[RDE_S(BRAKE_STATE)]);
    state_der[RDE_S(BRAKE_STATE)] = fminf(fmaxf((brake_error > 0) * brake_error * p.brake_delay_constant +
                                                      (brake_error < 0) * brake_error * p.brake_delay_constant_neg,
                                                  -p.max_brake_rate_neg),
                                            p.max_brake_rate_pos);
  }
  __device__ inline void computeParametricSteerDeriv(const float* state, const float* control, float* state_der) const
  {
    const PARAMS_T& p = this->S().params_;
    state_der[RDE_S(STEER_ANGLE)] =
        fmaxf(fminf((control[RDE_C(STEER_CMD)] * p.steer_command_angle_scale - state[RDE_S(STEER_ANGLE)]) *
                        p.steering_constant,
                    p.max_steer_rate),
              -p.max_steer_rate);
  }

  /**
   * Every trigonometric value a step takes from the CURRENT state, evaluated once (the reference re-evaluates the same
   * expressions in computeParametricAccelDeriv, computeUncertaintyJacobian, computeQ and Euler2DCM_NWU).  det::tan is
   * sin / cos of one det::sincos, det::sin and det::cos are its two results, so sharing them changes no bit.
   */
  struct StepTrig
  {
    float sin_yaw, cos_yaw;      ///< of normalizeAngle(yaw)
    float tan_steer_n;           ///< tan(normalizeAngle(steer / steer_angle_scale)): the yaw rate
    float tan_delta, cos_delta;  ///< of steer / steer_angle_scale, not wrapped: Jacobian and process noise
    float sin_pitch, cos_pitch;  ///< of normalizeAngle(pitch): gravity, body rotation
    float sin_roll, cos_roll;    ///< of normalizeAngle(roll): side force, body rotation
  };
  __device__ inline StepTrig stateTrig(const float* state) const
  {
    StepTrig g;
    const float delta = state[RDE_S(STEER_ANGLE)] / this->S().params_.steer_angle_scale;
    float s, c;
    mppi::det::sincos(angle_utils::normalizeAngle(state[RDE_S(YAW)]), &g.sin_yaw, &g.cos_yaw);
    mppi::det::sincos(angle_utils::normalizeAngle(delta), &s, &c);
    g.tan_steer_n = s / c;
    mppi::det::sincos(delta, &s, &c);
    g.tan_delta = s / c;
    g.cos_delta = c;
    mppi::det::sincos(angle_utils::normalizeAngle(state[RDE_S(PITCH)]), &g.sin_pitch, &g.cos_pitch);
    mppi::det::sincos(angle_utils::normalizeAngle(state[RDE_S(ROLL)]), &g.sin_roll, &g.cos_roll);
    return g;
  }

  /** racer_dubins_elevation.cu:753-798: longitudinal acceleration per speed regime, clamp, gravity along the pitch; yaw
   *  rate of the bicycle; position rate */
  __device__ inline void computeParametricAccelDeriv(const float* state, const float* control, float* state_der,
                                                     const StepTrig& g) const
  {
    const PARAMS_T& p = this->S().params_;
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
      ax -= p.gravity * g.sin_pitch;
    }
    state_der[RDE_S(VEL_X)] = ax;
    state_der[RDE_S(YAW)] = (vx / p.wheel_base) * g.tan_steer_n;
    state_der[RDE_S(POS_X)] = vx * g.cos_yaw;
    state_der[RDE_S(POS_Y)] = vx * g.sin_yaw;
  }

  /** racer_dubins_elevation.cu:800-834: Euler step of the six integrated states, as in RacerDubins */
  __device__ inline void updateState(const float* state, float* next_state, const float* state_der, const float dt) const
  {
    const PARAMS_T& p = this->S().params_;
#pragma unroll
    for (int i = 0; i < CLASS_T::NUM_EULER_STATES; i++)
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
  __device__ inline void computeUncertaintyJacobian(const float* state, const StepTrig& g, float* A) const
  {
    const PARAMS_T& p = this->S().params_;
    const float vx = state[RDE_S(VEL_X)];
    const float sin_yaw = g.sin_yaw, cos_yaw = g.cos_yaw;
    const float tan_steer_angle = g.tan_delta;
    const float cos_2_delta = g.cos_delta * g.cos_delta;
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
  __device__ inline void computeQ(const float* state, const float* state_der, const StepTrig& g, float* Q) const
  {
    const PARAMS_T& p = this->S().params_;
    const float abs_vx = fabsf(state[RDE_S(VEL_X)]);
    const float abs_acc_x = fabsf(state_der[RDE_S(VEL_X)]);
    const float delta = state[RDE_S(STEER_ANGLE)] / p.steer_angle_scale;
    const float sin_yaw = g.sin_yaw, cos_yaw = g.cos_yaw;
    const float tan_steer_angle = g.tan_delta;
    const float sin_roll = g.sin_roll;
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
  __device__ __forceinline__ void computeUncertaintyPropagation(const float* state, const float* state_der, float* next_state,
                                                       const float dt, const StepTrig& g) const
  {
    computeUncertaintyPropagation(state, state_der, next_state, dt, g,
                                  [this, state, state_der, &g](float* Q) { computeQ(state, state_der, g, Q); });
  }
  /** the same with the process noise supplied by the caller: process_noise(Q) fills the sixteen entries (the LSTM
   *  uncertainty model has a network produce them, racer_dubins_elevation_lstm_unc.cu:300-494) */
  template <class QFN>
  __device__ __forceinline__ void computeUncertaintyPropagation(const float* state, const float* state_der, float* next_state,
                                                       const float dt, const StepTrig& g, QFN&& process_noise) const
  {
    float A[UD * UD], Sigma_a[UD * UD], Sigma_b[UD * UD];
    computeUncertaintyJacobian(state, g, A);
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
    process_noise(Sigma_b);
#pragma unroll
    for (int i = 0; i < UD * UD; i++)
      Sigma_a[i] += Sigma_b[i] * dt;
    uncertaintyMatrixToState(Sigma_a, next_state);
  }

  /** Euler2DCM_NWU, device branch (math_utils.h:457-482): roll / pitch of the CURRENT state (g), yaw of the next one */
  __device__ static inline void bodyRotation(const StepTrig& g, const float sin_psi, const float cos_psi, float (&M)[3][3])
  {
    const float sin_phi = g.sin_roll, cos_phi = g.cos_roll, sin_theta = g.sin_pitch, cos_theta = g.cos_pitch;
    M[0][0] = cos_theta * cos_psi;
    M[0][1] = sin_phi * sin_theta * cos_psi - cos_phi * sin_psi;
    M[0][2] = cos_phi * sin_theta * cos_psi + sin_phi * sin_psi;
    M[1][0] = cos_theta * sin_psi;
    M[1][1] = sin_phi * sin_theta * sin_psi + cos_phi * cos_psi;
    M[1][2] = cos_phi * sin_theta * sin_psi - sin_phi * cos_psi;
    M[2][0] = -sin_theta;
    M[2][1] = sin_phi * cos_theta;
    M[2][2] = cos_phi * cos_theta;
  }
  /** bodyOffsetToWorldPoseEuler (math_utils.h:585-597) of a wheel contact point (off_x, off_y, 0): RotatePointByDCM is
   *  gemm1<3, 3, 1>, three terms accumulated from zero, then the body pose (x, y, 0) */
  __device__ static inline void wheelWorldPoint(const float (&M)[3][3], const float off_x, const float off_y, const float x,
                                                const float y, float (&world)[3])
  {
    const float body_pose[3] = { x, y, 0.0f };
#pragma unroll
    for (int r = 0; r < 3; r++)
    {
      float acc = 0.0f;
      acc += M[r][0] * off_x;
      acc += M[r][1] * off_y;
      acc += M[r][2] * 0.0f;
      world[r] = acc + body_pose[r];
    }
  }
  /** which = 0 front roll, 1 rear roll, 2 left pitch, 3 right pitch: the clamped height difference over the track width
   *  (2 x 0.737 m) or the wheel base (2.981 m) whose asin is that angle; h = heights front-left, front-right, rear-left,
   *  rear-right (racer_dubins.cu:389-404) */
  __device__ static inline float settlingSine(const int which, const float (&h)[4])
  {
    const bool roll_pair = which < 2;
    const float a = which == 0 ? h[0] : (which == 1 ? h[2] : (which == 2 ? h[2] : h[3]));
    const float b = which == 0 ? h[1] : (which == 1 ? h[3] : (which == 2 ? h[0] : h[1]));
    const float lim = roll_pair ? 0.736f * 2.0f : 2.98f;
    const float den = roll_pair ? 0.737f * 2.0f : 2.981f;
    float diff = a - b;
    diff = fmaxf(fminf(diff, lim), -lim);
    return diff / den;
  }
  /** racer_dubins.cu:393-424: roll, pitch, height from the four angles and the rear heights; non-finite results replaced */
  __device__ static inline void settle(const float (&angle)[4], const float (&h)[4], float& roll, float& pitch, float& height)
  {
    roll = (angle[0] + angle[1]) / 2.0f;
    pitch = (angle[2] + angle[3]) / 2.0f;
    height = (h[2] + h[3]) / 2.0f;
    // 2 pi: a rotation that accidentally uses such a value is the identity
    const float two_pi = 2.0f * 3.14159274101257324219f;
    if (!isfinite(roll) || fabsf(roll) > 3.14159274101257324219f)
      roll = two_pi;
    if (!isfinite(pitch) || fabsf(pitch) > 3.14159274101257324219f)
      pitch = two_pi;
    if (!isfinite(height))
      height = 0.0f;
  }
  /** front left, front right, rear left, rear right (racer_dubins.cu:363-366) */
  __device__ static inline float wheelOffsetX(const int w)
  {
    return w < 2 ? 2.981f : 0.0f;
  }
  __device__ static inline float wheelOffsetY(const int w)
  {
    return (w & 1) ? -0.737f : 0.737f;
  }

  /**
   * RACER::computeStaticSettling (racer_dubins.cu:358-434): the four wheel contact points of the body at (x, y, yaw) with
   * the CURRENT roll and pitch are looked up in the elevation map; roll and pitch for the next state follow from the
   * height differences across the track width and the wheel base.
   */
  __device__ __forceinline__ void computeStaticSettling(const float yaw, const float x, const float y, const StepTrig& g, float& roll,
                                               float& pitch, float& height) const
  {
    height = 0.0f;
    if (!this->S().tex_helper_.checkTextureUse(0))
    {
      roll = 0.0f;
      pitch = 0.0f;
      return;
    }
    float sin_psi, cos_psi, M[3][3];
    mppi::det::sincos(angle_utils::normalizeAngle(yaw), &sin_psi, &cos_psi);
    bodyRotation(g, sin_psi, cos_psi, M);
    float h[4], world[4][3];
#pragma unroll
    for (int w = 0; w < 4; w++)
      wheelWorldPoint(M, wheelOffsetX(w), wheelOffsetY(w), x, y, world[w]);
    this->S().tex_helper_.template queryTextureAtWorldPoseBatch<4>(0, world, h);  // the sixteen loads of the four wheels in flight together
    float angle[4];
#pragma unroll
    for (int w = 0; w < 4; w++)
      angle[w] = mppi::det::asin(settlingSine(w, h));
    settle(angle, h, roll, pitch, height);
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

  /* ---------------------------------------------------------------------------------------------------------------
   * Four lanes per rollout (REPLICATED_LANES = 4; see RacerDubinsElevationQuad below): lane = column + 16 * replica.
   * ------------------------------------------------------------------------------------------------------------- */
  /** value of replica `src` of this lane's rollout */
  __device__ static inline float fromReplica(const float v, const int src)
  {
    return __shfl(v, (int)(threadIdx.x & 15) + 16 * src, 64);
  }
  /** out[r] = v of replica r */
  __device__ static inline void allReplicas(const float v, float (&out)[4])
  {
#pragma unroll
    for (int r = 0; r < 4; r++)
      out[r] = fromReplica(v, r);
  }
  __device__ static inline float pick4(const int r, const float a, const float b, const float c, const float d)
  {
    return r == 0 ? a : (r == 1 ? b : (r == 2 ? c : d));
  }
  /** four lanes per rollout: replica `rep` forms row `rep` of (I + A dt) Sigma (I + A dt)^T + Q dt (the sums in the order of
   *  computeUncertaintyPropagation), the rows are exchanged and written to next_state */
  template <class QFN>
  __device__ __forceinline__ void covarianceFourLanes(const float* x, const float* xd, const StepTrig& g, const float dt,
                                                      const int rep, float* xn, QFN&& process_noise) const
  {
    float A[UD * UD], Sigma[UD * UD], Q[UD * UD], Ar[UD], Sb[UD], row[UD];
    computeUncertaintyJacobian(x, g, A);
    uncertaintyStateToMatrix(x, Sigma);
#pragma unroll
    for (int i = 0; i < UD * UD; i++)
      A[i] = (i % (UD + 1) == 0) + A[i] * dt;
#pragma unroll
    for (int k = 0; k < UD; k++)
      Ar[k] = pick4(rep, A[cm(0, k)], A[cm(1, k)], A[cm(2, k)], A[cm(3, k)]);
#pragma unroll
    for (int n = 0; n < UD; n++)
    {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < UD; k++)
        acc += Ar[k] * Sigma[cm(k, n)];
      Sb[n] = acc;
    }
    process_noise(Q);
#pragma unroll
    for (int n = 0; n < UD; n++)
    {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < UD; k++)
        acc += Sb[k] * A[cm(n, k)];
      acc += pick4(rep, Q[cm(0, n)], Q[cm(1, n)], Q[cm(2, n)], Q[cm(3, n)]) * dt;
      row[n] = acc;
    }
#pragma unroll
    for (int m = 0; m < UD; m++)
#pragma unroll
      for (int n = 0; n < UD; n++)
        Sigma[cm(m, n)] = fromReplica(row[n], m);
    uncertaintyMatrixToState(Sigma, xn);
  }

  __device__ __forceinline__ void covarianceFourLanes(const float* x, const float* xd, const StepTrig& g, const float dt,
                                                      const int rep, float* xn) const
  {
    covarianceFourLanes(x, xd, g, dt, rep, xn, [this, x, xd, &g](float* Q) { computeQ(x, xd, g, Q); });
  }

  /** static settling on four lanes: wheel `rep`'s map lookup at the pose (x, y, heading sin_psi / cos_psi) with the body
   *  angles of g, then angle `rep` of the four arcsines; returns (0, 0, 0) without a map */
  __device__ __forceinline__ void settleFourLanes(const StepTrig& g, const float sin_psi, const float cos_psi, const float x,
                                                  const float y, const int rep, float& roll, float& pitch,
                                                  float& height) const
  {
    roll = 0.0f;
    pitch = 0.0f;
    height = 0.0f;
    if (this->S().tex_helper_.checkTextureUse(0))
    {
      float M[3][3], world[3], h_own, h[4], angle[4];
      bodyRotation(g, sin_psi, cos_psi, M);
      wheelWorldPoint(M, wheelOffsetX(rep), wheelOffsetY(rep), x, y, world);
      this->S().tex_helper_.queryTextureAtWorldPose(0, world, &h_own);
      allReplicas(h_own, h);
      allReplicas(mppi::det::asin(settlingSine(rep, h)), angle);
      settle(angle, h, roll, pitch, height);
    }
  }

  /**
   * One step with the work of a rollout shared out over its four replica lanes; every replica holds the whole state
   * before and after.  steer(x, u, xd) fills the steering entries of the derivative, post(x, xd, xn) runs after the
   * Euler update of the six integrated states (the LSTM-steering model integrates its steering rate there).
   * XD: entries of the derivative the model produces (6, or 9 with the steering-rate derivative).
   */
  template <int XD, class STEER, class POST>
  __device__ __forceinline__ void stepFourLanes(float* state, float* next_state, float* state_der, float* control, float* output,
                                       const float dt, STEER&& steer, POST&& post)
  {
    const PARAMS_T& p = this->S().params_;
    const int rep = (int)(threadIdx.x & 63) >> 4;
    float x[STATE_DIM], xn[STATE_DIM], xd[XD], u[CONTROL_DIM];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      x[i] = state[i];
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      u[i] = control[i];

    // ---- angles of the current state: replica 0 yaw, 1 wrapped steering angle, 2 raw steering angle, 3 pitch
    StepTrig g;
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
    computeParametricDelayDeriv(x, u, xd);
    steer(x, u, xd);
    // the acceleration does not look at the roll: g.sin_roll / g.cos_roll are filled in below
    computeParametricAccelDeriv(x, u, xd, g);
    updateState(x, xn, xd, dt);
    post(x, xd, xn);

    // ---- replicas 0, 1: roll of the current state; replicas 2, 3: yaw of the next one
    float sin_psi, cos_psi;
    {
      float s, c;
      mppi::det::sincos(angle_utils::normalizeAngle(rep < 2 ? x[RDE_S(ROLL)] : xn[RDE_S(YAW)]), &s, &c);
      g.sin_roll = fromReplica(s, 0);
      g.cos_roll = fromReplica(c, 0);
      sin_psi = fromReplica(s, 2);
      cos_psi = fromReplica(c, 2);
    }

    // ---- static settling: wheel `rep`, then angle `rep`
    float roll, pitch, height;
    settleFourLanes(g, sin_psi, cos_psi, xn[RDE_S(POS_X)], xn[RDE_S(POS_Y)], rep, roll, pitch, height);
    xn[RDE_S(PITCH)] = pitch;
    xn[RDE_S(ROLL)] = roll;

    covarianceFourLanes(x, xd, g, dt, rep, xn);

#pragma unroll
    for (int i = 0; i < 6; i++)
      state_der[i] = xd[i];
    if (XD > RDE_S(STEER_ANGLE_RATE))
      state_der[RDE_S(STEER_ANGLE_RATE)] = xd[XD - 1];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      next_state[i] = xn[i];
    output[RDE_O(BASELINK_POS_I_Z)] = height;
    setOutputs(xd, xn, output);
  }

  /** racer_dubins_elevation.cu:836-874 */
  __device__ __forceinline__ void step(float* state, float* next_state, float* state_der, float* control, float* output,
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
    const StepTrig g = stateTrig(x);
    computeParametricDelayDeriv(x, u, xd);
    computeParametricSteerDeriv(x, u, xd);
    computeParametricAccelDeriv(x, u, xd, g);
    updateState(x, xn, xd, dt);
    computeUncertaintyPropagation(x, xd, xn, dt, g);
    float roll, pitch, height;
    computeStaticSettling(xn[RDE_S(YAW)], xn[RDE_S(POS_X)], xn[RDE_S(POS_Y)], g, roll, pitch, height);
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
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  using PARAMS_T = RacerDubinsElevationParams;
  RacerDubinsElevation(hipStream_t stream = nullptr) : RacerDubinsElevationImpl<RacerDubinsElevation>(stream)
  {
  }
};

/**
 * The same model with FOUR lanes per rollout (REPLICATED_LANES, include/mppi_amd/engine/rollout_kernel.hpp): a wave
 * carries 16 rollouts, lane (column c, replica r) = c + 16 r, every replica keeps a private register copy of the whole
 * state and all four end a step with identical values.  Inside the step the replicas run the SAME instructions on
 * DIFFERENT data wherever the model has four of something:
 *   - the six angles whose sine / cosine a step needs: two passes of one det::sincos per lane (yaw, wrapped and raw
 *     steering angle, pitch | roll, next yaw) instead of six,
 *   - the four wheels: one bilinear map lookup per lane (4 loads in flight instead of 16), one asin per lane,
 *   - the four rows of the covariance update: 2 x 16 multiply-adds per lane instead of 2 x 64,
 * and exchange the results with ds_bpermute (`__shfl`): 36 values per step.  Each value is produced by exactly the
 * operations of the one-lane form, so the two forms and the oracle agree bit for bit.  What it buys: a lone wave issues
 * one instruction every ~2 ns whatever its dependences, so a step costs its instruction count on the lane
 * that carries the rollout — this form roughly halves it, and a block of 64 rollouts becomes four dynamics waves, one per
 * SIMD of the CU, where the one-lane form leaves three SIMDs to the helper waves.
 * The reference strides the same work over threadIdx.y with shared memory and block barriers
 * (racer_dubins_elevation.cu:336-419, 672-738: `for (i = pi; i < 16; i += step)`).
 */
class RacerDubinsElevationQuad : public RacerDubinsElevationImpl<RacerDubinsElevationQuad>
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
   *  nothing — profiles/r06_robust_racer_ab.json) */
  static constexpr int MPPI_RMPPI_PIPE_SAMPLERS = 1;
  static constexpr int MPPI_RMPPI_PIPE_COSTS = 1;
  using ELEVATION = RacerDubinsElevationImpl<RacerDubinsElevationQuad>;
  using PARAMS_T = RacerDubinsElevationParams;
  static constexpr int REPLICATED_LANES = 4;

  RacerDubinsElevationQuad(const RacerDubinsElevation& other) : ELEVATION(other.stream_)
  {
    this->params_ = other.params_;
    for (int i = 0; i < CONTROL_DIM; i++)
    {
      this->control_rngs_[i] = other.control_rngs_[i];
      this->control_deadband_[i] = other.control_deadband_[i];
      this->zero_control_[i] = other.zero_control_[i];
    }
    this->tex_helper_ = other.tex_helper_;
  }

  __device__ __forceinline__ void step(float* state, float* next_state, float* state_der, float* control, float* output,
                              float* theta_s, const float t, const float dt)
  {
    stepFourLanes<6>(
        state, next_state, state_der, control, output, dt,
        [this](const float* x, const float* u, float* xd) { computeParametricSteerDeriv(x, u, xd); },
        [](const float*, const float*, float*) {});
  }
};

#endif
