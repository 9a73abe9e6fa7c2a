/**
 * dynamics.hpp — CRTP base class of every Dynamics plugin: the contract the rollout kernels call.
 *
 * Source-level mirror of the reference's device-side contract (include/mppi/dynamics/dynamics.cuh:68-522 and
 * dynamics.cu:82-168): same class/param/method names, same argument meaning, same threadIdx.y lane striding.
 *   STATE_DIM / CONTROL_DIM / OUTPUT_DIM from PARAMS_T::{State,Control,Output}Index::NUM_*   (dynamics.cuh:74-76)
 *   initializeDynamics, enforceConstraints, computeStateDeriv, computeKinematics, updateState, step, stateToOutput
 * A model written for the reference keeps its device code (computeDynamics / computeKinematics / overrides) as is.
 *
 * The Eigen host overloads of a model (computeDynamics / step on Eigen vectors) are not mirrored: the engine never
 * evaluates a model on the host — the nominal state trajectory is re-rolled on the device by the same plugin code
 * (include/mppi_amd/engine/finalize_kernel.hpp) — so a model only needs its device methods.  What a CALLER does with the
 * model on the host — `model->enforceConstraints(x, u); model->step(x, x_next, xdot, u, y, t, dt);` in its simulation loop
 * (examples/cartpole_example.cu:76-80) — is served by the two host overloads declared below and defined in
 * plugin/dynamics_host.hpp: they run the model's DEVICE code for one rollout, so the caller's plant integrates with the
 * arithmetic the rollouts use.
 */
#ifndef MPPI_AMD_PLUGIN_DYNAMICS_HPP_
#define MPPI_AMD_PLUGIN_DYNAMICS_HPP_

#include <cfloat>
#include <hip/hip_runtime.h>
#include "mppi_amd/plugin/managed.hpp"
#include "mppi_amd/plugin/math_utils.hpp"
#include "mppi_amd/plugin/parallel_utils.hpp"
#include "mppi_amd/plugin/host_arrays.hpp"

#ifndef E_INDEX
#define E_INDEX(ENUM, enum_val) static_cast<int>(ENUM::enum_val)
#endif
#ifndef S_INDEX
#define S_IND_CLASS(CLASS, enum_val) E_INDEX(CLASS::StateIndex, enum_val)
#define S_INDEX(enum_val) S_IND_CLASS(PARENT_CLASS::DYN_PARAMS_T, enum_val)
#endif
#ifndef C_INDEX
#define C_IND_CLASS(CLASS, enum_val) E_INDEX(CLASS::ControlIndex, enum_val)
#define C_INDEX(enum_val) C_IND_CLASS(PARENT_CLASS::DYN_PARAMS_T, enum_val)
#endif
#ifndef O_INDEX
#define O_IND_CLASS(CLASS, enum_val) E_INDEX(CLASS::OutputIndex, enum_val)
#define O_INDEX(enum_val) O_IND_CLASS(PARENT_CLASS::DYN_PARAMS_T, enum_val)
#endif

/** reference: dynamics/dynamics.cuh:40-58 */
struct DynamicsParams
{
  enum class StateIndex : int
  {
    POS_X = 0,
    NUM_STATES
  };
  enum class ControlIndex : int
  {
    VEL_X = 0,
    NUM_CONTROLS
  };
  enum class OutputIndex : int
  {
    POS_X = 0,
    NUM_OUTPUTS
  };
};

namespace MPPI_internal
{
template <class CLASS_T, class PARAMS_T>
class Dynamics : public mppi::Managed
{
public:
  static const int STATE_DIM = S_IND_CLASS(PARAMS_T, NUM_STATES);
  static const int CONTROL_DIM = C_IND_CLASS(PARAMS_T, NUM_CONTROLS);
  static const int OUTPUT_DIM = O_IND_CLASS(PARAMS_T, NUM_OUTPUTS);
  typedef CLASS_T DYN_T;
  typedef PARAMS_T DYN_PARAMS_T;
  /* host-side vector types (reference: Eigen columns, dynamics.cuh:80-90; here plugin/host_arrays.hpp) */
  typedef mppi::host::Array<STATE_DIM> state_array;
  typedef mppi::host::Array<CONTROL_DIM> control_array;
  typedef mppi::host::Array<OUTPUT_DIM> output_array;

  Dynamics(hipStream_t stream = 0)
  {
    bindToStream(stream);
    for (int i = 0; i < CONTROL_DIM; i++)
    {
      control_rngs_[i].x = -FLT_MAX;  // reference: dynamics.cuh:84-93
      control_rngs_[i].y = FLT_MAX;
      control_deadband_[i] = 0.0f;
      zero_control_[i] = 0.0f;
    }
  }

  /* ------------------------------ host side: lifecycle (reference: dynamics.cu:3-81) ------------------------------ */
  hipError_t GPUSetup()
  {
    CLASS_T* derived = static_cast<CLASS_T*>(this);
    if (!GPUMemStatus_)
    {
      hipError_t e = Managed::GPUSetup(derived, &model_d_);
      if (e != hipSuccess)
        return e;
    }
    return derived->paramsToDevice();
  }
  hipError_t freeCudaMem()
  {
    hipError_t e = hipSuccess;
    if (GPUMemStatus_)
    {
      e = hipFree(model_d_);
      GPUMemStatus_ = false;
      model_d_ = nullptr;
    }
    return e;
  }
  hipError_t paramsToDevice(bool synchronize = true)
  {
    if (!GPUMemStatus_)
      return hipSuccess;
    hipError_t e = hipMemcpyAsync(&model_d_->params_, &params_, sizeof(PARAMS_T), hipMemcpyHostToDevice, stream_);
    if (e == hipSuccess)
      e = hipMemcpyAsync(&model_d_->control_rngs_, &control_rngs_, CONTROL_DIM * sizeof(float2),
                         hipMemcpyHostToDevice, stream_);
    if (e == hipSuccess)
      e = hipMemcpyAsync(&model_d_->control_deadband_, &control_deadband_, CONTROL_DIM * sizeof(float),
                         hipMemcpyHostToDevice, stream_);
    if (e == hipSuccess)
      e = hipMemcpyAsync(&model_d_->zero_control_, &zero_control_, CONTROL_DIM * sizeof(float), hipMemcpyHostToDevice,
                         stream_);
    if (e == hipSuccess && synchronize)
      e = hipStreamSynchronize(stream_);
    return e;
  }
  void setParams(const PARAMS_T& params)
  {
    params_ = params;
    (void)paramsToDevice();
  }
  __host__ __device__ PARAMS_T getParams() const
  {
    return params_;
  }
  void setControlRanges(const float2* control_rngs)
  {
    for (int i = 0; i < CONTROL_DIM; i++)
      control_rngs_[i] = control_rngs[i];
    (void)paramsToDevice();
  }
  void setControlDeadbands(const float* control_deadband)
  {
    for (int i = 0; i < CONTROL_DIM; i++)
      control_deadband_[i] = control_deadband[i];
    (void)paramsToDevice();
  }

  /* ------------------------------ host side: a caller's simulation loop ------------------------------ */
  /** Dynamics::enforceConstraints / step on host vectors (reference: dynamics.cuh:300-340 host overloads): evaluated by the
   *  model's device code on one lane — include "mppi_amd/plugin/dynamics_host.hpp" (hipcc) in the translation unit that
   *  calls them */
  void enforceConstraints(state_array& state, control_array& control);
  void step(state_array& state, state_array& next_state, state_array& state_der, control_array& control,
            output_array& output, const float t, const float dt);
  /** reference: dynamics.cuh printState */
  void printState(const float* state) const
  {
    printf("State:");
    for (int i = 0; i < STATE_DIM; i++)
      printf(" %f", state[i]);
    printf("\n");
  }

  /* ------------------------------ device side: what the kernels call ------------------------------ */
  /** reference: dynamics.cuh:429-435 — default: output <- state */
  __device__ inline void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    static_cast<CLASS_T*>(this)->stateToOutput(state, output);
  }

  /**
   * true when a plugin's enforceConstraints() reads `state` (override it in the plugin together with the function).  The
   * base rule — deadband, then clamp to control_rngs_ — does not, and the role-pipelined rollout kernels then apply it in
   * the SAMPLER waves (same function, same input, same result) instead of on the dynamics wave, where every instruction
   * lengthens the rollout's serial chain.
   */
  static constexpr bool CONSTRAINTS_DEPEND_ON_STATE = false;
  /**
   * true while the plugin keeps the base enforceConstraints() below (deadband, then clamp to control_rngs_).  The engine then
   * evaluates the rule on the HOST where the reference does (Controller::getCurrentControl -> model_->enforceConstraints on
   * the state-estimator thread, controllers/controller.cuh:329-345) — the same IEEE operations, no kernel launch behind
   * the rollouts in flight.  A plugin that overrides enforceConstraints() sets this to false; the engine then runs the
   * plugin's device code for it.
   */
  static constexpr bool BASE_CONSTRAINTS = true;

  /** reference: dynamics.cu:97-116 — deadband then clamp, lanes strided over threadIdx.y */
  __device__ inline void enforceConstraints(float* state, float* control)
  {
    int i, p_index, step;
    mppi::p1::getParallel1DIndex<mppi::p1::Parallel1Dir::THREAD_Y>(p_index, step);
    for (i = p_index; i < CONTROL_DIM; i += step)
    {
      if (fabsf(control[i]) < this->control_deadband_[i])
      {
        control[i] = this->zero_control_[i];
      }
      else
      {
        control[i] += this->control_deadband_[i] * -mppi::math::sign(control[i]);
      }
      control[i] = fminf(fmaxf(this->control_rngs_[i].x, control[i]), this->control_rngs_[i].y);
    }
  }

  /** default: no kinematic part */
  __device__ inline void computeKinematics(float* state, float* state_der)
  {
  }

  /** reference: dynamics.cu:82-95 */
  __device__ inline void computeStateDeriv(float* state, float* control, float* state_der, float* theta_s)
  {
    CLASS_T* derived = static_cast<CLASS_T*>(this);
    if (__builtin_amdgcn_workitem_id_y() == 0)
    {
      derived->computeKinematics(state, state_der);
    }
    derived->computeDynamics(state, control, state_der, theta_s);
  }

  /** reference: dynamics.cu:118-128 — explicit Euler */
  __device__ inline void updateState(float* state, float* next_state, float* state_der, const float dt)
  {
    int i, p_index, step;
    mppi::p1::getParallel1DIndex<mppi::p1::Parallel1Dir::THREAD_Y>(p_index, step);
#if defined(__HIP_DEVICE_COMPILE__)
    if (STATE_DIM % 2 == 0 && step == 1)
    {  // one lane per rollout: two states per packed v_pk_mul_f32 / v_pk_add_f32 (same roundings as the scalar form)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (i = 0; i + 1 < STATE_DIM; i += 2)
      {
        const f32x2 x = { state[i], state[i + 1] };
        const f32x2 d = { state_der[i], state_der[i + 1] };
        const f32x2 inc = d * f32x2{ dt, dt };
        const f32x2 n = x + inc;
        next_state[i] = n.x;
        next_state[i + 1] = n.y;
      }
      return;
    }
#endif
    for (i = p_index; i < STATE_DIM; i += step)
    {
      next_state[i] = state[i] + state_der[i] * dt;
    }
  }

  /** reference: dynamics.cu:130-142; the two block barriers become lane_sync() (parallel_utils.hpp) */
  __device__ inline void step(float* state, float* next_state, float* state_der, float* control, float* output,
                              float* theta_s, const float t, const float dt)
  {
    CLASS_T* derived = static_cast<CLASS_T*>(this);
    derived->computeStateDeriv(state, control, state_der, theta_s);
    mppi::lane_sync();
    derived->updateState(state, next_state, state_der, dt);
    mppi::lane_sync();
    derived->stateToOutput(next_state, output);
  }

  /** reference: dynamics.cu:144-155 */
  __device__ inline void stateToOutput(const float* __restrict__ state, float* __restrict__ output)
  {
    int p_index, step;
    mppi::p1::getParallel1DIndex<mppi::p1::Parallel1Dir::THREAD_Y>(p_index, step);
    for (int i = p_index; i < OUTPUT_DIM && i < STATE_DIM; i += step)
    {
      output[i] = state[i];
    }
  }
  /** reference: dynamics.cu:157-168 */
  __device__ inline void outputToState(const float* __restrict__ output, float* __restrict__ state)
  {
    int p_index, step;
    mppi::p1::getParallel1DIndex<mppi::p1::Parallel1Dir::THREAD_Y>(p_index, step);
    for (int i = p_index; i < OUTPUT_DIM && i < STATE_DIM; i += step)
    {
      state[i] = output[i];
    }
  }

  CLASS_T* model_d_ = nullptr;

  PARAMS_T params_;
  float2 control_rngs_[CONTROL_DIM];
  float control_deadband_[CONTROL_DIM];
  float zero_control_[CONTROL_DIM];
};
}  // namespace MPPI_internal

#endif
