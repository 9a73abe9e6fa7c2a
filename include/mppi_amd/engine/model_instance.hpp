/**
 * model_instance.hpp — type-erased wrapper around one (DYN, COST, SAMPLER) template instantiation.
 *
 * The C ABI cannot carry templates, so each registered model is one ModelT<...> object behind the ModelBase
 * interface — the role played in the reference by its explicit instantiations
 * (reference: include/mppi/instantiations/cartpole_mppi/cartpole_mppi.cuh, src/controllers/cartpole/cartpole_mppi.cu:30-42).
 * Block shapes are compile-time (see rollout_kernel.hpp); each model lists the shapes it is instantiated for.
 */
#ifndef MPPI_AMD_MODEL_INSTANCE_HPP_
#define MPPI_AMD_MODEL_INSTANCE_HPP_

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "mppi_amd.h"
#include "kernarg_view.hpp"
#include "rollout_kernel.hpp"
#include "finalize_kernel.hpp"
#include "rollout_pipeline_kernel.hpp"
#include "rmppi_kernels.hpp"
#include "rmppi_pipeline_kernel.hpp"
#include "mppi_amd/feedback_controllers/ddp_feedback.hpp"
#include "mppi_amd/sampling_distributions/colored_noise.hpp"

/**
 * hipFuncAttributeMaxDynamicSharedMemorySize of a kernel, raised only when a launch needs more than was granted before: the
 * attribute call is a runtime round trip (function lookup, locks) that every rollout launch of a > 48 KiB block paid in front of
 * its dispatch — on the host's critical path of mppi_compute_control, where the first launch's enqueue time decides when the
 * device starts.  The grant is per (device, kernel) and monotonic, so handles sharing a kernel on a device never lower each
 * other's limit.
 */
inline hipError_t ensureDynamicLds(const void* fn, size_t smem)
{
  if (smem <= 48 * 1024)
    return hipSuccess;
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t& g = granted[std::make_pair(dev, fn)];
  if (smem <= g)
    return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e == hipSuccess)
    g = smem;
  return e;
}

namespace mppi
{
namespace engine
{
constexpr size_t MAX_LDS_BYTES = 160 * 1024;  // gfx950: 160 KiB per CU, one workgroup may take all of it

template <int X, int Y, int Z>
struct Shape
{
};
template <class... S>
struct Shapes
{
};

/** everything the sampler needs to know for one launch (the engine refreshes it before every launch) */
struct SamplerLaunchState
{
  int num_rollouts_local, num_rollouts_global, rollout_offset;
  int num_timesteps, num_distributions;
  float* control_means_d;
  const float* eps_d;  // nullptr => Philox
  float* control_samples_d;
  uint64_t seed;
  uint32_t generation;
  int iteration;  // opt_iter for std_dev_decay
  int optimization_stride;
  int independent_noise;  // != 0: use_same_noise_for_all_distributions off (every distribution its own stream / eps slab)
};

struct ModelBase
{
  int S = 0, C = 0, O = 0;
  int default_bx = 64, default_by = 1;
  virtual ~ModelBase() = default;
  virtual mppi_status setDynamicsParams(const void* pod, size_t n) = 0;
  virtual mppi_status setCostParams(const void* pod, size_t n) = 0;
  virtual void setControlRanges(const float* lo_hi) = 0;
  virtual void setControlDeadband(const float* db) = 0;
  virtual void getZeroControl(float* out) const = 0;
  /** Dynamics::enforceLeash (dynamics/dynamics.cuh:448-466), host: per state dimension, the nominal state if it is within
   *  the leash of the true state, else the true state moved towards it by the leash.  A plugin with an enforceLeash() host
   *  method of its own (RacerDubins: position leash in the body frame, racer_dubins.cu:177-230) overrides it. */
  virtual void hostEnforceLeash(const float* state_true, const float* state_nominal, const float* leash, float* out) const
  {
    for (int i = 0; i < S; i++)
    {
      const float diff = fabsf(state_nominal[i] - state_true[i]);
      if (leash[i] < diff)
        out[i] = state_true[i] + fminf(fmaxf(state_nominal[i] - state_true[i], -leash[i]), leash[i]);
      else
        out[i] = state_nominal[i];
    }
  }
  /** the base enforceConstraints rule on the host (dynamics.cu:97-116); false: the plugin overrides it, run the device code */
  virtual bool hostEnforceConstraints(float* u) const = 0;
  virtual void setSamplerParams(const mppi_gaussian_params* p, int D) = 0;
  /** bulk data (NN weights, costmap); default: the model has none */
  virtual mppi_status setBlob(const std::string& name, const float* data, size_t count, const int* dims, int ndims,
                              hipStream_t stream, std::string& err)
  {
    err = "model has no blob named '" + name + "'";
    return MPPI_ERR_INVALID_ARG;
  }
  /** LSTMHelper::copyHiddenCellToDevice (lstm_helper.cu:480-500): new initial hidden / cell state of the rollouts' LSTM,
   *  on the engine's stream without re-uploading the weights; default: the model has no LSTM */
  virtual mppi_status setLSTMInitialState(const float* hidden, const float* cell, hipStream_t stream, std::string& err)
  {
    err = "model has no LSTM";
    return MPPI_ERR_INVALID_ARG;
  }
  /** time_specific_std_dev: sigma[D][T][C] table in device memory (owned by the engine), nullptr switches it off */
  virtual void setTimeSpecificStdDev(const float* table_d) = 0;
  /** colored-noise sampler parameters (ColoredNoiseParamsImpl, colored_noise.cuh:45-73); default: Gaussian sampler */
  virtual mppi_status setColoredNoiseParams(const float* exponents, float offset_decay_rate, float fmin)
  {
    return MPPI_ERR_STATE;
  }
  /** floats of injected noise per rollout: T*C time-domain eps, or C*(2T+2) spectrum entries for colored noise */
  virtual size_t noiseFloatsPerRollout(int T) const = 0;
  /** the sampled noise eps[K_local][T][C] the next launch would use, written to out_d (generator parity tests) */
  virtual mppi_status launchNoiseDump(const SamplerLaunchState& s, float* out_d, hipStream_t stream, std::string& err) = 0;
  /* ---- Robust MPPI (rmppi_kernels.hpp); default: the model is not instantiated for it ---- */
  virtual bool supportsRMPPI() const
  {
    return false;
  }
  /** DDP feedback gains [T][S][C] (ddp.cuh:18-60) -> device; accumulate_all_states: see ddp_feedback.hpp */
  virtual mppi_status setFeedbackGains(const float* gains, int T, bool accumulate_all_states, hipStream_t stream,
                                       std::string& err)
  {
    err = "model is not instantiated for Robust MPPI";
    return MPPI_ERR_UNSUPPORTED;
  }
  virtual mppi_status launchInitEval(bool pipeline, const kernels::InitEvalArgs& a, const SamplerLaunchState& s,
                                     hipStream_t stream, std::string& err)
  {
    err = "model is not instantiated for Robust MPPI";
    return MPPI_ERR_UNSUPPORTED;
  }
  /** pipeline: the role-pipelined kernel (rmppi_pipeline_kernel.hpp) when the model has one and its replicated-lane form is
   *  usable with the loaded networks, else rolloutRMPPIKernel */
  virtual mppi_status launchRMPPI(int bx, bool pipeline, const kernels::RMPPIArgs& a, const SamplerLaunchState& s,
                                  hipStream_t stream, std::string& err)
  {
    err = "model is not instantiated for Robust MPPI";
    return MPPI_ERR_UNSUPPORTED;
  }
  virtual size_t rmppiSharedBytes(int bx, int T)
  {
    return 0;
  }
  /** LDS request of the role-pipelined Robust MPPI kernel (64 rollouts x 2 systems per block); 0: the model has none, its
   *  replicated-lane form does not take the loaded networks, or not even the shortest rings fit */
  virtual size_t rmppiPipelineSharedBytes(int T)
  {
    return 0;
  }
  /** world -> texture transform of a costmap cost (ARStandardCost::updateTransform, ar_standard_cost.cu:132-137) */
  virtual mppi_status setCostmapTransform(const float* r_c1, const float* r_c2, const float* trs)
  {
    return MPPI_ERR_INVALID_ARG;
  }
  virtual bool supportsShape(int bx, int by, int bz) const = 0;
  /** role-pipelined variant (rollout_pipeline_kernel.hpp) available for this model? */
  virtual bool supportsPipeline() const
  {
    return false;
  }
  /** "" when every plugin class the model's ROLE-SEPARATED kernels would run declares MPPI_BARRIER_FREE_STEP
   *  (plugin/parallel_utils.hpp: barrier_free_step) — or when the model has no such kernels; otherwise the names the
   *  declaration is missing from.  mppi_create refuses a model with a non-empty answer: a block barrier inside a per-step
   *  method would hang those kernels (only the wave whose role it is calls the method). */
  virtual std::string undeclaredBarrierFreePlugins() const
  {
    return std::string();
  }
  /** rolloutPipelineKernel's STREAM_MERGE form exists: one system, Gaussian sampler drawing in the loop */
  virtual bool supportsStreamedMerge() const
  {
    return false;
  }
  /** role-pipelined variant with the two systems of Tube-MPPI folded into the lanes of a wave, shape (32, 1, 2) */
  virtual bool supportsPipelineFold(int bx, int by, int bz) const
  {
    return false;
  }
  /** role-pipelined variant for replicated-lane (MFMA) dynamics, shape (64, REP, 1) */
  virtual bool supportsPipelineRep(int bx, int by, int bz) const
  {
    return false;
  }
  /** the shape is a replicated-lane one that the model's present data rule out (networks of another shape than the
   *  replicated-lane form is compiled for) */
  virtual bool fastShapeRefused(int bx, int by, int bz) const
  {
    return false;
  }
  virtual size_t rolloutSharedBytes(int bx, int by, int bz, int T, int D, bool pipeline) = 0;
  /** floats the sampler needs for the sample rows of `blocks` blocks of `slots` rollout slots in HBM (long horizons), 0: the
   *  sampler keeps its rows in LDS only; setGlobalRows(ptr) switches the fused kernel over (nullptr: back to LDS) */
  virtual size_t globalRowsFloats(int blocks, int slots, int T) const
  {
    return 0;
  }
  virtual void setGlobalRows(float* rows_d)
  {
  }
  /** every registered block shape as (bx, by, bz) triples, flattened */
  virtual void listShapes(std::vector<int>& out) const = 0;
  virtual mppi_status launchRollout(int bx, int by, int bz, bool pipeline, const kernels::RolloutArgs& args,
                                    const SamplerLaunchState& s, hipStream_t stream, std::string& err) = 0;
  virtual mppi_status launchFinalize(int D, const kernels::FinalizeArgs& a, hipStream_t stream, std::string& err) = 0;
  /** the control phase of a split hand-over that merges the last launch's block records itself (kernels::mergeControlKernel):
   *  one system on the plain one-wave finalize form, control sequence in LDS */
  virtual bool supportsMergeControl(int num_timesteps) const
  {
    return false;
  }
  virtual mppi_status launchMergeControl(const kernels::FinalizeArgs& a, const kernels::MergeControlArgs& m, hipStream_t stream,
                                         std::string& err)
  {
    err = "model has no merging control phase";
    return MPPI_ERR_LAUNCH_SHAPE;
  }
  /** x <- one model step (optionally after enforceConstraints on u), one block (1, by, 1) */
  virtual mppi_status launchModelStep(float* x_d, float* u_d, float dt, int enforce, hipStream_t stream,
                                      std::string& err) = 0;
};

/** fingerprint of the structures a model translation unit and libmppi_amd.so exchange (mppi_register_model refuses a
 *  plugin built against other headers: its kernels would read the argument blocks with the wrong layout) */
/** bumped BY HAND whenever ModelBase's virtual methods are added, removed or reordered or a field of an argument struct is
 *  swapped at equal size — changes sizeof() cannot see (a stale plugin would dispatch to the wrong vtable slot).
 *  3: rows-in-HBM arguments; 4: release-flag arguments of the finalize kernels (both round 3); 5: supportsStreamedMerge (round 4);
 *  6: FinalizeArgs::phases / carry block of the split hand-over (round 5); 7: undeclaredBarrierFreePlugins; 8: the transposed
 *  record copy (RolloutArgs) and supportsMergeControl / launchMergeControl (both round 6). */
#define MPPI_ENGINE_ABI_VERSION 8

constexpr int engineAbiFingerprint()
{
  return MPPI_ENGINE_ABI_VERSION * 1000003 +
         (int)(sizeof(ModelBase) + 131 * sizeof(kernels::RolloutArgs) + 131 * 131 * sizeof(kernels::FinalizeArgs) +
               7 * sizeof(kernels::RMPPIArgs) + 17 * sizeof(kernels::InitEvalArgs) + 31 * sizeof(SamplerLaunchState) +
               37 * sizeof(kernels::MergeControlArgs));
}

template <class DYN_T, int BY>
__global__ void __launch_bounds__(BY) modelStepKernel(DYN_T dynamics_obj, float* x_d, float* u_d, float dt, int enforce)
{
  __builtin_assume(__builtin_amdgcn_workgroup_size_x() == 1);
  __builtin_assume(__builtin_amdgcn_workgroup_size_y() == BY);
  __builtin_assume(__builtin_amdgcn_workgroup_size_z() == 1);
  __builtin_assume(__builtin_amdgcn_workitem_id_x() == 0);
  __builtin_assume(__builtin_amdgcn_workitem_id_y() < BY);
  __builtin_assume(__builtin_amdgcn_workitem_id_z() == 0);
  DYN_T* dynamics = &dynamics_obj;
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s = reinterpret_cast<float*>(smem_raw);
  float* x = theta_s + calcClassSharedMemSize(dynamics, 1) / (int)sizeof(float);
  float* xn = x + math::nearest_multiple_4(S);
  float* xdot = xn + math::nearest_multiple_4(S);
  float* u = xdot + math::nearest_multiple_4(S);
  float* y = u + math::nearest_multiple_4(C);
  const int ty = (int)__builtin_amdgcn_workitem_id_y();
  for (int i = ty; i < S; i += BY)
  {
    x[i] = x_d[i];
    xdot[i] = 0.0f;
  }
  for (int i = ty; i < C; i += BY)
    u[i] = u_d[i];
  for (int i = ty; i < O; i += BY)
    y[i] = 0.0f;
  __syncthreads();
  dynamics->initializeDynamics(x, u, y, theta_s, 0.0f, dt);
  __syncthreads();
  if (enforce)
  {
    dynamics->enforceConstraints(x, u);
    __syncthreads();
  }
  dynamics->step(x, xn, xdot, u, y, theta_s, 0, dt);
  __syncthreads();
  for (int i = ty; i < S; i += BY)
    x_d[i] = xn[i];
  for (int i = ty; i < C; i += BY)
    u_d[i] = u[i];
}

}  // namespace engine
namespace kernels
{
/** the block's sample rows exactly as the rollout kernels see them after initializeDistributions(): raw eps[k][t][c]
 *  (pre-filled modes only), block = 64 rollouts x one lane */
template <class SAMPLING_T>
__global__ void __launch_bounds__(64) noiseDumpKernel(SAMPLING_T sampling_obj, float* out_d)
{
  SAMPLING_T* sampling = &sampling_obj;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_d = reinterpret_cast<float*>(smem_raw);
  sampling->initializeDistributions(nullptr, 0.0f, 0.0f, theta_d);
  __syncthreads();
  const int T = sampling->params_.num_timesteps;
  const int TC = T * SAMPLING_T::CONTROL_DIM;
  const int stride = SAMPLING_T::rowStride(T);
  const int row0 = (int)blockIdx.x * 64;
  const int nrows = min(64, sampling->params_.num_rollouts - row0);
  for (int e = (int)threadIdx.x; e < nrows * TC; e += 64)
  {
    const int r = e / TC, j = e - r * TC;
    out_d[(size_t)(row0 + r) * TC + j] = theta_d[r * stride + j];
  }
}
}  // namespace kernels
namespace engine
{
/* detection of optional plugin members */
template <class T, class = void>
struct has_fnn_helper : std::false_type
{
};
template <class T>
struct has_fnn_helper<T, std::void_t<decltype(std::declval<T&>().helper_.theta_d_)>> : std::true_type
{
};
template <class T, class = void>
struct has_lstm_helper : std::false_type
{
};
template <class T>
struct has_lstm_helper<T, std::void_t<decltype(std::declval<T&>().lstm_.weights_d_)>> : std::true_type
{
};
template <class T, class = void>
struct has_host_leash : std::false_type
{
};
template <class T>
struct has_host_leash<T, std::void_t<decltype(std::declval<const T&>().enforceLeash((const float*)nullptr, (const float*)nullptr,
                                                                                     (const float*)nullptr, (float*)nullptr))>>
  : std::true_type
{
};
template <class T, class = void>
struct has_lstm_structure : std::false_type
{
};
/** dynamics whose LSTM shape can be changed at run time (setLSTMStructure, the "lstm_structure" blob) */
template <class T>
struct has_lstm_structure<T, std::void_t<decltype(std::declval<T&>().setLSTMStructure((const int*)nullptr, 0))>> : std::true_type
{
};
/** a replicated-lane form that names a sibling made for ONE rollout per wave (`using FINALIZE_FORM = ...`) */
template <class T, class = void>
struct has_finalize_form : std::false_type
{
};
template <class T>
struct has_finalize_form<T, std::void_t<typename T::FINALIZE_FORM>> : std::true_type
{
};
template <class T, class = void>
struct has_register_form : std::false_type
{
};
/** dynamics whose replicated-lane (FAST) variant exists for one network shape only (`register_form_`) */
template <class T>
struct has_register_form<T, std::void_t<decltype(std::declval<T&>().register_form_)>> : std::true_type
{
};
template <class T, class = void>
struct has_elevation_map : std::false_type
{
};
/** dynamics with a TwoDTextureHelper member `tex_helper_` (the elevation-map RACER models) */
template <class T>
struct has_elevation_map<T, std::void_t<decltype(std::declval<T&>().tex_helper_.textures_[0].data)>> : std::true_type
{
};
template <class T, class = void>
struct has_mean_unc_networks : std::false_type
{
};
/** dynamics with the mean and uncertainty LSTMs of RacerDubinsElevationLSTMUncertainty next to the steering one */
template <class T>
struct has_mean_unc_networks<T, std::void_t<decltype(std::declval<T&>().mean_lstm_d_), decltype(std::declval<T&>().unc_lstm_d_)>>
  : std::true_type
{
};
template <class T, class = void>
struct has_normals_map : std::false_type
{
};
/** dynamics with a second, four-channel map `normals_tex_helper_` (the suspension RACER models) */
template <class T>
struct has_normals_map<T, std::void_t<decltype(std::declval<T&>().normals_tex_helper_.textures_[0].data)>> : std::true_type
{
};
template <class T, class = void>
struct has_costmap : std::false_type
{
};
template <class T>
struct has_costmap<T, std::void_t<decltype(std::declval<T&>().costmap_d_)>> : std::true_type
{
};

/**
 * DYN_FAST_T / FAST_SHAPES: an optional second Dynamics type implementing the same model for particular block shapes
 * (e.g. the MFMA forward of the NN model for shape (16, 4)); it is constructed from the primary object at launch, so
 * parameters, control ranges and blobs are shared.
 */
template <class DYN_T, class COST_T, class SAMPLING_T, class SHAPES, int FIN_BY = 1, class DYN_FAST_T = void,
          class FAST_SHAPES = Shapes<>, bool PIPELINE = false, bool RMPPI = false>
struct ModelT : ModelBase
{
  /* ---- Robust MPPI: block = (64 rollouts, 1, 2 systems), one lane per rollout and system ---- */
  DeviceDDP<DYN_T> fb;
  float* gains_d = nullptr;
  int gains_T = 0;
  bool supportsRMPPI() const override
  {
    return RMPPI;
  }
  mppi_status setFeedbackGains(const float* gains, int T, bool accumulate_all_states, hipStream_t stream,
                               std::string& err) override
  {
    if constexpr (RMPPI)
    {
      const size_t n = (size_t)T * DYN_T::STATE_DIM * DYN_T::CONTROL_DIM;
      hipError_t e = hipSuccess;
      if (gains_T != T)
      {
        e = hipStreamSynchronize(stream);
        if (e == hipSuccess && gains_d)
          e = hipFree(gains_d);
        gains_d = nullptr;
        if (e == hipSuccess)
          e = hipMalloc((void**)&gains_d, n * sizeof(float));
        gains_T = T;
      }
      if (e == hipSuccess)
        e = hipMemcpyAsync(gains_d, gains, n * sizeof(float), hipMemcpyHostToDevice, stream);
      if (e == hipSuccess)
        e = hipStreamSynchronize(stream);
      if (e != hipSuccess)
      {
        err = std::string("feedback gain upload: ") + hipGetErrorString(e);
        return MPPI_ERR_HIP;
      }
      fb.fb_gain_traj_d_ = gains_d;
      fb.num_timesteps_ = T;
      fb.accumulate_all_states_ = accumulate_all_states;
      return MPPI_OK;
    }
    return ModelBase::setFeedbackGains(gains, T, accumulate_all_states, stream, err);
  }
  /** Robust MPPI's two kernels run on the replicated-lane (FAST) form of a model when there is one — and when it is usable:
   *  the four-lane forms of the RACER models are compiled for the reference's network shapes, a network of another shape
   *  (`register_form_` false) runs the kernels on the model itself, one lane per rollout and system (round 3; it used to be
   *  refused).  f(rm_dyn) is called with the object the kernels take. */
  static constexpr bool RMPPI_HAS_FALLBACK = has_register_form<DYN_T>::value && !std::is_void<DYN_FAST_T>::value;
  bool rmppiUseFast() const
  {
    if constexpr (std::is_void<DYN_FAST_T>::value)
      return false;
    else if constexpr (has_register_form<DYN_T>::value)
      return dyn.register_form_;
    else
      return true;
  }
  template <class F>
  auto withRmppiDynamics(F&& f)
  {
    if constexpr (std::is_void<DYN_FAST_T>::value)
      return f(dyn);
    else if constexpr (RMPPI_HAS_FALLBACK)
    {
      if (!dyn.register_form_)
        return f(dyn);
      DYN_FAST_T fast(dyn);
      return f(fast);
    }
    else
    {
      DYN_FAST_T fast(dyn);
      return f(fast);
    }
  }
  size_t rmppiSharedBytes(int bx, int T) override
  {
    if constexpr (RMPPI)
    {
      smp.params_.num_timesteps = T;
      smp.params_.num_distributions = 2;
      return withRmppiDynamics([&](auto& rm_dyn) { return kernels::rmppiSharedBytes(rm_dyn, cost, fb, smp, bx); });
    }
    return 0;
  }
  /** the role-pipelined Robust MPPI kernels (rmppi_pipeline_kernel.hpp) exist for models with a replicated-lane form, and for
   *  models registered for the role-pipelined rollout (PIPELINE: one lane per rollout, no block barrier inside the per-step
   *  methods) — with a sampler whose rows may live in HBM and need no block-wide prologue of their own (the Gaussian one) */
  static constexpr bool rmppiHasPipeline()
  {
    if constexpr (!RMPPI || !SAMPLING_T::SUPPORTS_GLOBAL_ROWS || SAMPLING_T::COLORED)
      return false;
    else if constexpr (!std::is_void<DYN_FAST_T>::value)
      return kernels::replicated_lanes<DYN_FAST_T>::value > 1;
    else
      return PIPELINE;
  }
  /** can the pipelined kernels run with the networks loaded right now (replicated-lane forms exist for one shape only) */
  bool rmppiPipelineUsable() const
  {
    if constexpr (!rmppiHasPipeline())
      return false;
    else if constexpr (std::is_void<DYN_FAST_T>::value)
      return true;
    else
      return rmppiUseFast();
  }
  /** f(pipe_dyn): the dynamics object the pipelined kernels take — the replicated-lane form (or the form IT names for this
   *  kernel: `using RMPPI_PIPELINE_FORM = ...` — a class of the same arithmetic whose trade-offs are made for a block of 11-15
   *  waves at 128-168 VGPRs each: dynamics/racer_dubins/racer_dubins_elevation_lstm_unc.hpp), or the model itself */
  template <class T, class = void>
  struct rmppi_pipeline_form
  {
    using type = T;
  };
  template <class T>
  struct rmppi_pipeline_form<T, std::void_t<typename T::RMPPI_PIPELINE_FORM>>
  {
    using type = typename T::RMPPI_PIPELINE_FORM;
  };
  template <class F>
  auto withRmppiPipelineDynamics(F&& f)
  {
    if constexpr (std::is_void<DYN_FAST_T>::value)
      return f(dyn);
    else
    {
      typename rmppi_pipeline_form<DYN_FAST_T>::type fast(dyn);
      return f(fast);
    }
  }
  size_t rmppiPipelineSharedBytes(int T) override
  {
    if constexpr (rmppiHasPipeline())
    {
      if (!rmppiPipelineUsable())
        return 0;
      smp.params_.num_timesteps = T;
      smp.params_.num_distributions = 2;
      return withRmppiPipelineDynamics([&](auto& pd) -> size_t {
        const kernels::RMPPIPipeRings r = kernels::rmppiPipelineRings(pd, cost, fb, smp, MAX_LDS_BYTES);
        return r.out_steps ? kernels::rmppiPipelineSharedBytes(pd, cost, fb, smp, r) : 0;
      });
    }
    return 0;
  }
  mppi_status launchRMPPIPipeline(const kernels::RMPPIArgs& a, hipStream_t stream, std::string& err)
  {
    if constexpr (rmppiHasPipeline())
    {
      return withRmppiPipelineDynamics([&](auto& pd) -> mppi_status {
        using PD_T = std::decay_t<decltype(pd)>;
        const kernels::RMPPIPipeRings r = kernels::rmppiPipelineRings(pd, cost, fb, smp, MAX_LDS_BYTES);
        if (r.out_steps == 0)
        {
          err = "pipelined RMPPI rollout kernel: the rings do not fit 160 KiB of LDS next to the sample rows";
          return MPPI_ERR_LDS_OVERFLOW;
        }
        const size_t smem = kernels::rmppiPipelineSharedBytes(pd, cost, fb, smp, r);
        const bool in_loop = SAMPLING_T::IN_LOOP_DRAW && smp.noise_source_ == 0;
        auto kfn = in_loop ? kernels::rolloutRMPPIPipelineKernel<PD_T, COST_T, DeviceDDP<DYN_T>, SAMPLING_T, SAMPLING_T::IN_LOOP_DRAW>
                           : kernels::rolloutRMPPIPipelineKernel<PD_T, COST_T, DeviceDDP<DYN_T>, SAMPLING_T, false>;
        if (smem > 48 * 1024)
          (void)ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
        hipLaunchKernelGGL(kfn, dim3((a.base.num_rollouts + 63) / 64), dim3(64 * kernels::rmppiPipelineWaves<PD_T>(), 1, 1), smem,
                           stream, pd, cost, fb, smp, a, r);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
        {
          err = std::string("rolloutRMPPIPipelineKernel launch: ") + hipGetErrorString(e);
          return MPPI_ERR_HIP;
        }
        return MPPI_OK;
      });
    }
    err = "model has no role-pipelined Robust MPPI kernel";
    return MPPI_ERR_LAUNCH_SHAPE;
  }
  mppi_status launchInitEval(bool pipeline, const kernels::InitEvalArgs& a, const SamplerLaunchState& s, hipStream_t stream,
                             std::string& err) override
  {
    if constexpr (RMPPI)
    {
      if (!blobsReady(err))
        return MPPI_ERR_STATE;
      prepSampler(s);
      constexpr int BX = 64;
      if constexpr (rmppiHasPipeline())
      {
        // the candidate rollouts as blocks of role waves (rmppi_pipeline_kernel.hpp) when the rows of 64 rollouts and a ring
        // fit the LDS; else the fused kernel below
        if (pipeline && rmppiPipelineUsable())
        {
          bool launched = false;
          const mppi_status st = withRmppiPipelineDynamics([&](auto& pd) -> mppi_status {
            using PD_T = std::decay_t<decltype(pd)>;
            const int ring = kernels::initEvalPipelineRing(pd, cost, a.num_timesteps, MAX_LDS_BYTES);
            if (ring == 0)
              return MPPI_OK;  // rows + ring do not fit: the fused kernel below
            launched = true;
            constexpr int REP = kernels::replicated_lanes<PD_T>::value;
            const size_t smem = kernels::initEvalPipelineSharedBytes(pd, cost, a.num_timesteps, ring);
            auto kfn = kernels::initEvalPipelineKernel<PD_T, COST_T, SAMPLING_T>;
            if (smem > 48 * 1024)
              (void)ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
            constexpr int WAVES = REP + kernels::INIT_EVAL_PIPE_SAMPLERS + kernels::INIT_EVAL_PIPE_COSTS;
            hipLaunchKernelGGL(kfn, dim3((a.num_eval_rollouts + BX - 1) / BX), dim3(64 * WAVES, 1, 1), smem, stream, pd, cost, smp, a,
                               ring);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess)
            {
              err = std::string("initEvalPipelineKernel launch: ") + hipGetErrorString(e);
              return MPPI_ERR_HIP;
            }
            return MPPI_OK;
          });
          if (launched || st != MPPI_OK)
            return st;
        }
      }
      // models with replicated-lane (MFMA / four-lane) dynamics run both Robust MPPI kernels on them (withRmppiDynamics)
      return withRmppiDynamics([&](auto& rm_dyn) -> mppi_status {
        using RM_DYN_T = std::decay_t<decltype(rm_dyn)>;
        constexpr int REP = kernels::replicated_lanes<RM_DYN_T>::value;
        const size_t smem = kernels::initEvalSharedBytes<RM_DYN_T, COST_T, SAMPLING_T>(rm_dyn, cost, BX);
        auto kfn = kernels::initEvalKernel<RM_DYN_T, COST_T, SAMPLING_T, BX>;
        if (smem > 48 * 1024)
          (void)ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
        hipLaunchKernelGGL(kfn, dim3((a.num_eval_rollouts + BX - 1) / BX), dim3(BX * REP, 1, 1), smem, stream, rm_dyn, cost,
                           smp, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
        {
          err = std::string("initEvalKernel launch: ") + hipGetErrorString(e);
          return MPPI_ERR_HIP;
        }
        return MPPI_OK;
      });
    }
    return ModelBase::launchInitEval(pipeline, a, s, stream, err);
  }
  template <int BX>
  mppi_status launchRMPPIShape(const kernels::RMPPIArgs& a, hipStream_t stream, std::string& err)
  {
    if constexpr (RMPPI)
    {
      return withRmppiDynamics([&](auto& rm_dyn) -> mppi_status {
        using RM_DYN_T = std::decay_t<decltype(rm_dyn)>;
        constexpr int REP = kernels::replicated_lanes<RM_DYN_T>::value;
        const size_t smem = kernels::rmppiSharedBytes(rm_dyn, cost, fb, smp, BX);
        if (smem > MAX_LDS_BYTES)
        {
          err = "RMPPI rollout kernel needs " + std::to_string(smem) + " B of LDS per block; gfx950 has 163840";
          return MPPI_ERR_LDS_OVERFLOW;
        }
        const bool in_loop = SAMPLING_T::IN_LOOP_DRAW && smp.noise_source_ == 0;
        auto kfn = in_loop ? kernels::rolloutRMPPIKernel<RM_DYN_T, COST_T, DeviceDDP<DYN_T>, SAMPLING_T, BX, SAMPLING_T::IN_LOOP_DRAW>
                           : kernels::rolloutRMPPIKernel<RM_DYN_T, COST_T, DeviceDDP<DYN_T>, SAMPLING_T, BX, false>;
        if (smem > 48 * 1024)
          (void)ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
        hipLaunchKernelGGL(kfn, dim3((a.base.num_rollouts + BX - 1) / BX), dim3(BX * REP, 1, 2), smem, stream, rm_dyn, cost, fb,
                           smp, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
        {
          err = std::string("rolloutRMPPIKernel launch: ") + hipGetErrorString(e);
          return MPPI_ERR_HIP;
        }
        return MPPI_OK;
      });
    }
    err = "model is not registered for Robust MPPI";
    return MPPI_ERR_UNSUPPORTED;
  }
  mppi_status launchRMPPI(int bx, bool pipeline, const kernels::RMPPIArgs& a, const SamplerLaunchState& s, hipStream_t stream,
                          std::string& err) override
  {
    if constexpr (RMPPI)
    {
      if (!blobsReady(err))
        return MPPI_ERR_STATE;
      if (!fb.fb_gain_traj_d_ || fb.num_timesteps_ != s.num_timesteps)
      {
        err = "Robust MPPI needs the DDP feedback gains [T][S][C] (mppi_set_feedback_gains) before it can run";
        return MPPI_ERR_STATE;
      }
      prepSampler(s);
      if constexpr (rmppiHasPipeline())
      {
        if (pipeline && bx == 64 && rmppiPipelineUsable())
          return launchRMPPIPipeline(a, stream, err);
      }
      if (bx == 64)
        return launchRMPPIShape<64>(a, stream, err);
      if (bx == 32)  // horizons whose sample rows for 64 rollouts x 2 systems do not fit the LDS
        return launchRMPPIShape<32>(a, stream, err);
      err = "Robust MPPI rollout kernel is instantiated for 64 or 32 rollouts per block";
      return MPPI_ERR_LAUNCH_SHAPE;
    }
    return ModelBase::launchRMPPI(bx, pipeline, a, s, stream, err);
  }

  bool supportsPipeline() const override
  {
    return PIPELINE;
  }
  /** does the instantiation ask for role-separated kernels (the one-lane pipeline, or the replicated-lane one), and do the
   *  classes those kernels would run say that their per-step methods hold no block barrier? */
  static constexpr bool ROLE_SEPARATED = PIPELINE || !std::is_void<DYN_FAST_T>::value;
  static constexpr bool fastDeclared()
  {
    if constexpr (std::is_void<DYN_FAST_T>::value)
      return true;
    else
      return mppi::barrier_free_step<DYN_FAST_T>::value;
  }
  static constexpr bool BARRIER_FREE_DECLARED = (!PIPELINE || mppi::barrier_free_step<DYN_T>::value) && fastDeclared() &&
                                                (!ROLE_SEPARATED || (mppi::barrier_free_step<COST_T>::value &&
                                                                     mppi::barrier_free_step<SAMPLING_T>::value));
  std::string undeclaredBarrierFreePlugins() const override
  {
    std::string missing;
    if constexpr (ROLE_SEPARATED)
    {
      if (PIPELINE && !mppi::barrier_free_step<DYN_T>::value)
        missing += "Dynamics ";
      if (!fastDeclared())
        missing += "Dynamics(replicated-lane form) ";
      if (!mppi::barrier_free_step<COST_T>::value)
        missing += "Cost ";
      if (!mppi::barrier_free_step<SAMPLING_T>::value)
        missing += "SamplingDistribution ";
    }
    return missing;
  }
  bool supportsStreamedMerge() const override
  {
    return PIPELINE && SAMPLING_T::IN_LOOP_DRAW && !SAMPLING_T::COLORED;
  }
  bool supportsPipelineRep(int bx, int by, int bz) const override
  {
    if constexpr (!std::is_void<DYN_FAST_T>::value)
      return bx == 64 && bz == 1 && by == kernels::replicated_lanes<DYN_FAST_T>::value && hasShape(FAST_SHAPES{}, bx, by, bz);
    return false;
  }
  bool fastShapeRefused(int bx, int by, int bz) const override
  {
    if constexpr (has_register_form<DYN_T>::value && !std::is_void<DYN_FAST_T>::value)
      return !dyn.register_form_ && hasShape(FAST_SHAPES{}, bx, by, bz);
    return false;
  }
  template <class FAST = DYN_FAST_T>
  mppi_status launchPipelineRep(const kernels::RolloutArgs& args, hipStream_t stream, std::string& err)
  {
    if constexpr (!std::is_void<FAST>::value)
    {
      FAST fast(dyn);
      constexpr int REP = kernels::replicated_lanes<FAST>::value;
      const int grid = (args.num_rollouts + 63) / 64;
      const bool in_loop = SAMPLING_T::IN_LOOP_DRAW && smp.noise_source_ == 0;
      const int ring = kernels::pipelineRepRingSteps(fast, cost, smp, MAX_LDS_BYTES);
      if (ring == 0)
      {
        err = "pipeline rollout kernel: the sample rows leave no room for the output ring in 160 KiB of LDS";
        return MPPI_ERR_LDS_OVERFLOW;
      }
      const size_t smem = kernels::pipelineRepSharedBytes(fast, cost, smp, ring);
      auto kfn = in_loop ? kernels::rolloutPipelineRepKernel<FAST, COST_T, SAMPLING_T, SAMPLING_T::IN_LOOP_DRAW>
                         : kernels::rolloutPipelineRepKernel<FAST, COST_T, SAMPLING_T, false>;
      if constexpr (SAMPLING_T::SUPPORTS_GLOBAL_ROWS)
      {
        if (smp.rows_global_d_)  // long horizons: the sample rows in HBM
          kfn = in_loop ? kernels::rolloutPipelineRepKernel<FAST, COST_T, SAMPLING_T, SAMPLING_T::IN_LOOP_DRAW, true>
                        : kernels::rolloutPipelineRepKernel<FAST, COST_T, SAMPLING_T, false, true>;
      }
      if (smem > 48 * 1024)
        (void)ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
      constexpr int WAVES = REP + kernels::PIPE_REP_SAMPLERS + kernels::PIPE_REP_COSTS;  // dynamics + helper waves
      hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * WAVES, 1, 1), smem, stream, fast, cost, smp, args, ring);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess)
      {
        err = std::string("rolloutPipelineRepKernel launch: ") + hipGetErrorString(e);
        return MPPI_ERR_HIP;
      }
      return MPPI_OK;
    }
    err = "model has no replicated-lane dynamics";
    return MPPI_ERR_LAUNCH_SHAPE;
  }

  bool supportsPipelineFold(int bx, int by, int bz) const override
  {
    return PIPELINE && bx == 32 && by == 1 && bz == 2;
  }
  template <int Z, bool FOLD = false>
  mppi_status launchPipeline(const kernels::RolloutArgs& args, hipStream_t stream, std::string& err)
  {
    if constexpr (PIPELINE)
    {
      const size_t smem = kernels::pipelineSharedBytes(dyn, cost, smp, Z, FOLD);
      if (smem > MAX_LDS_BYTES)
      {
        err = "pipeline rollout kernel needs " + std::to_string(smem) + " B of LDS per block; gfx950 has 163840";
        return MPPI_ERR_LDS_OVERFLOW;
      }
      const bool in_loop = SAMPLING_T::IN_LOOP_DRAW && smp.noise_source_ == 0;
      auto kfn = in_loop ? kernels::rolloutPipelineKernel<DYN_T, COST_T, SAMPLING_T, Z, SAMPLING_T::IN_LOOP_DRAW, FOLD>
                         : kernels::rolloutPipelineKernel<DYN_T, COST_T, SAMPLING_T, Z, false, FOLD>;
      if constexpr (SAMPLING_T::SUPPORTS_GLOBAL_ROWS)
      {
        if (smp.rows_global_d_)  // long horizons: the sample rows in HBM
          kfn = in_loop ? kernels::rolloutPipelineKernel<DYN_T, COST_T, SAMPLING_T, Z, SAMPLING_T::IN_LOOP_DRAW, FOLD, true>
                        : kernels::rolloutPipelineKernel<DYN_T, COST_T, SAMPLING_T, Z, false, FOLD, true>;
      }
      if constexpr (Z == 1 && !FOLD && SAMPLING_T::IN_LOOP_DRAW && !SAMPLING_T::COLORED)
      {
        // the previous iteration's block records merged by this launch's sampler waves (RolloutArgs::prev_records_d)
        if (args.prev_records_d && args.prev_records_t_d && in_loop && !smp.rows_global_d_)
          kfn = kernels::rolloutPipelineKernel<DYN_T, COST_T, SAMPLING_T, 1, SAMPLING_T::IN_LOOP_DRAW, false, false, true>;
        else if (args.prev_records_d)
        {
          err = "prev_records_d: this launch cannot merge the previous records itself (noise source / rows in HBM / no transposed copy)";
          return MPPI_ERR_STATE;
        }
      }
      else if (args.prev_records_d)
      {
        err = "prev_records_d: the streamed merge exists for one-system Gaussian pipeline launches only";
        return MPPI_ERR_STATE;
      }
      if (smem > 48 * 1024)
        (void)ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
      constexpr int RPB = FOLD ? 64 / Z : 64;  // rollouts per block
      const int grid = (args.num_rollouts + RPB - 1) / RPB;
      hipLaunchKernelGGL(kfn, dim3(grid), dim3(kernels::pipelineBlockX(Z, FOLD), 1, FOLD ? 1 : Z), smem, stream, dyn, cost,
                         smp, args);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess)
      {
        err = std::string("rolloutPipelineKernel launch: ") + hipGetErrorString(e);
        return MPPI_ERR_HIP;
      }
      return MPPI_OK;
    }
    err = "model is not registered for the pipeline variant";
    return MPPI_ERR_LAUNCH_SHAPE;
  }

  DYN_T dyn;
  COST_T cost;
  SAMPLING_T smp;
  /* colored noise: basis table cache */
  float* basis_d = nullptr;
  int basis_T = -1, basis_stride = -1;
  size_t basis_floats = 0;
  bool basis_dirty = true;
  float* weights_d = nullptr;
  float* weights2_d = nullptr;
  float* costmap_d = nullptr;
  float* elevation_d = nullptr;
  float* normals_d = nullptr;
  bool normals_transform_set = false;
  float* extra_net_d[4] = { nullptr, nullptr, nullptr, nullptr };  ///< mean LSTM, mean MLP, uncertainty LSTM, uncertainty MLP

  ~ModelT() override
  {
    if (gains_d)
      (void)hipFree(gains_d);
    if (basis_d)
      (void)hipFree(basis_d);
    if (weights_d)
      (void)hipFree(weights_d);
    if (weights2_d)
      (void)hipFree(weights2_d);
    if (costmap_d)
      (void)hipFree(costmap_d);
    if (elevation_d)
      (void)hipFree(elevation_d);
    if (normals_d)
      (void)hipFree(normals_d);
    for (float* q : extra_net_d)
      if (q)
        (void)hipFree(q);
  }

  static mppi_status upload(float** dst, const float* src, size_t count, hipStream_t stream, std::string& err)
  {
    if (*dst)
      (void)hipFree(*dst);
    *dst = nullptr;
    hipError_t e = hipMalloc((void**)dst, count * sizeof(float));
    if (e == hipSuccess)
      e = hipMemcpyAsync(*dst, src, count * sizeof(float), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess)
      e = hipStreamSynchronize(stream);
    if (e != hipSuccess)
    {
      err = std::string("blob upload: ") + hipGetErrorString(e);
      return MPPI_ERR_HIP;
    }
    return MPPI_OK;
  }

  mppi_status setBlob(const std::string& name, const float* data, size_t count, const int* dims, int ndims,
                      hipStream_t stream, std::string& err) override
  {
    if constexpr (has_fnn_helper<DYN_T>::value)
    {
      if (name == "dynamics_weights")
      {
        if ((int)count != dyn.helper_.NUM_PARAMS)
        {
          err = "dynamics_weights: expected " + std::to_string(dyn.helper_.NUM_PARAMS) + " floats, got " +
                std::to_string(count);
          return MPPI_ERR_INVALID_ARG;
        }
        mppi_status st = upload(&weights_d, data, count, stream, err);
        dyn.helper_.theta_d_ = weights_d;
        return st;
      }
    }
    if constexpr (has_lstm_structure<DYN_T>::value)
    {
      /* the reference sizes its LSTM from the network file (LSTMHelper(npz), lstm_helper.cu:13-62); here the shape is
       * {hidden size, output-network layer sizes ...} as floats, given before the weights */
      if (name == "lstm_structure")
      {
        std::vector<int> desc(count);
        for (size_t i = 0; i < count; i++)
          desc[i] = (int)data[i];
        if (!dyn.setLSTMStructure(desc.data(), (int)count))
        {
          err = "lstm_structure: expected {H, H + inputs, ..., outputs} with at least one output-network layer";
          return MPPI_ERR_INVALID_ARG;
        }
        dyn.lstm_.weights_d_ = nullptr;  // blobs of the previous shape no longer fit
        dyn.lstm_.output_nn_.theta_d_ = nullptr;
        return MPPI_OK;
      }
    }
    if constexpr (has_lstm_helper<DYN_T>::value)
    {
      if (name == "lstm_weights")
      {
        if ((int)count != dyn.lstm_.getNumParams())
        {
          err = "lstm_weights: expected " + std::to_string(dyn.lstm_.getNumParams()) + " floats, got " +
                std::to_string(count);
          return MPPI_ERR_INVALID_ARG;
        }
        mppi_status st = upload(&weights_d, data, count, stream, err);
        dyn.lstm_.weights_d_ = weights_d;
        return st;
      }
      if (name == "lstm_output_weights")
      {
        if ((int)count != dyn.lstm_.output_nn_.NUM_PARAMS)
        {
          err = "lstm_output_weights: expected " + std::to_string(dyn.lstm_.output_nn_.NUM_PARAMS) + " floats, got " +
                std::to_string(count);
          return MPPI_ERR_INVALID_ARG;
        }
        mppi_status st = upload(&weights2_d, data, count, stream, err);
        dyn.lstm_.output_nn_.theta_d_ = weights2_d;
        return st;
      }
    }
    if constexpr (has_elevation_map<DYN_T>::value)
    {
      /* TwoDTextureHelper::updateTexture + enableTexture (texture_helper.cu:135-190): {height, width} heights, row-major */
      if (name == "elevation_map")
      {
        if (ndims != 2 || dims[0] <= 0 || dims[1] <= 0 || (size_t)dims[0] * dims[1] != count)
        {
          err = "elevation_map: dims must be {height, width} with height*width == count";
          return MPPI_ERR_INVALID_ARG;
        }
        mppi_status st = upload(&elevation_d, data, count, stream, err);
        auto& tex = dyn.tex_helper_.textures_[0];
        tex.data = elevation_d;
        tex.height = dims[0];
        tex.width = dims[1];
        tex.use = (st == MPPI_OK);
        return st;
      }
      /* updateOrigin / updateRotation / updateResolution (texture_helper.cu:192-268): origin[3], rotations[9] row-major,
       * resolution[3] */
      if (name == "elevation_map_transform")
      {
        if (count != 15)
        {
          err = "elevation_map_transform: expected 15 floats (origin[3], rotations[9], resolution[3])";
          return MPPI_ERR_INVALID_ARG;
        }
        auto& tex = dyn.tex_helper_.textures_[0];
        for (int i = 0; i < 3; i++)
          tex.origin[i] = data[i];
        for (int i = 0; i < 9; i++)
          tex.rotations[i] = data[3 + i];
        for (int i = 0; i < 3; i++)
          tex.resolution[i] = data[12 + i];
        if constexpr (has_normals_map<DYN_T>::value)
        {  // the normals map shares the frame unless it was given its own
          if (!normals_transform_set)
          {
            auto& ntex = dyn.normals_tex_helper_.textures_[0];
            for (int i = 0; i < 3; i++)
            {
              ntex.origin[i] = tex.origin[i];
              ntex.resolution[i] = tex.resolution[i];
            }
            for (int i = 0; i < 9; i++)
              ntex.rotations[i] = tex.rotations[i];
          }
        }
        return MPPI_OK;
      }
    }
    if constexpr (has_mean_unc_networks<DYN_T>::value)
    {
      /* LSTMLSTMHelper(path, "terra/mean_network/") / (…, "terra/uncertainty_network/") (racer_dubins_elevation_lstm_unc.cu:30-33):
       * parameter blobs in the layouts of lstm_helper.hpp / fnn_helper.hpp; "<net>_lstm_state" = [hidden | cell], the
       * per-cycle update of updateFromBuffer (:98-141), written in place */
      if (name == "mean_lstm_structure" || name == "unc_lstm_structure")
      {
        /* another hidden size / output network than the reference's test shapes: {H, H + inputs, ..., outputs} as floats,
         * given before the weights (the reference sizes the networks from the file, lstm_helper.cu:13-62) */
        const int which = name[0] == 'm' ? 1 : 2;
        std::vector<int> desc(count);
        for (size_t i = 0; i < count; i++)
          desc[i] = (int)data[i];
        if (!dyn.setNetworkStructure(which, desc.data(), (int)count))
        {
          err = name + ": expected {H, H + inputs, ..., outputs} with the model's input / output sizes (12 -> 2, 13 -> 5)";
          return MPPI_ERR_INVALID_ARG;
        }
        // blobs of the previous shape no longer fit
        for (int k = 2 * (which - 1); k < 2 * which; k++)
        {
          if (extra_net_d[k])
            (void)hipFree(extra_net_d[k]);
          extra_net_d[k] = nullptr;
        }
        (which == 1 ? dyn.mean_lstm_d_ : dyn.unc_lstm_d_) = nullptr;
        (which == 1 ? dyn.mean_fnn_d_ : dyn.unc_fnn_d_) = nullptr;
        (which == 1 ? dyn.mean_lstm_ : dyn.unc_lstm_).weights_d_ = nullptr;
        (which == 1 ? dyn.mean_lstm_ : dyn.unc_lstm_).output_nn_.theta_d_ = nullptr;
        return MPPI_OK;
      }
      const struct
      {
        const char* name;
        int slot;
        size_t count;
      } nets[6] = { { "mean_lstm_weights", 0, (size_t)dyn.mean_lstm_.getNumParams() },
                    { "mean_lstm_output_weights", 1, (size_t)dyn.mean_lstm_.output_nn_.NUM_PARAMS },
                    { "unc_lstm_weights", 2, (size_t)dyn.unc_lstm_.getNumParams() },
                    { "unc_lstm_output_weights", 3, (size_t)dyn.unc_lstm_.output_nn_.NUM_PARAMS },
                    { "mean_lstm_state", 0, (size_t)2 * dyn.mean_lstm_.HIDDEN_DIM },
                    { "unc_lstm_state", 2, (size_t)2 * dyn.unc_lstm_.HIDDEN_DIM } };
      for (int i = 0; i < 6; i++)
      {
        if (name != nets[i].name)
          continue;
        if (count != nets[i].count)
        {
          err = name + ": expected " + std::to_string(nets[i].count) + " floats, got " + std::to_string(count);
          return MPPI_ERR_INVALID_ARG;
        }
        if (i >= 4)
        {  // initial hidden / cell state into the tail of the uploaded LSTM blob
          float* blob = extra_net_d[nets[i].slot];
          if (!blob)
          {
            err = "set the network's weights before its state";
            return MPPI_ERR_STATE;
          }
          const size_t params = (size_t)(i == 4 ? dyn.mean_lstm_.LSTM_NUM_PARAMS : dyn.unc_lstm_.LSTM_NUM_PARAMS);
          hipError_t e = hipMemcpyAsync(blob + params, data, count * sizeof(float), hipMemcpyHostToDevice, stream);
          if (e == hipSuccess)
            e = hipStreamSynchronize(stream);
          if (e != hipSuccess)
          {
            err = std::string("LSTM state upload: ") + hipGetErrorString(e);
            return MPPI_ERR_HIP;
          }
          return MPPI_OK;
        }
        mppi_status st = upload(&extra_net_d[nets[i].slot], data, count, stream, err);
        // only what was uploaded for the present shapes (a structure blob clears its network's pair)
        if (nets[i].slot == 0)
          dyn.mean_lstm_.weights_d_ = dyn.mean_lstm_d_ = extra_net_d[0];
        else if (nets[i].slot == 1)
          dyn.mean_lstm_.output_nn_.theta_d_ = dyn.mean_fnn_d_ = extra_net_d[1];
        else if (nets[i].slot == 2)
          dyn.unc_lstm_.weights_d_ = dyn.unc_lstm_d_ = extra_net_d[2];
        else
          dyn.unc_lstm_.output_nn_.theta_d_ = dyn.unc_fnn_d_ = extra_net_d[3];
        return st;
      }
    }
    if constexpr (has_normals_map<DYN_T>::value)
    {
      /* getTextureHelperNormals()->updateTexture(0, float4 data) (racer_dubins_elevation_suspension_lstm.cuh:131-134; the tests'
       * set-up tests/dynamics/racer_dubins_elevation_suspension_test.cu:170-190): {height, width, 4} = nx, ny, nz, unused */
      if (name == "normals_map")
      {
        if (ndims != 3 || dims[0] <= 0 || dims[1] <= 0 || dims[2] != 4 || (size_t)dims[0] * dims[1] * 4 != count)
        {
          err = "normals_map: dims must be {height, width, 4} with height*width*4 == count";
          return MPPI_ERR_INVALID_ARG;
        }
        mppi_status st = upload(&normals_d, data, count, stream, err);
        auto& tex = dyn.normals_tex_helper_.textures_[0];
        tex.data = normals_d;
        tex.height = dims[0];
        tex.width = dims[1];
        tex.use = (st == MPPI_OK);
        return st;
      }
      if (name == "normals_map_transform")
      {
        if (count != 15)
        {
          err = "normals_map_transform: expected 15 floats (origin[3], rotations[9], resolution[3])";
          return MPPI_ERR_INVALID_ARG;
        }
        auto& tex = dyn.normals_tex_helper_.textures_[0];
        for (int i = 0; i < 3; i++)
          tex.origin[i] = data[i];
        for (int i = 0; i < 9; i++)
          tex.rotations[i] = data[3 + i];
        for (int i = 0; i < 3; i++)
          tex.resolution[i] = data[12 + i];
        normals_transform_set = true;
        return MPPI_OK;
      }
    }
    if constexpr (has_costmap<COST_T>::value)
    {
      if (name == "costmap")
      {
        if (ndims != 2 || dims[0] <= 0 || dims[1] <= 0 || (size_t)dims[0] * dims[1] != count)
        {
          err = "costmap: dims must be {height, width} with height*width == count";
          return MPPI_ERR_INVALID_ARG;
        }
        mppi_status st = upload(&costmap_d, data, count, stream, err);
        cost.costmap_d_ = costmap_d;
        cost.height_ = dims[0];
        cost.width_ = dims[1];
        return st;
      }
    }
    return ModelBase::setBlob(name, data, count, dims, ndims, stream, err);
  }

  mppi_status setLSTMInitialState(const float* hidden, const float* cell, hipStream_t stream, std::string& err) override
  {
    if constexpr (has_lstm_helper<DYN_T>::value)
    {
      if (!weights_d || !dyn.lstm_.weights_d_)
      {
        err = "set the 'lstm_weights' blob before the initial hidden / cell state";
        return MPPI_ERR_STATE;
      }
      const int H = dyn.lstm_.HIDDEN_DIM;
      for (int i = 0; i < H; i++)
        if (!std::isfinite(hidden[i]) || !std::isfinite(cell[i]))
        {  // det::tanh(NaN) = -1: a NaN recurrent state would be masked, not propagated (see mppi_set_model_blob)
          err = "non-finite hidden / cell state";
          return MPPI_ERR_NAN;
        }
      float* tail = weights_d + dyn.lstm_.LSTM_NUM_PARAMS;
      hipError_t e = hipMemcpyAsync(tail, hidden, H * sizeof(float), hipMemcpyHostToDevice, stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(tail + H, cell, H * sizeof(float), hipMemcpyHostToDevice, stream);
      if (e == hipSuccess)
        e = hipStreamSynchronize(stream);  // the caller's buffers are pageable: they may go away after the call
      if (e != hipSuccess)
      {
        err = std::string("LSTM initial state upload: ") + hipGetErrorString(e);
        return MPPI_ERR_HIP;
      }
      return MPPI_OK;
    }
    return ModelBase::setLSTMInitialState(hidden, cell, stream, err);
  }

  mppi_status setCostmapTransform(const float* r_c1, const float* r_c2, const float* trs) override
  {
    if constexpr (has_costmap<COST_T>::value)
    {
      for (int i = 0; i < 3; i++)
      {
        cost.params_.r_c1[i] = r_c1[i];
        cost.params_.r_c2[i] = r_c2[i];
        cost.params_.trs[i] = trs[i];
      }
      return MPPI_OK;
    }
    return MPPI_ERR_INVALID_ARG;
  }

  /** models with bulk data must have it before the first launch */
  bool blobsReady(std::string& err) const
  {
    if constexpr (has_fnn_helper<DYN_T>::value)
      if (!dyn.helper_.theta_d_)
      {
        err = "model needs the 'dynamics_weights' blob (mppi_set_model_blob) before it can run";
        return false;
      }
    if constexpr (has_lstm_helper<DYN_T>::value)
      if (!dyn.lstm_.weights_d_ || !dyn.lstm_.output_nn_.theta_d_)
      {
        err = "model needs the 'lstm_weights' and 'lstm_output_weights' blobs (mppi_set_model_blob) before it can run";
        return false;
      }
    if constexpr (has_costmap<COST_T>::value)
      if (!cost.costmap_d_)
      {
        err = "model needs the 'costmap' blob (mppi_set_model_blob) before it can run";
        return false;
      }
    if constexpr (has_mean_unc_networks<DYN_T>::value)
      if (!dyn.mean_lstm_d_ || !dyn.mean_fnn_d_ || !dyn.unc_lstm_d_ || !dyn.unc_fnn_d_)
      {
        err = "model needs the 'mean_lstm_weights', 'mean_lstm_output_weights', 'unc_lstm_weights' and "
              "'unc_lstm_output_weights' blobs (mppi_set_model_blob) before it can run";
        return false;
      }
    return true;
  }

  ModelT()
  {
    S = DYN_T::STATE_DIM;
    C = DYN_T::CONTROL_DIM;
    O = DYN_T::OUTPUT_DIM;
  }

  mppi_status setDynamicsParams(const void* pod, size_t n) override
  {
    // the params struct adds its fields after the (empty) DynamicsParams base: a flat block of floats / ints
    if (std::is_empty<typename DYN_T::DYN_PARAMS_T>::value)
      return n == 0 ? MPPI_OK : MPPI_ERR_INVALID_ARG;
    if (n != sizeof(typename DYN_T::DYN_PARAMS_T))
      return MPPI_ERR_INVALID_ARG;
    memcpy((void*)&dyn.params_, pod, n);
    return MPPI_OK;
  }
  mppi_status setCostParams(const void* pod, size_t n) override
  {
    if (n != sizeof(typename COST_T::COST_PARAMS_T))
      return MPPI_ERR_INVALID_ARG;
    memcpy((void*)&cost.params_, pod, n);
    return MPPI_OK;
  }
  void setControlRanges(const float* lo_hi) override
  {
    for (int i = 0; i < DYN_T::CONTROL_DIM; i++)
    {
      dyn.control_rngs_[i].x = lo_hi[2 * i];
      dyn.control_rngs_[i].y = lo_hi[2 * i + 1];
    }
  }
  void setControlDeadband(const float* db) override
  {
    for (int i = 0; i < DYN_T::CONTROL_DIM; i++)
      dyn.control_deadband_[i] = db[i];
  }
  void getZeroControl(float* out) const override
  {
    for (int i = 0; i < DYN_T::CONTROL_DIM; i++)
      out[i] = dyn.zero_control_[i];
  }
  void hostEnforceLeash(const float* state_true, const float* state_nominal, const float* leash, float* out) const override
  {
    if constexpr (has_host_leash<DYN_T>::value)
      dyn.enforceLeash(state_true, state_nominal, leash, out);
    else
      ModelBase::hostEnforceLeash(state_true, state_nominal, leash, out);
  }
  bool hostEnforceConstraints(float* u) const override
  {
    if (!DYN_T::BASE_CONSTRAINTS || DYN_T::CONSTRAINTS_DEPEND_ON_STATE)
      return false;
    for (int i = 0; i < DYN_T::CONTROL_DIM; i++)
    {  // the operations of Dynamics::enforceConstraints (plugin/dynamics.hpp), correctly rounded on either side
      if (fabsf(u[i]) < dyn.control_deadband_[i])
        u[i] = dyn.zero_control_[i];
      else
        u[i] += dyn.control_deadband_[i] * -mppi::math::sign(u[i]);
      u[i] = fminf(fmaxf(dyn.control_rngs_[i].x, u[i]), dyn.control_rngs_[i].y);
    }
    return true;
  }
  void setSamplerParams(const mppi_gaussian_params* p, int D) override
  {
    for (int i = 0; i < DYN_T::CONTROL_DIM * D && i < DYN_T::CONTROL_DIM * 2; i++)
      smp.params_.std_dev[i] = p->std_dev[i];
    if (D == 1)  // keep distribution 1 defined (copyStdDevToDistribution semantics, gaussian.cuh:45-60)
      for (int i = 0; i < DYN_T::CONTROL_DIM; i++)
        smp.params_.std_dev[DYN_T::CONTROL_DIM + i] = p->std_dev[i];
    for (int i = 0; i < DYN_T::CONTROL_DIM; i++)
      smp.params_.control_cost_coeff[i] = p->control_cost_coeff[i];
    smp.params_.pure_noise_trajectories_percentage = p->pure_noise_trajectories_percentage;
    smp.params_.std_dev_decay = p->std_dev_decay;
    smp.params_.sum_strides = p->sum_strides;
  }

  void setTimeSpecificStdDev(const float* table_d) override
  {
    smp.std_dev_time_d_ = table_d;
    smp.params_.time_specific_std_dev = table_d != nullptr;
  }
  mppi_status setColoredNoiseParams(const float* exponents, float offset_decay_rate, float fmin) override
  {
    if constexpr (SAMPLING_T::COLORED)
    {
      for (int i = 0; i < DYN_T::CONTROL_DIM; i++)
        smp.exponents_[i] = exponents[i];
      smp.offset_decay_rate_ = offset_decay_rate;
      smp.fmin_ = fmin;
      basis_dirty = true;
      return MPPI_OK;
    }
    return MPPI_ERR_STATE;
  }
  size_t noiseFloatsPerRollout(int T) const override
  {
    if constexpr (SAMPLING_T::COLORED)
      return (size_t)DYN_T::CONTROL_DIM * sampling_distributions::coloredSpectrumFloats(T);
    return (size_t)DYN_T::CONTROL_DIM * T;
  }
  /** colored noise: (re)build the basis table for this (T, optimization_stride, parameters) and upload it */
  mppi_status prepBasis(const SamplerLaunchState& s, hipStream_t stream, std::string& err)
  {
    if constexpr (SAMPLING_T::COLORED)
    {
      if (basis_dirty || basis_T != s.num_timesteps || basis_stride != s.optimization_stride)
      {
        std::vector<float> frag;
        const bool radix4 = sampling_distributions::coloredUseRadix4(s.num_timesteps, s.optimization_stride);
        if (radix4)  // T a multiple of 4: two FFT decimation steps in front of a GEMM a quarter the size (colored_noise.hpp)
          sampling_distributions::buildColoredNoiseBasisRadix4(s.num_timesteps, DYN_T::CONTROL_DIM, smp.exponents_,
                                                               smp.offset_decay_rate_, smp.fmin_, frag);
        else
          sampling_distributions::buildColoredNoiseBasis(s.num_timesteps, DYN_T::CONTROL_DIM, smp.exponents_,
                                                         smp.offset_decay_rate_, smp.fmin_, s.optimization_stride, frag);
        hipError_t e = hipStreamSynchronize(stream);  // earlier launches may still read the old table
        if (e == hipSuccess && basis_d && (basis_T != s.num_timesteps || basis_floats != frag.size()))
        {
          e = hipFree(basis_d);
          basis_d = nullptr;
        }
        if (e == hipSuccess && !basis_d)
          e = hipMalloc((void**)&basis_d, frag.size() * sizeof(float));
        if (e == hipSuccess)
          e = hipMemcpyAsync(basis_d, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess)
          e = hipStreamSynchronize(stream);  // frag is a local
        if (e != hipSuccess)
        {
          err = std::string("colored-noise basis upload: ") + hipGetErrorString(e);
          return MPPI_ERR_HIP;
        }
        basis_T = s.num_timesteps;
        basis_floats = frag.size();
        basis_stride = s.optimization_stride;
        basis_dirty = false;
      }
      smp.basis_d_ = basis_d;
    }
    return MPPI_OK;
  }

  mppi_status launchNoiseDump(const SamplerLaunchState& s, float* out_d, hipStream_t stream, std::string& err) override
  {
    if (SAMPLING_T::IN_LOOP_DRAW && !s.eps_d)
    {
      err = "noise dump: this sampler draws inside the step loop; use mppi_philox_normal for its stream";
      return MPPI_ERR_UNSUPPORTED;
    }
    prepSampler(s);
    mppi_status st = prepBasis(s, stream, err);
    if (st != MPPI_OK)
      return st;
    smp.params_.num_distributions = 1;
    SAMPLING_T dump_smp = smp;
    dump_smp.rows_global_d_ = nullptr;  // the dump kernel reads the rows back from ITS LDS, whatever the rollouts use
    const size_t smem = calcClassSharedMemSize(&dump_smp, 64);
    if (smem > MAX_LDS_BYTES)
    {
      err = "noise dump kernel: the rows of 64 rollouts do not fit the LDS at this horizon";
      return MPPI_ERR_LDS_OVERFLOW;
    }
    auto kfn = kernels::noiseDumpKernel<SAMPLING_T>;
    if (smem > 48 * 1024)
      (void)ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
    hipLaunchKernelGGL(kfn, dim3((s.num_rollouts_local + 63) / 64), dim3(64, 1, 1), smem, stream, dump_smp, out_d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
      err = std::string("noiseDumpKernel launch: ") + hipGetErrorString(e);
      return MPPI_ERR_HIP;
    }
    return MPPI_OK;
  }

  template <int X, int Y, int Z, class... Rest>
  static bool hasShape(Shapes<Shape<X, Y, Z>, Rest...>, int bx, int by, int bz)
  {
    return (bx == X && by == Y && bz == Z) || hasShape(Shapes<Rest...>{}, bx, by, bz);
  }
  static bool hasShape(Shapes<>, int, int, int)
  {
    return false;
  }
  bool supportsShape(int bx, int by, int bz) const override
  {
    return hasShape(SHAPES{}, bx, by, bz) || hasShape(FAST_SHAPES{}, bx, by, bz);
  }
  template <int X, int Y, int Z, class... Rest>
  static void appendShapes(Shapes<Shape<X, Y, Z>, Rest...>, std::vector<int>& out)
  {
    out.push_back(X);
    out.push_back(Y);
    out.push_back(Z);
    appendShapes(Shapes<Rest...>{}, out);
  }
  static void appendShapes(Shapes<>, std::vector<int>&)
  {
  }
  void listShapes(std::vector<int>& out) const override
  {
    appendShapes(FAST_SHAPES{}, out);  // the MFMA shapes first: preferred at equal rollouts per block
    appendShapes(SHAPES{}, out);
  }

  void prepSampler(const SamplerLaunchState& s)
  {
    smp.params_.num_rollouts = s.num_rollouts_local;
    smp.params_.num_timesteps = s.num_timesteps;
    smp.params_.num_distributions = s.num_distributions;
    smp.control_means_d_ = s.control_means_d;
    smp.eps_d_ = s.eps_d;
    smp.control_samples_d_ = s.control_samples_d;
    smp.noise_source_ = s.eps_d ? 1 /* NOISE_EPS_BUFFER */ : 0 /* NOISE_PHILOX_FUSED */;
    smp.seed_ = s.seed;
    smp.generation_ = s.generation;
    smp.rollout_offset_ = s.rollout_offset;
    smp.num_rollouts_global_ = s.num_rollouts_global;
    smp.params_.use_same_noise_for_all_distributions = s.independent_noise == 0;
    smp.setIteration(s.iteration, s.optimization_stride);
  }

  size_t globalRowsFloats(int blocks, int slots, int T) const override
  {
    if constexpr (SAMPLING_T::SUPPORTS_GLOBAL_ROWS)
      return (size_t)blocks * slots * SAMPLING_T::rowStrideGlobal(T);
    return 0;
  }
  void setGlobalRows(float* rows_d) override
  {
    if constexpr (SAMPLING_T::SUPPORTS_GLOBAL_ROWS)
      smp.rows_global_d_ = rows_d;
  }
  size_t rolloutSharedBytes(int bx, int by, int bz, int T, int D, bool pipeline) override
  {
    smp.params_.num_timesteps = T;
    smp.params_.num_distributions = D;
    if constexpr (!std::is_void<DYN_FAST_T>::value)
    {
      if (pipeline && supportsPipelineRep(bx, by, bz))
      {
        DYN_FAST_T fast(dyn);
        const int ring = kernels::pipelineRepRingSteps(fast, cost, smp, MAX_LDS_BYTES);
        return ring > 0 ? kernels::pipelineRepSharedBytes(fast, cost, smp, ring) : MAX_LDS_BYTES + 1;
      }
    }
    if (pipeline)
      return kernels::pipelineSharedBytes(dyn, cost, smp, bz, supportsPipelineFold(bx, by, bz));
    if constexpr (!std::is_void<DYN_FAST_T>::value)
    {
      if (hasShape(FAST_SHAPES{}, bx, by, bz))
      {
        DYN_FAST_T fast(dyn);
        return kernels::rolloutSharedBytes(fast, cost, smp, bx, 1, bz);
      }
    }
    return kernels::rolloutSharedBytes(dyn, cost, smp, bx, by, bz);
  }

  /** fast variant: shape (X, REP, Z) runs as X rollouts x REP replicated lanes with one contract lane (BY = 1) */
  template <int X, int Y, int Z, class FAST = DYN_FAST_T>
  mppi_status launchFastShape(const kernels::RolloutArgs& args, hipStream_t stream, std::string& err)
  {
    FAST fast(dyn);
    constexpr int REP = kernels::replicated_lanes<FAST>::value;
    static_assert(REP == Y, "fast shape: BY must equal the plugin's REPLICATED_LANES");
    const size_t smem = kernels::rolloutSharedBytes(fast, cost, smp, X, 1, Z);
    if (smem > MAX_LDS_BYTES)
    {
      err = "rollout kernel needs " + std::to_string(smem) + " B of LDS per block; gfx950 has 163840";
      return MPPI_ERR_LDS_OVERFLOW;
    }
    const bool in_loop = SAMPLING_T::IN_LOOP_DRAW && smp.noise_source_ == 0;
    auto kfn = in_loop ? kernels::rolloutKernel<FAST, COST_T, SAMPLING_T, X, 1, Z, SAMPLING_T::IN_LOOP_DRAW>
                       : kernels::rolloutKernel<FAST, COST_T, SAMPLING_T, X, 1, Z, false>;
    if (smem > 48 * 1024)
      (void)ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
    const int grid = (args.num_rollouts + X - 1) / X;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(X * REP, 1, Z), smem, stream, fast, cost, smp, args);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
      err = std::string("rolloutKernel (fast variant) launch: ") + hipGetErrorString(e);
      return MPPI_ERR_HIP;
    }
    return MPPI_OK;
  }
  template <int X, int Y, int Z, class... Rest>
  mppi_status dispatchFast(Shapes<Shape<X, Y, Z>, Rest...>, int bx, int by, int bz, const kernels::RolloutArgs& args,
                           hipStream_t stream, std::string& err, bool& handled)
  {
    if (bx == X && by == Y && bz == Z)
    {
      handled = true;
      return launchFastShape<X, Y, Z>(args, stream, err);
    }
    return dispatchFast(Shapes<Rest...>{}, bx, by, bz, args, stream, err, handled);
  }
  mppi_status dispatchFast(Shapes<>, int, int, int, const kernels::RolloutArgs&, hipStream_t, std::string&, bool& handled)
  {
    handled = false;
    return MPPI_OK;
  }

  template <int X, int Y, int Z>
  mppi_status launchShape(const kernels::RolloutArgs& args, hipStream_t stream, std::string& err)
  {
    const size_t smem = kernels::rolloutSharedBytes(dyn, cost, smp, X, Y, Z);
    if (smem > MAX_LDS_BYTES)
    {
      err = "rollout kernel needs " + std::to_string(smem) + " B of LDS per block; gfx950 has 163840";
      return MPPI_ERR_LDS_OVERFLOW;
    }
    // in-loop Philox draw needs one lane per rollout; otherwise the rows are pre-filled by initializeDistributions
    const bool in_loop = (Y == 1) && SAMPLING_T::IN_LOOP_DRAW && smp.noise_source_ == 0;
    auto kfn = in_loop ? kernels::rolloutKernel<DYN_T, COST_T, SAMPLING_T, X, Y, Z, (Y == 1) && SAMPLING_T::IN_LOOP_DRAW>
                       : kernels::rolloutKernel<DYN_T, COST_T, SAMPLING_T, X, Y, Z, false>;
    if (smem > 48 * 1024)
    {
      hipError_t e = ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
      if (e != hipSuccess)
      {
        err = std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e);
        return MPPI_ERR_HIP;
      }
    }
    const int grid = (args.num_rollouts + X - 1) / X;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(X, Y, Z), smem, stream, dyn, cost, smp, args);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
      err = std::string("rolloutKernel launch: ") + hipGetErrorString(e);
      return MPPI_ERR_HIP;
    }
    return MPPI_OK;
  }

  template <int X, int Y, int Z, class... Rest>
  mppi_status dispatch(Shapes<Shape<X, Y, Z>, Rest...>, int bx, int by, int bz, const kernels::RolloutArgs& args,
                       hipStream_t stream, std::string& err)
  {
    if (bx == X && by == Y && bz == Z)
      return launchShape<X, Y, Z>(args, stream, err);
    return dispatch(Shapes<Rest...>{}, bx, by, bz, args, stream, err);
  }
  mppi_status dispatch(Shapes<>, int bx, int by, int bz, const kernels::RolloutArgs&, hipStream_t, std::string& err)
  {
    err = "block shape (" + std::to_string(bx) + "," + std::to_string(by) + "," + std::to_string(bz) +
          ") is not instantiated for this model";
    return MPPI_ERR_LAUNCH_SHAPE;
  }

  mppi_status launchRollout(int bx, int by, int bz, bool pipeline, const kernels::RolloutArgs& args,
                            const SamplerLaunchState& s, hipStream_t stream, std::string& err) override
  {
    if (!blobsReady(err))
      return MPPI_ERR_STATE;
    prepSampler(s);
    {
      mppi_status st = prepBasis(s, stream, err);
      if (st != MPPI_OK)
        return st;
    }
    if constexpr (has_register_form<DYN_T>::value && !std::is_void<DYN_FAST_T>::value)
    {
      if (!dyn.register_form_ && hasShape(FAST_SHAPES{}, bx, by, bz))
      {
        err = "the replicated-lane block shapes of this model exist for its default network shape only: create the "
              "controller with block_y = 1 for a network set through 'lstm_structure'";
        return MPPI_ERR_LAUNCH_SHAPE;
      }
    }
    if (pipeline && supportsPipelineRep(bx, by, bz))
      return launchPipelineRep(args, stream, err);
    if (pipeline && supportsPipelineFold(bx, by, bz))
      return launchPipeline<2, true>(args, stream, err);
    if (pipeline)
      return bz == 1 ? launchPipeline<1>(args, stream, err) : launchPipeline<2>(args, stream, err);
    if constexpr (!std::is_void<DYN_FAST_T>::value)
    {
      bool handled = false;
      mppi_status st = dispatchFast(FAST_SHAPES{}, bx, by, bz, args, stream, err, handled);
      if (handled)
        return st;
    }
    return dispatch(SHAPES{}, bx, by, bz, args, stream, err);
  }

  mppi_status launchFinalize(int D, const kernels::FinalizeArgs& a, hipStream_t stream, std::string& err) override
  {
    if (!blobsReady(err))
      return MPPI_ERR_STATE;
    if constexpr (has_finalize_form<DYN_FAST_T>::value)
    {
      // a form of the model made for ONE rollout on a wave (NN models: lane = neuron, utils/nn_helpers/fnn_wave.hpp): the
      // re-rollout is a chain of T dependent steps, what counts is the length of a step's dependent chain
      // (MPPI_AMD_FINALIZE_FORM=rep: the replicated-lane form instead — tests compare the two)
      const char* form = getenv("MPPI_AMD_FINALIZE_FORM");
      if (!(form && form[0] == 'r'))
      {
        using WAVE_T = typename DYN_FAST_T::FINALIZE_FORM;
        WAVE_T wave_form(dyn);
        const size_t smem_w = kernels::finalizeRepSharedBytes(wave_form, a.num_timesteps, a.scratch_d != nullptr);
        if (smem_w <= MAX_LDS_BYTES)
        {
          auto kw = a.scratch_d ? kernels::finalizeRepKernel<WAVE_T, true> : kernels::finalizeRepKernel<WAVE_T, false>;
          if (smem_w > 48 * 1024)
            (void)ensureDynamicLds(reinterpret_cast<const void*>(kw), smem_w);
          hipLaunchKernelGGL(kw, dim3(D), dim3(64, 1, 1), smem_w, stream, wave_form, a);
          hipError_t e = hipGetLastError();
          if (e != hipSuccess)
          {
            err = std::string("finalizeRepKernel (wave form) launch: ") + hipGetErrorString(e);
            return MPPI_ERR_HIP;
          }
          return MPPI_OK;
        }
      }
    }
    if constexpr (!std::is_void<DYN_FAST_T>::value)
    {  // replicated-lane (MFMA) dynamics: the register-resident single-wave variant
      DYN_FAST_T fast(dyn);
      const size_t smem_rep = kernels::finalizeRepSharedBytes(fast, a.num_timesteps, a.scratch_d != nullptr);
      bool usable = smem_rep <= MAX_LDS_BYTES;
      if constexpr (has_register_form<DYN_T>::value)
        usable = usable && dyn.register_form_;
      if (usable)
      {
        auto krep = a.scratch_d ? kernels::finalizeRepKernel<DYN_FAST_T, true> : kernels::finalizeRepKernel<DYN_FAST_T, false>;
        if (smem_rep > 48 * 1024)
          (void)ensureDynamicLds(reinterpret_cast<const void*>(krep), smem_rep);
        hipLaunchKernelGGL(krep, dim3(D), dim3(64, 1, 1), smem_rep, stream, fast, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
        {
          err = std::string("finalizeRepKernel launch: ") + hipGetErrorString(e);
          return MPPI_ERR_HIP;
        }
        return MPPI_OK;
      }
    }
    const size_t smem = kernels::finalizeSharedBytes(dyn, a.num_timesteps, FIN_BY, a.scratch_d != nullptr);
    if (smem > MAX_LDS_BYTES)
    {
      err = "finalize kernel LDS overflow";
      return MPPI_ERR_LDS_OVERFLOW;
    }
    auto kfn = a.scratch_d ? kernels::finalizeKernel<DYN_T, FIN_BY, true> : kernels::finalizeKernel<DYN_T, FIN_BY, false>;
    if (smem > 48 * 1024)
      (void)ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
    hipLaunchKernelGGL(kfn, dim3(D), dim3(kernels::finalizeBlockX(FIN_BY), FIN_BY, 1), smem, stream, dyn, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
      err = std::string("finalizeKernel launch: ") + hipGetErrorString(e);
      return MPPI_ERR_HIP;
    }
    return MPPI_OK;
  }

  /** plain one-wave finalize form only (FIN_BY == 1, no replicated-lane / wave form): what the one-lane pipeline models have */
  static constexpr bool MERGE_CONTROL = FIN_BY == 1 && std::is_void<DYN_FAST_T>::value;
  bool supportsMergeControl(int num_timesteps) const override
  {
    if constexpr (MERGE_CONTROL)
      return kernels::mergeControlSharedBytes(dyn, num_timesteps) <= MAX_LDS_BYTES;
    return false;
  }
  mppi_status launchMergeControl(const kernels::FinalizeArgs& a, const kernels::MergeControlArgs& m, hipStream_t stream,
                                 std::string& err) override
  {
    if constexpr (MERGE_CONTROL)
    {
      if (!blobsReady(err))
        return MPPI_ERR_STATE;
      if (a.scratch_d || a.phases != 1 || ((a.num_timesteps * DYN_T::CONTROL_DIM) & 3) != 0 || m.num_records > 256)
      {
        err = "mergeControlKernel: control phase of a split hand-over, sequence in LDS, T*C a multiple of 4, <= 256 records";
        return MPPI_ERR_LAUNCH_SHAPE;
      }
      const size_t smem = kernels::mergeControlSharedBytes(dyn, a.num_timesteps);
      auto kfn = kernels::mergeControlKernel<DYN_T>;
      if (smem > 48 * 1024)
        (void)ensureDynamicLds(reinterpret_cast<const void*>(kfn), smem);
      hipLaunchKernelGGL(kfn, dim3(1), dim3(64 * kernels::MERGE_CONTROL_WAVES, 1, 1), smem, stream, dyn, a, m);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess)
      {
        err = std::string("mergeControlKernel launch: ") + hipGetErrorString(e);
        return MPPI_ERR_HIP;
      }
      return MPPI_OK;
    }
    err = "model has no merging control phase";
    return MPPI_ERR_LAUNCH_SHAPE;
  }

  mppi_status launchModelStep(float* x_d, float* u_d, float dt, int enforce, hipStream_t stream,
                              std::string& err) override
  {
    if (!blobsReady(err))
      return MPPI_ERR_STATE;
    constexpr int Sd = DYN_T::STATE_DIM, Cd = DYN_T::CONTROL_DIM, Od = DYN_T::OUTPUT_DIM;
    const size_t smem = calcClassSharedMemSize(&dyn, 1) +
                        sizeof(float) * (3 * math::nearest_multiple_4(Sd) + math::nearest_multiple_4(Cd) +
                                         math::nearest_multiple_4(Od));
    hipLaunchKernelGGL((modelStepKernel<DYN_T, FIN_BY>), dim3(1), dim3(1, FIN_BY, 1), smem, stream, dyn, x_d, u_d, dt,
                       enforce);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
      err = std::string("modelStepKernel launch: ") + hipGetErrorString(e);
      return MPPI_ERR_HIP;
    }
    return MPPI_OK;
  }
};

}  // namespace engine
}  // namespace mppi

#endif
