/**
 * gaussian.hpp — GaussianDistribution sampler plugin, MI355X design.
 *
 * Replaces, on the device side (reference paths relative to include/mppi/sampling_distributions/):
 *   generateSamples + setGaussianControls      gaussian/gaussian.cu:374-431, :17-277   (a1, a2 in SURVEY.md §8a)
 *   readControlSample / writeControlSample     sampling_distribution.cu:169-205, 278-314   (a4)
 *   computeLikelihoodRatioCost (device)        gaussian/gaussian.cu:480-569                (a5)
 *   the sample-side of updateDistributionParamsFromDevice (weighted reduction input)      gaussian/gaussian.cu:433-457
 *
 * Reference data flow: cuRAND writes eps[K][T][C] to HBM, setGaussianControls rewrites it as v = mu + sigma*eps, the
 * rollout reads v with a T*C*4-byte stride between neighbouring threads (uncoalesced), clamps, writes v back, and the
 * weighted reduction reads all of v once more with the same stride: >= 6 passes over V.
 *
 * Here the samples of a block never leave the CU: the sampler's per-rollout LDS request (getBlkSharedSizeBytes) IS the
 * sample row v[k][0..T*C) (+1 float of padding when T*C is even, so that the 64 lanes of a wave, which walk the rows at
 * the same t, hit 64 different banks).  Where eps comes from:
 *   - single-lane rollouts (blockDim.y == 1), Philox: drawn INSIDE the step loop, one Philox4x32-10 quad per 4 row
 *     elements into registers (drawQuad + shapeControlSample).  The draw does not depend on the state, so its instructions
 *     fill the issue slots the dependent dynamics chain leaves empty;
 *   - otherwise initializeDistributions() fills the rows first: Philox by all threads of the block, or — parity /
 *     rocRAND-host mode — a coalesced copy from the eps buffer in HBM.
 * readControlSample*() applies the setGaussianControls rule on the fly, writeControlSample() stores the clamped control
 * into the row, and after the rollout the kernel's epilogue forms the block's weighted partial sum from the rows.
 * The special-trajectory rules use the GLOBAL rollout index (rollout_offset_ + local index) so they survive sharding.
 *
 * Same names and argument meaning as the reference's SamplingDistribution device interface
 * (sampling_distribution.cuh:32-430); `sample_index` is the rollout index local to this GPU.
 */
#ifndef MPPI_AMD_GAUSSIAN_DISTRIBUTION_HPP_
#define MPPI_AMD_GAUSSIAN_DISTRIBUTION_HPP_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mppi_amd/plugin/managed.hpp"
#include "mppi_amd/plugin/dynamics.hpp"
#include "mppi_amd/philox.h"

namespace mppi
{
namespace sampling_distributions
{
enum NoiseSource : int
{
  NOISE_PHILOX_FUSED = 0,  ///< eps drawn inside the rollout kernel (default)
  NOISE_EPS_BUFFER = 1,    ///< eps read from eps_d_[K_local][T][C] (mppi_inject_noise, or the rocRAND host-API fill)
};

/** reference: sampling_distribution.cuh:17-30 (SamplingParams) + gaussian/gaussian.cuh:21-61 (GaussianParamsImpl) */
template <int C_DIM, int MAX_DISTRIBUTIONS_T = 2>
struct GaussianParamsImpl
{
  static const int CONTROL_DIM = C_DIM;
  static const int MAX_DISTRIBUTIONS = MAX_DISTRIBUTIONS_T;
  bool use_same_noise_for_all_distributions = true;
  int num_rollouts = 1;  ///< rollouts on THIS GPU
  int num_timesteps = 1;
  int num_distributions = 1;
  float std_dev[C_DIM * MAX_DISTRIBUTIONS_T];
  float control_cost_coeff[C_DIM];
  float pure_noise_trajectories_percentage = 0.01f;
  float std_dev_decay = 1.0f;
  int sum_strides = 32;  ///< kept for API compatibility; the block-local reduction does not use it
  bool time_specific_std_dev = false;

  GaussianParamsImpl(int num_rollouts = 1, int num_timesteps = 1, int num_distributions = 1)
    : num_rollouts(num_rollouts), num_timesteps(num_timesteps), num_distributions(num_distributions)
  {
    for (int i = 0; i < C_DIM * MAX_DISTRIBUTIONS_T; i++)
      std_dev[i] = 1.0f;
    for (int i = 0; i < C_DIM; i++)
      control_cost_coeff[i] = 0.0f;
  }
};

template <class DYN_PARAMS_T>
class GaussianDistribution : public Managed
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  static const int CONTROL_DIM = E_INDEX(DYN_PARAMS_T::ControlIndex, NUM_CONTROLS);
  typedef GaussianParamsImpl<CONTROL_DIM, 2> SAMPLING_PARAMS_T;
  typedef GaussianDistribution<DYN_PARAMS_T> SAMPLING_T;
  static constexpr bool IN_LOOP_DRAW = true;  ///< single-lane rollouts may draw eps inside the step loop
  static constexpr bool COLORED = false;
  /** the fused rollout kernel can keep this sampler's rows in global memory when they do not fit the LDS (long horizons) */
  static constexpr bool SUPPORTS_GLOBAL_ROWS = true;

  SAMPLING_PARAMS_T params_;

  /* device buffers (owned by the engine) */
  float* control_means_d_ = nullptr;      ///< mu [D][T][C]              (reference: gaussian.cuh control_means_d_)
  const float* eps_d_ = nullptr;          ///< eps [K_local][T][C], NOISE_EPS_BUFFER only
  float* control_samples_d_ = nullptr;    ///< optional dump of the clamped samples v [D][K_local][T][C]
  /**
   * Long horizons: the sample rows of a block, [blocks][slots][rowStride], in HBM instead of LDS (the reference keeps its
   * samples in global memory always and has no horizon limit, sampling_distribution.cu:169-205).  nullptr: rows in LDS.
   * The kernels address the rows through sampleRow() / theta_d either way; a lane's accesses walk its own row (one cache
   * line serves 16 / C steps), the weighted reduction of the epilogue reads them coalesced across the columns.
   */
  float* rows_global_d_ = nullptr;

  /* per-call state, refreshed by the engine before every launch (the engine passes the object by value) */
  int noise_source_ = NOISE_PHILOX_FUSED;
  uint64_t seed_ = 0;
  uint32_t generation_ = 0;          ///< number of generateSamples calls so far (cuRAND's advancing offset)
  int optimization_stride_ = 0;
  float std_dev_decayed_[CONTROL_DIM * 2];  ///< std_dev_decay^iteration * std_dev  (gaussian.cu:421, :86)
  /** time_specific_std_dev (gaussian.cu:21-43; GaussianTimeVaryingStdDevParams, gaussian.cuh:64-95): sigma[d][t][c] in device
   *  memory, undecayed; nullptr: one sigma per distribution and control (params_.std_dev) */
  const float* std_dev_time_d_ = nullptr;
  float decay_now_ = 1.0f;                  ///< std_dev_decay^iteration
  int rollout_offset_ = 0;           ///< global index of this GPU's first rollout
  int num_rollouts_global_ = 1;      ///< K over all GPUs
  /* thread mapping, set per thread by kernels whose blockDim.x is not "one thread per rollout" (replicated lanes):
   * the kernel's copy of the object is private to the thread, so these are ordinary registers */
  unsigned noise_stream_ = 0;        ///< Philox stream of this thread's draws: 0, or the distribution index when
                                     ///< use_same_noise_for_all_distributions is off (setNoiseStream)
  int thread_slot_ = -1;             ///< rollout slot of this thread, -1: blockDim.x * threadIdx.z + threadIdx.x
  int block_rollouts_ = 0;           ///< rollouts per block, 0: blockDim.x
  int block_systems_ = 0;            ///< systems per block, 0: blockDim.z
  float* staging_lds_ = nullptr;     ///< base of this sampler's LDS region (setStagingBase; thread-private register)

  __device__ inline void setThreadMapping(int slot, int block_rollouts, int block_systems = 0)
  {
    thread_slot_ = slot;
    block_rollouts_ = block_rollouts;
    block_systems_ = block_systems;
  }
  /** use_same_noise_for_all_distributions == false (gaussian.cu:378-394: one curandGenerateNormal over all distributions
   *  instead of a copy of distribution 0's noise): distribution d draws Philox stream d / reads slab d of the eps buffer */
  __device__ inline bool independentNoise() const
  {
    return !params_.use_same_noise_for_all_distributions;
  }
  __device__ inline void setNoiseStream(const int distribution_index)
  {
    noise_stream_ = independentNoise() ? (unsigned)distribution_index : 0u;
  }
  __device__ inline int systemsPerBlock() const
  {
    return block_systems_ > 0 ? block_systems_ : (int)__builtin_amdgcn_workgroup_size_z();
  }
  __device__ inline int slotOfThread() const
  {
    return thread_slot_ >= 0 ? thread_slot_ : (int)(blockDim.x * threadIdx.z + threadIdx.x);
  }
  __device__ inline int rolloutsPerBlock() const
  {
    return block_rollouts_ > 0 ? block_rollouts_ : (int)__builtin_amdgcn_workgroup_size_x();
  }

  GaussianDistribution(hipStream_t stream = 0)
  {
    bindToStream(stream);
  }
  /** reference: sampling_distribution.cuh:60-66 (construct from a parameter struct), :120-135 (setParams / getParams) */
  GaussianDistribution(const SAMPLING_PARAMS_T& params, hipStream_t stream = 0) : params_(params)
  {
    bindToStream(stream);
  }
  void setParams(const SAMPLING_PARAMS_T& params)
  {
    params_ = params;
  }
  __host__ __device__ SAMPLING_PARAMS_T getParams() const
  {
    return params_;
  }

  /** row stride in floats: T*C, +1 when even (bank-conflict-free column walk) */
  __host__ __device__ static inline int rowStride(int num_timesteps)
  {
    const int n = num_timesteps * CONTROL_DIM;
    return (n & 1) ? n : n + 1;
  }
  /** rows in HBM (rows_global_d_): every row starts on a 128-byte line — a lane that fills its row a few steps at a time then
   *  completes whole lines instead of straddling two (round 5: the Robust MPPI kernel's write-back) */
  __host__ __device__ static inline int rowStrideGlobal(int num_timesteps)
  {
    return (num_timesteps * CONTROL_DIM + 31) & ~31;
  }
  /** the stride of THIS launch's rows: LDS rows or HBM rows */
  __host__ __device__ inline int rowStrideNow() const
  {
    return rows_global_d_ ? rowStrideGlobal(params_.num_timesteps) : rowStride(params_.num_timesteps);
  }
  /** LDS request per rollout slot: one sample row (reference: managed.cuh:104-111 Blk request) */
  __host__ __device__ inline int getBlkSharedSizeBytes() const
  {
    return rows_global_d_ ? 0 : rowStride(params_.num_timesteps) * (int)sizeof(float);
  }
  /** where the rows of block `block_idx` (slots_per_block rows) live: the LDS region the kernel reserved, or the HBM buffer */
  __device__ inline float* blockRows(float* theta_d_lds, const int block_idx, const int slots_per_block) const
  {
    return rows_global_d_ ? rows_global_d_ + (size_t)block_idx * slots_per_block * rowStrideGlobal(params_.num_timesteps) : theta_d_lds;
  }
  __host__ __device__ inline int getGrdSharedSizeBytes() const
  {
    return 0;
  }
  /** the LDS region the kernel reserved for this sampler ([slot rows][block-shared part]); with the rows in HBM the
   *  block-shared part is all that is left there — samplers that use it (colored noise) remember the base */
  __device__ inline void setStagingBase(float* theta_d_lds)
  {
    staging_lds_ = theta_d_lds;
  }

  /** decay^iter by repeated multiplication (exact for decay == 1; the CPU oracle does the same) */
  void setIteration(int iteration_num, int optimization_stride)
  {
    float decay = 1.0f;
    for (int i = 0; i < iteration_num; i++)
      decay *= params_.std_dev_decay;
    for (int i = 0; i < CONTROL_DIM * 2; i++)
      std_dev_decayed_[i] = decay * params_.std_dev[i];
    decay_now_ = decay;
    optimization_stride_ = optimization_stride;
  }

  __device__ inline float* sampleRow(float* theta_d, int slot) const
  {
    return theta_d + slot * rowStrideNow();
  }

  __device__ inline bool isPureNoise(int sample_index) const
  {
    // reference: gaussian.cu:108 / :512 — float compare of the index against (1 - p) * K, GLOBAL index and K
    return (float)(sample_index + rollout_offset_) >=
           (1.0f - params_.pure_noise_trajectories_percentage) * (float)num_rollouts_global_;
  }

  /** true when the kernel draws eps inside the step loop instead of pre-filling the rows */
  __device__ inline bool drawsInLoop() const
  {
    return noise_source_ == NOISE_PHILOX_FUSED && __builtin_amdgcn_workgroup_size_y() == 1;
  }

  /**
   * Fills the block's sample rows with eps (all threads of the block cooperate; block-uniform control flow), unless
   * the kernel draws in the loop.  Rows are consecutive rollouts, so in buffer mode the block's eps is one contiguous
   * span of blockDim.x * T*C floats.
   */
  __device__ inline void initializeDistributions(const float* __restrict__ output, const float t_0, const float dt,
                                                 float* __restrict__ theta_d)
  {
    if (drawsInLoop())
      return;
    const int TC = params_.num_timesteps * CONTROL_DIM;
    const int stride = rowStrideNow();
    const int bx = rolloutsPerBlock();
    const int tid_flat = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
    const int nthreads = (int)(blockDim.x * blockDim.y * blockDim.z);
    const int row0 = (int)(blockIdx.x * bx);  // first local rollout of the block
    const int nrows = min(bx, params_.num_rollouts - row0);
    const int nz = systemsPerBlock();
    if (nrows <= 0)
      return;
    if (noise_source_ == NOISE_EPS_BUFFER)
    {
      const int total = nrows * TC;  // floats in the block's eps span
      const float* __restrict__ src = eps_d_ + (size_t)row0 * TC;
      int row = 0, col = tid_flat;
      while (col >= TC)
      {
        col -= TC;
        row++;
      }
      const size_t slab = independentNoise() ? (size_t)params_.num_rollouts * TC : 0;  // eps [D][K_local][T][C]
      for (int e = tid_flat; e < total; e += nthreads)
      {
        for (int z = 0; z < nz; z++)
          theta_d[(z * bx + row) * stride + col] = src[e + z * slab];
        col += nthreads;
        while (col >= TC)
        {
          col -= TC;
          row++;
        }
      }
    }
    else
    {
      const int qpr = (TC + 3) >> 2;  // quads per row
      const int nquads = nrows * qpr;
      for (int i = tid_flat; i < nquads; i += nthreads)
      {
        const int row = i / qpr;
        const int q = i - row * qpr;
        for (int z = 0; z < (independentNoise() ? nz : 1); z++)
        {
          float zn[4];
          mppi::rng::normal4(seed_, generation_, (uint32_t)z, (uint32_t)(row0 + row + rollout_offset_), (uint32_t)q, zn);
#pragma unroll
          for (int l = 0; l < 4; l++)
          {
            const int col = q * 4 + l;
            if (col < TC)
            {
              if (independentNoise())
                theta_d[(z * bx + row) * stride + col] = zn[l];
              else
                for (int zz = 0; zz < nz; zz++)
                  theta_d[(zz * bx + row) * stride + col] = zn[l];
            }
          }
        }
      }
    }
  }

  /**
   * Per-distribution table entry / mean.  LANE_D: the distribution index differs between the lanes of a wave (the folded
   * Tube kernel carries both systems of a rollout in one wave).  Both rows are then fetched with wave-uniform loads and
   * the lane selects: indexing the kernel-argument struct with a per-lane value would push it into scratch memory, and
   * per-lane mean pointers would turn scalar loads into flat loads inside the step loop.
   */
  template <bool LANE_D>
  __device__ inline float distValue(const float* __restrict__ table, const int d, const int j) const
  {
    if constexpr (LANE_D)
    {
      const float a = table[j], b = table[CONTROL_DIM + j];
      return d == 0 ? a : b;
    }
    else
    {
      return table[CONTROL_DIM * d + j];
    }
  }
  /** The means are written by other kernels only (finalize / setters), never during a rollout: reading them through
   *  the constant address space tells the compiler so — wave-uniform addresses then become scalar loads that may be
   *  hoisted above the pipeline's acquire fences instead of per-step vector loads with a full L2 round trip. */
  typedef const __attribute__((address_space(4))) float* const_float_ptr;
  template <bool LANE_D>
  __device__ inline float meanValue(const int d, const int t, const int j) const
  {
    const_float_ptr m0 = (const_float_ptr)(control_means_d_ + (size_t)t * CONTROL_DIM);
    if constexpr (LANE_D)
    {
      const_float_ptr m1 = m0 + (params_.num_distributions > 1 ? params_.num_timesteps * CONTROL_DIM : 0);
      const float a = m0[j], b = m1[j];
      return d == 0 ? a : b;
    }
    else
    {
      return m0[(size_t)params_.num_timesteps * d * CONTROL_DIM + j];
    }
  }

  /** sigma of (distribution d, step t, control j): the per-step table when time_specific_std_dev is on (a wave-uniform
   *  branch; like the means, the table is only written between launches), else the per-distribution value.  DECAYED: times
   *  std_dev_decay^iteration, as setGaussianControls applies it (gaussian.cu:84-87); the likelihood-ratio cost reads the
   *  undecayed value (gaussian.cu:488-493). */
  template <bool LANE_D, bool DECAYED>
  __device__ inline float sigmaValue(const int d, const int t, const int j) const
  {
    if (std_dev_time_d_)
    {
      const_float_ptr s0 = (const_float_ptr)(std_dev_time_d_ + (size_t)t * CONTROL_DIM);
      float v;
      if constexpr (LANE_D)
      {
        const_float_ptr s1 = s0 + (params_.num_distributions > 1 ? params_.num_timesteps * CONTROL_DIM : 0);
        const float a = s0[j], b = s1[j];
        v = d == 0 ? a : b;
      }
      else
      {
        v = s0[(size_t)params_.num_timesteps * d * CONTROL_DIM + j];
      }
      return DECAYED ? decay_now_ * v : v;
    }
    return DECAYED ? distValue<LANE_D>(std_dev_decayed_, d, j) : distValue<LANE_D>(params_.std_dev, d, j);
  }

  /** the setGaussianControls rule (gaussian.cu:99-127), branch-free */
  __device__ inline float shapeSample(float m, float sd, float e, bool use_mean, bool pure) const
  {
    const float se = sd * e;
    const float full = m + se;
    return use_mean ? m : (pure ? se : full);
  }

  /** quad `quad` (row elements 4*quad .. 4*quad+3) of local rollout `sample_index`, into registers */
  __device__ inline void drawQuad(const int sample_index, const int quad, float z[4]) const
  {
    mppi::rng::normal4(seed_, generation_, noise_stream_, (uint32_t)(sample_index + rollout_offset_), (uint32_t)quad, z);
  }

  /**
   * readControlSample for a rollout that owns one lane, with eps[CONTROL_DIM] already in registers (in-loop Philox
   * draw): applies the setGaussianControls rule.  Everything but eps is wave-uniform.
   */
  template <bool LANE_D = false>
  __device__ inline void shapeControlSample(const int sample_index, const int t, const int distribution_index,
                                            const float* __restrict__ eps, float* __restrict__ control) const
  {
    const int d = distribution_index >= params_.num_distributions ? 0 : distribution_index;
    const bool use_mean = ((sample_index + rollout_offset_) == 0) || (t < optimization_stride_);
    const bool pure = isPureNoise(sample_index);
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      control[i] = shapeSample(meanValue<LANE_D>(d, t, i), sigmaValue<LANE_D, true>(d, t, i), eps[i], use_mean,
                               pure);
  }

  /** the same with the means handed in (a kernel that has just merged them itself: rolloutPipelineKernel, STREAM_MERGE) */
  template <bool LANE_D = false>
  __device__ inline void shapeControlSampleMean(const int sample_index, const int t, const int distribution_index,
                                                const float* __restrict__ eps, const float* __restrict__ mean_vals,
                                                float* __restrict__ control) const
  {
    const int d = distribution_index >= params_.num_distributions ? 0 : distribution_index;
    const bool use_mean = ((sample_index + rollout_offset_) == 0) || (t < optimization_stride_);
    const bool pure = isPureNoise(sample_index);
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      control[i] = shapeSample(mean_vals[i], sigmaValue<LANE_D, true>(d, t, i), eps[i], use_mean, pure);
  }

  /**
   * Random access to one shaped sample v[d][sample_index][t][:] without the block's LDS rows: eps comes from the eps
   * buffer or from the Philox quad that holds the element.  Used by the Robust-MPPI init-eval kernel, whose rollouts
   * read sample `candidate_sample_idx` at the time-shifted index min(t + stride, T - 1)
   * (reference: core/rmppi_kernels.cu:309-312 readControlSample(candidate_sample_idx, candidate_t, ...)).
   */
  /** the last Philox quad a lane drew through sampleAt(): consecutive elements of a row share a quad (a step of a
   *  two-control model is half of one), so a caller that walks a row keeps one of these across its calls */
  struct QuadCache
  {
    int quad = -1;
    float z[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
  };
  __device__ inline void sampleAt(const int sample_index, const int t, const int distribution_index,
                                  float* __restrict__ control, QuadCache* cache = nullptr) const
  {
    const int d = distribution_index >= params_.num_distributions ? 0 : distribution_index;
    const float* mean = control_means_d_ + (size_t)(params_.num_timesteps * d + t) * CONTROL_DIM;
    const bool use_mean = ((sample_index + rollout_offset_) == 0) || (t < optimization_stride_);
    const bool pure = isPureNoise(sample_index);
    QuadCache local;
    QuadCache& qc = cache ? *cache : local;
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
    {
      const int e = t * CONTROL_DIM + i;
      float eps;
      if (noise_source_ == NOISE_EPS_BUFFER)
      {
        const size_t slab = independentNoise() ? (size_t)d * params_.num_rollouts * params_.num_timesteps * CONTROL_DIM : 0;
        eps = eps_d_[slab + (size_t)sample_index * params_.num_timesteps * CONTROL_DIM + e];
      }
      else
      {
        if ((e >> 2) != qc.quad)
        {  // (rows of one distribution and one rollout only: the cache is keyed by the quad index)
          qc.quad = e >> 2;
          mppi::rng::normal4(seed_, generation_, independentNoise() ? (uint32_t)d : 0u,
                             (uint32_t)(sample_index + rollout_offset_), (uint32_t)(e >> 2), qc.z);
        }
        eps = (e & 3) == 0 ? qc.z[0] : ((e & 3) == 1 ? qc.z[1] : ((e & 3) == 2 ? qc.z[2] : qc.z[3]));
      }
      control[i] = shapeSample(mean[i], sigmaValue<false, true>(d, t, i), eps, use_mean, pure);
    }
  }

  /**
   * reference: sampling_distribution.cu:169-205 (readControlSample) fused with the setGaussianControls rule
   * (gaussian.cu:99-127): k == 0 or t < stride -> mu; pure-noise rollouts -> sigma*eps; else mu + sigma*eps.
   */
  template <bool LANE_D = false>
  __device__ inline void readControlSample(const int& sample_index, const int& t, const int& distribution_index,
                                           float* __restrict__ control, float* __restrict__ theta_d,
                                           const int& block_size, const int& thread_index,
                                           const float* __restrict__ output = nullptr)
  {
    const int d = distribution_index >= params_.num_distributions ? 0 : distribution_index;
    const int slot = slotOfThread();
    const float* row = sampleRow(theta_d, slot) + t * CONTROL_DIM;
    const bool use_mean = ((sample_index + rollout_offset_) == 0) || (t < optimization_stride_);
    const bool pure = isPureNoise(sample_index);
    for (int i = thread_index; i < CONTROL_DIM; i += block_size)
    {
      control[i] = shapeSample(meanValue<LANE_D>(d, t, i), sigmaValue<LANE_D, true>(d, t, i), row[i], use_mean,
                               pure);
    }
  }

  /** reference: sampling_distribution.cu:278-314 — the clamped control replaces the sample (mppi_common.cu:110-117) */
  __device__ inline void writeControlSample(const int& sample_index, const int& t, const int& distribution_index,
                                            const float* __restrict__ control, float* __restrict__ theta_d,
                                            const int& block_size, const int& thread_index,
                                            const float* __restrict__ output = nullptr)
  {
    const int slot = slotOfThread();
    float* row = sampleRow(theta_d, slot) + t * CONTROL_DIM;
    for (int i = thread_index; i < CONTROL_DIM; i += block_size)
    {
      row[i] = control[i];
    }
  }

  /**
   * reference: gaussian.cu:480-569, device flavour: 0.5*lambda*(1-alpha) * sum_i coeff_i*mu_i*(mu_i - 2u_i)/sigma_i^2
   * with mu := 0 on pure-noise rollouts; vector-lane accumulation order of the CONTROL_DIM % 4 / % 2 / scalar branches.
   */
  template <bool LANE_D = false, bool MEAN_ROW = false>
  __device__ inline float computeLikelihoodRatioCost(const float* __restrict__ u, float* __restrict__ theta_d,
                                                     const int sample_index, const int t, const int distribution_idx,
                                                     const float lambda = 1.0f, const float alpha = 0.0f,
                                                     const float* __restrict__ mean_row = nullptr)
  {  // MEAN_ROW: the means of distribution 0 from `mean_row` ([T][C], LDS) instead of control_means_d_ (STREAM_MERGE kernels)
    const int d = distribution_idx >= params_.num_distributions ? 0 : distribution_idx;
    const float* control_cost_coeff = params_.control_cost_coeff;
    const bool pure = isPureNoise(sample_index);
    // coeff == 0 for every control (the reference's default, gaussian.cuh:26): each term is exactly +-0 and the sum does
    // not change the running cost, so the divisions are skipped (uniform branch)
    bool any_coeff = false;
#pragma unroll
    for (int j = 0; j < CONTROL_DIM; j++)
      any_coeff |= (control_cost_coeff[j] != 0.0f);
    if (!any_coeff)
      return 0.0f;
    float cost = 0.0f;
    int i = (int)__builtin_amdgcn_workitem_id_y();
    const int step = (int)__builtin_amdgcn_workgroup_size_y();
    constexpr int W = (CONTROL_DIM % 4 == 0) ? 4 : ((CONTROL_DIM % 2 == 0) ? 2 : 1);
    if constexpr (W > 1)
    {
      float lane[W];
#pragma unroll
      for (int l = 0; l < W; l++)
        lane[l] = 0.0f;
      for (; i < CONTROL_DIM / W; i += step)
      {
#pragma unroll
        for (int l = 0; l < W; l++)
        {
          const int j = i * W + l;
          const float mu = MEAN_ROW ? mean_row[t * CONTROL_DIM + j] : meanValue<LANE_D>(d, t, j);  // unconditional: a wave-uniform load
          const float mean_i = pure ? 0.0f : mu;
          const float sd = sigmaValue<LANE_D, false>(d, t, j);
          lane[l] += control_cost_coeff[j] * mean_i * (mean_i - 2.0f * u[j]) / (sd * sd);
        }
      }
      if constexpr (W == 4)
        cost += lane[0] + lane[1] + lane[2] + lane[3];
      else
        cost += lane[0] + lane[1];
    }
    else
    {
      for (; i < CONTROL_DIM; i += step)
      {
        const float mu = MEAN_ROW ? mean_row[t * CONTROL_DIM + i] : meanValue<LANE_D>(d, t, i);  // unconditional: a wave-uniform load
        const float mean_i = pure ? 0.0f : mu;
        const float sd = sigmaValue<LANE_D, false>(d, t, i);
        cost += control_cost_coeff[i] * mean_i * (mean_i - 2.0f * u[i]) / (sd * sd);
      }
    }
    return 0.5f * lambda * (1.0f - alpha) * cost;
  }

  /** reference: gaussian.cu:571-629 */
  template <bool LANE_D = false>
  __device__ inline float computeFeedbackCost(const float* __restrict__ u_fb, float* __restrict__ theta_d, const int t,
                                              const int distribution_idx, const float lambda = 1.0f,
                                              const float alpha = 0.0f)
  {
    const int d = distribution_idx >= params_.num_distributions ? 0 : distribution_idx;
    float cost = 0.0f;
    constexpr int W = (CONTROL_DIM % 4 == 0) ? 4 : ((CONTROL_DIM % 2 == 0) ? 2 : 1);
    float lane[W];
    for (int l = 0; l < W; l++)
      lane[l] = 0.0f;
    for (int i = (int)threadIdx.y; i < CONTROL_DIM / W; i += (int)blockDim.y)
      for (int l = 0; l < W; l++)
      {
        const int j = i * W + l;
        const float sd = sigmaValue<LANE_D, false>(d, t, j);  // the per-step table when time_specific_std_dev (:579-583)
        lane[l] += params_.control_cost_coeff[j] * (u_fb[j] * u_fb[j]) / (sd * sd);
      }
    for (int l = 0; l < W; l++)
      cost += lane[l];
    return 0.5f * lambda * (1.0f - alpha) * cost;
  }
};

}  // namespace sampling_distributions
}  // namespace mppi

#endif
