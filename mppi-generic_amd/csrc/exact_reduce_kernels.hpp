/**
 * exact_reduce_kernels.hpp — the last stage of an iteration in the REFERENCE'S OWN ARITHMETIC ORDER (opt-in:
 * mppi_set_reduction_mode(h, MPPI_REDUCTION_REFERENCE_ORDER)).
 *
 * The default path (rollout-kernel epilogue + combineKernel, reduce_kernels.hpp) forms u* as a rescale-merge of per-block
 * softmin records: exact in real arithmetic, ~6e-8 away from the reference in fp32 because the sums run in another order.
 * One iteration is far inside the 1e-5 bar either way, but a closed loop on a plant at the limit of grip amplifies ANY
 * difference (BASELINE.md §3 asks for the bar after 100 free-running iterations).  This file reproduces what the reference
 * computes, operation for operation, from the trajectory costs S[D][K] and the clamped samples v[D][K][T][C] in HBM:
 *
 *   rho     first-occurring minimum by linear scan            core/mppi_common.cu:885-900 (computeBestIndex/BaselineCost)
 *   w_k     exp(-(1/lambda) (S_k - rho)), GLOBAL rho          core/mppi_common.cu:958-966 (normExpTransform)
 *   eta     float(sum_k double(w_k)) in index order           core/mppi_common.cu:1055-1063 (computeNormalizer, host)
 *   F       norm += w, var += w*w serially in fp32            core/mppi_common.cu:1065-1081 (computeFreeEnergy, host)
 *   u*      weight = w_k / eta per rollout; cell j sums sum_stride consecutive rollouts serially
 *           (inter += weight * v), then the ceil(K / sum_stride) cells serially
 *                                                             core/mppi_common.cu:1115-1160 (weightedReductionKernel)
 *
 * The serial sums are what they are — K dependent additions; they run as three waves of one block side by side (eta in
 * double, norm and var in float), each walking the weights through LDS with 16-byte broadcast reads (every lane of the wave
 * performs the same additions).  Measured at K = 16384: ~145 us for this kernel (a dependent v_add_f64 every ~20 cycles: the
 * chain's latency, not its issue rate), 196 us per Cartpole iteration against 27 us with the fused reduction, 369 against
 * 197 us for AutoRally-NN (bench.py: reference_order_reduction); the samples' round trip through HBM (2 x K T C x 4 B) is a few
 * microseconds of that.  This mode buys bit-equality with the reference's order, not speed.
 */
#ifndef MPPI_AMD_EXACT_REDUCE_KERNELS_HPP_
#define MPPI_AMD_EXACT_REDUCE_KERNELS_HPP_

#include <hip/hip_runtime.h>
#include <math.h>
#include "mppi_amd/det_math.h"
#include "reduce_kernels.hpp"

namespace mppi
{
namespace kernels
{
constexpr int EXACT_TILE = 8192;  ///< weights per LDS tile: (8192 + 16) x (8 + 4 + 4) B = 128.25 KiB
constexpr int EXACT_PAD = 16;     ///< zeros behind a tile: the serial waves fetch one trip ahead without a bounds test
constexpr size_t EXACT_WEIGHTS_LDS_BYTES = (size_t)(EXACT_TILE + EXACT_PAD) * (sizeof(double) + 2 * sizeof(float));

/** how the weights come about */
struct ExactWeightsArgs
{
  int num_rollouts;
  const float* costs_d;   ///< [D][K]
  float* weights_d;       ///< [D][K] out: w_k (NOT divided by eta — the reduction divides per rollout, as the reference does)
  float* stats_out_d;     ///< [D][STATS_STRIDE]
  float lambda;
  float lambda_inv;       ///< float(1.0 / lambda): mppi_controller.cu:201 narrows the double quotient
  float tsallis_gamma, tsallis_r;  ///< both != 0: TsallisTransform (core/mppi_common.cu:968-985) instead of normExp
};

/**
 * grid = D systems, block = COMBINE_THREADS, dynamic LDS = EXACT_WEIGHTS_LDS_BYTES.
 * Pass 1 (all waves): rho.  Pass 2, tile by tile: all waves compute the tile's weights (-> HBM, and as double / float /
 * squared float -> LDS); then waves 0, 1, 2 each add the tile to their running sum IN INDEX ORDER.
 */
static __global__ void __launch_bounds__(COMBINE_THREADS) exactWeightsKernel(const ExactWeightsArgs a)
{
  extern __shared__ __attribute__((aligned(16))) char exact_smem_raw[];
  double* wd_s = reinterpret_cast<double*>(exact_smem_raw);              // [EXACT_TILE + EXACT_PAD]
  float* wf_s = reinterpret_cast<float*>(wd_s + EXACT_TILE + EXACT_PAD); // [EXACT_TILE + EXACT_PAD]
  float* w2_s = wf_s + EXACT_TILE + EXACT_PAD;                           // [EXACT_TILE + EXACT_PAD]
  __shared__ float red_f[COMBINE_THREADS / 64];
  __shared__ double eta_s;
  __shared__ float norm_s, var_s;

  const int z = blockIdx.x;
  const int K = a.num_rollouts;
  const int tid = threadIdx.x, wave = tid >> 6;
  const float* costs = a.costs_d + (size_t)z * K;
  float* weights = a.weights_d + (size_t)z * K;

  // first-occurring minimum of a linear `<` scan: its VALUE is the minimum over the non-NaN costs, unless costs[0] is NaN
  // (then no comparison ever succeeds and the baseline stays NaN)
  float m = INFINITY;
  for (int i = tid; i < K; i += COMBINE_THREADS)
    m = fminf(m, costs[i]);
  m = blockMin(m, red_f);
  const float c0 = costs[0];
  const float rho = (c0 != c0) ? c0 : m;
  const bool tsallis = a.tsallis_gamma != 0.0f && a.tsallis_r != 0.0f;

  double eta = 0.0;
  float norm = 0.0f, var = 0.0f;
  for (int base = 0; base < K; base += EXACT_TILE)
  {
    const int n = min(EXACT_TILE, K - base);
    const int n_pad = (n + 15) & ~15;  // the serial waves walk 16 weights per trip; +0 leaves every sum unchanged
    for (int i = tid; i < n_pad + EXACT_PAD; i += COMBINE_THREADS)
    {
      float w = 0.0f;
      if (i < n)
      {
        const float cost_dif = costs[base + i] - rho;
        if (tsallis)
          w = cost_dif < a.tsallis_gamma ?
                  mppi::det::exp(mppi::det::log(1.0f - cost_dif / a.tsallis_gamma) / (a.tsallis_r - 1.0f)) :
                  0.0f;
        else
          w = mppi::det::exp(-a.lambda_inv * cost_dif);
        weights[base + i] = w;
      }
      wd_s[i] = (double)w;
      wf_s[i] = w;
      w2_s[i] = w * w;
    }
    __syncthreads();
    if (wave == 0)
    {  // eta: double accumulate in index order
      const double2* p = reinterpret_cast<const double2*>(wd_s);
      double2 cur[8], nxt[8];
#pragma unroll
      for (int i = 0; i < 8; i++)
        cur[i] = p[i];
      for (int b = 0; b < n_pad; b += 16)
      {
        // the next trip's sixteen weights are on their way while this trip's are added (zeros behind the tile: no bounds test)
#pragma unroll
        for (int i = 0; i < 8; i++)
          nxt[i] = p[(b + 16) / 2 + i];
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
          eta += cur[i].x;
          eta += cur[i].y;
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
          cur[i] = nxt[i];
      }
    }
    else if (wave == 1 || wave == 2)
    {  // computeFreeEnergy's two fp32 running sums, one wave each
      const float4* p = reinterpret_cast<const float4*>(wave == 1 ? wf_s : w2_s);
      float acc = wave == 1 ? norm : var;
      float4 cur[4], nxt[4];
#pragma unroll
      for (int i = 0; i < 4; i++)
        cur[i] = p[i];
      for (int b = 0; b < n_pad; b += 16)
      {
#pragma unroll
        for (int i = 0; i < 4; i++)
          nxt[i] = p[(b + 16) / 4 + i];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
          acc += cur[i].x;
          acc += cur[i].y;
          acc += cur[i].z;
          acc += cur[i].w;
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
          cur[i] = nxt[i];
      }
      if (wave == 1)
        norm = acc;
      else
        var = acc;
    }
    __syncthreads();
  }
  if (tid == 0)
    eta_s = eta;
  if (tid == 64)
    norm_s = norm;
  if (tid == 128)
    var_s = var;
  __syncthreads();
  if (tid == 0)
  {
    // reference: mppi_common.cu:1065-1081 computeFreeEnergy (logf -> det::log, sqrtf -> det::sqrt)
    const float Kf = (float)K;
    const float eta_f = (float)eta_s;
    float nrm = norm_s;
    nrm /= Kf;
    const float fe = -a.lambda * mppi::det::log(nrm) + rho;
    const float fe_var = a.lambda * (var_s / Kf - nrm * nrm);
    const float weird = fe_var / (nrm * mppi::det::sqrt(1.0f * Kf));
    float* st = a.stats_out_d + (size_t)z * STATS_STRIDE;
    st[0] = rho;
    st[1] = eta_f;
    st[2] = fe;
    st[3] = fe_var;
    st[4] = a.lambda * (weird + 0.5f * (weird * weird));
    st[5] = var_s;
    st[6] = 0.0f;
    st[7] = 0.0f;
  }
}

/**
 * strideControlWeightReduction (core/mppi_common.cu:1115-1135): cell = sum_stride consecutive rollouts, summed serially
 * with weight = w_k / eta formed per rollout.  grid = (ceil(T C / 64), ceil(cells / 16), D), block = 1024: wave w = one
 * cell, lane = one column of v (a row of v is contiguous: the 64 lanes of a wave read 256 contiguous bytes per rollout).
 * FMA != 0: inter = fma(weight, v, inter) — what nvcc emits for `inter += weight * v` under its default -fmad=true;
 * FMA == 0: product rounded, then added — the reference's own CPU statement of the kernel and the oracle.
 * inter_d: [D][cells][T C].
 */
template <int FMA>
__global__ void __launch_bounds__(COMBINE_THREADS)
    exactReductionCellsKernel(const float* __restrict__ weights_d, const float* __restrict__ v_d,
                              const float* __restrict__ stats_d, int TC, int num_rollouts, int sum_stride, int cells,
                              float* __restrict__ inter_d)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int z = blockIdx.z;
  const int cell = blockIdx.y * (COMBINE_THREADS / 64) + wave;
  const int j = blockIdx.x * 64 + lane;
  if (cell >= cells || j >= TC)
    return;
  const float normalizer = stats_d[(size_t)z * STATS_STRIDE + 1];
  const float* w = weights_d + (size_t)z * num_rollouts;
  const float* v = v_d + (size_t)z * num_rollouts * TC + j;
  float acc = 0.0f;
  const int k0 = cell * sum_stride;
  const int k1 = min(num_rollouts, k0 + sum_stride);
#pragma unroll 8
  for (int k = k0; k < k1; k++)
  {
    const float weight = w[k] / normalizer;
    const float s = v[(size_t)k * TC];
    if (FMA)
      acc = mppi::det::fma(weight, s, acc);
    else
      acc += weight * s;
  }
  inter_d[((size_t)z * cells + cell) * TC + j] = acc;
}

/** rolloutWeightReductionAndSaveControl (core/mppi_common.cu:1138-1160): thread 0's serial sum over the cells.
 *  grid = (ceil(T C / 64), D), block = 64: lane = column. */
static __global__ void __launch_bounds__(64)
    exactReductionFinalKernel(const float* __restrict__ inter_d, int TC, int cells, float* __restrict__ mean_out_d)
{
  const int z = blockIdx.y;
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= TC)
    return;
  const float* p = inter_d + (size_t)z * cells * TC + j;
  float acc = 0.0f;
#pragma unroll 16
  for (int i = 0; i < cells; i++)
    acc += p[(size_t)i * TC];
  mean_out_d[(size_t)z * TC + j] = acc;
}

}  // namespace kernels
}  // namespace mppi

#endif
