/**
 * merge_wave.hpp — the arithmetic of the softmin merge for ONE wave, shared by the merge kernel (csrc/reduce_kernels.hpp:
 * combineWave) and by the rollout kernel that merges the PREVIOUS iteration's block records itself while it samples
 * (engine/rollout_pipeline_kernel.hpp, STREAM_MERGE) — one definition, so that both produce the same bits.
 *
 * Lane l of the wave owns records l, l + 64, l + 128, l + 192 (MERGE_LANE_RECORDS = 4: up to 256 records).  Merge rule
 * (SURVEY.md §8e): rho = min_b rho_b; s_b = exp(-(rho_b - rho)/lambda); eta = sum_b s_b eta_b (double);
 * U[j] = sum_b s_b U_b[j]; u*[j] = U[j] / float(eta).  The sums over a lane's records run in ascending order, the sums over the
 * lanes through the DPP all-reduces of wave_ops.hpp (every lane, wave, block and rank gets the same bits).
 */
#ifndef MPPI_AMD_ENGINE_MERGE_WAVE_HPP_
#define MPPI_AMD_ENGINE_MERGE_WAVE_HPP_

#include <hip/hip_runtime.h>
#include <math.h>
#include "mppi_amd/det_math.h"
#include "mppi_amd/utils/wave_ops.hpp"

namespace mppi
{
namespace kernels
{
constexpr int MERGE_COLS = 4;          ///< columns of u* per wave (and per load: one 16-byte quad)
constexpr int MERGE_LANE_RECORDS = 4;  ///< records a lane keeps in registers: up to 256 records without a second pass

/** a record whose rollouts all cost +inf has rho_b = inf and U_b = eta_b = 0: scale 0 (inf - inf would be NaN when the global
 *  minimum is inf as well — then nothing has weight, as with the reference's global baseline) */
__device__ inline float mergeScale(const float rho_b, const float rho, const float lambda_inv)
{
  const float dist = rho_b - rho;
  return (dist == dist) ? mppi::det::exp(-lambda_inv * dist) : 0.0f;
}

/** what the tails of the records give: the global baseline, this lane's scale factors, the normaliser (wave-uniform) */
struct MergeTails
{
  float rho;
  float s[MERGE_LANE_RECORDS];
  double eta, eta2;
  float eta_f;
};

/** rho_b / eta_b / eta2_b: the lane's records' tails (padding records: rho_b = inf, eta_b = eta2_b = 0).
 *  WITH_ETA2 = false: a wave that does not write the statistics skips the sum of w^2 and its all-reduce in double (out.eta2 = 0);
 *  rho, the scale factors and eta are the same instructions on the same data either way. */
template <bool WITH_ETA2 = true>
__device__ inline void mergeTails(const float (&rho_b)[MERGE_LANE_RECORDS], const float (&eta_b)[MERGE_LANE_RECORDS],
                                  const float (&eta2_b)[MERGE_LANE_RECORDS], const float lambda_inv, MergeTails& out)
{
  float m = rho_b[0];
#pragma unroll
  for (int i = 1; i < MERGE_LANE_RECORDS; i++)
    m = fminf(m, rho_b[i]);
  out.rho = mppi::wave::waveAllMin(m);
  double eta = 0.0, eta2 = 0.0;
#pragma unroll
  for (int i = 0; i < MERGE_LANE_RECORDS; i++)
  {
    const float s = mergeScale(rho_b[i], out.rho, lambda_inv);  // 0 for the padding records
    out.s[i] = s;
    eta += (double)s * (double)eta_b[i];
    if (WITH_ETA2)
      eta2 += (double)s * (double)s * (double)eta2_b[i];
  }
  out.eta = mppi::wave::waveAllSum(eta);
  out.eta2 = WITH_ETA2 ? mppi::wave::waveAllSum(eta2) : 0.0;
  out.eta_f = (float)out.eta;
}

/** tot[c] = sum over all records of s_b U_b[col0 + c]; v[i][c]: the lane's records' column values (0 for padding) */
__device__ inline void mergeColumns(const float (&s)[MERGE_LANE_RECORDS], const float (&v)[MERGE_LANE_RECORDS][MERGE_COLS],
                                    float (&tot)[MERGE_COLS])
{
  float acc[MERGE_COLS];
#pragma unroll
  for (int c = 0; c < MERGE_COLS; c++)
    acc[c] = 0.0f;
#pragma unroll
  for (int i = 0; i < MERGE_LANE_RECORDS; i++)
#pragma unroll
    for (int c = 0; c < MERGE_COLS; c++)
      acc[c] += s[i] * v[i][c];
#pragma unroll
  for (int c = 0; c < MERGE_COLS; c++)
    tot[c] = mppi::wave::waveAllSum(acc[c]);
}

/** statistics block of a finished merge, [STATS_STRIDE = 8] floats: baseline, normaliser, free energy and its variance terms
 *  (reference: mppi_common.cu:1065-1081 computeFreeEnergy).  st[6] — the exchange-failure mark (a bounded wait that ran out) —
 *  is STICKY: no kernel clears it, so that a later successful merge cannot erase it before the host has looked; it is not
 *  written here.  One definition for the merge kernel and the merging control phase (finalize_kernel.hpp: mergeControlKernel). */
__device__ inline void mergeStatistics(const float rho, const float eta_f, const double eta2, const float lambda,
                                       const int num_rollouts_total, float* st)
{
  const float K = (float)num_rollouts_total;
  const float norm = eta_f / K;
  const float var = (float)eta2;
  const float fe = -lambda * mppi::det::log(norm) + rho;
  const float fe_var = lambda * (var / K - norm * norm);
  const float weird = fe_var / (norm * mppi::det::sqrt(K));
  st[0] = rho;
  st[1] = eta_f;
  st[2] = fe;
  st[3] = fe_var;
  st[4] = lambda * (weird + 0.5f * (weird * weird));
  st[5] = var;
  st[7] = 0.0f;
}
}  // namespace kernels
}  // namespace mppi

#endif
