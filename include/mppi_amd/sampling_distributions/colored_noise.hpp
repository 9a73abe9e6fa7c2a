/**
 * colored_noise.hpp — ColoredNoiseDistribution sampler plugin (power-law "colored" control noise), MI355X design.
 *
 * Replaces (reference paths relative to include/mppi/sampling_distributions/):
 *   ColoredNoiseDistributionImpl::generateSamples     colored_noise/colored_noise.cu:285-392   (a3 in SURVEY.md §8a)
 *   configureFrequencyNoise / rearrangeNoise           colored_noise/colored_noise.cu:12-56
 *   the cuFFT C2R batch of length 2T                   colored_noise/colored_noise.cu:279-281, 358
 * Reference data flow per iteration: cuRAND fills a complex spectrum [K*C][T+1] in HBM, a kernel scales it by f^(-beta/2),
 * cuFFT writes [K*C][2T] time samples, rearrangeNoise keeps the first T of them (minus the decayed sample at
 * `optimization_stride`, divided by sigma*2T) transposed to [K][T][C], setGaussianControls rewrites that tensor — about
 * eight passes over >= 2V-sized buffers (1.8 GB at K = 65536, T = 200, C = 2).
 *
 * Here: every step between the Gaussian spectrum z and the time-domain noise is LINEAR, so for control c
 *     eps[k][t][c] = sum_kk  G_c[t][kk] * z[k][c][kk],        kk = 2 f + (0: real, 1: imaginary part),  f = 0..T
 * with one table G_c[T][2T+2] that folds the f^(-beta/2) weights, the Hermitian inverse DFT of length N = 2T, the
 * offset subtraction and the 1/(sigma_c N) normalisation.  That is a dense [rollouts x 2T+2] x [2T+2 x T] contraction —
 * genuine matrix-core work — and it runs INSIDE the rollout kernel's prologue: each wave takes 16 rollouts x one control,
 * draws z with Philox in registers (B fragments), streams the pre-swizzled table from L2 (A fragments, one coalesced
 * 256-byte read per MFMA) and writes the result straight into the block's LDS sample rows.  Nothing but the table
 * (T*(2T+2)*C floats, 0.64 MB at config 5) is read from memory and nothing is written to HBM.
 * readControlSample() then applies the setGaussianControls rule exactly like the Gaussian sampler
 * (colored_noise.cu:378-386 ends with the same kernel).
 *
 * Numerics: v_mfma_f32_16x16x4_f32 chains = one kk-ascending fp32 fma chain per sample, restated by the CPU oracle
 * bit for bit; the table itself is built on the host in double precision from the reference's definition
 * (buildColoredNoiseBasis below) and rounded once.  Against a float64 evaluation of the reference's pipeline the
 * samples agree to ~1e-6 (tests/test_colored_noise.py) — the reference's own cuFFT path is not pinned any tighter.
 *
 * Spectrum layout for injected noise == the reference's samples_in_freq_complex_d_: z[K][C][T+1][2] (re, im).
 */
#ifndef MPPI_AMD_COLORED_NOISE_DISTRIBUTION_HPP_
#define MPPI_AMD_COLORED_NOISE_DISTRIBUTION_HPP_

#include <cmath>
#include <vector>
#include "mppi_amd/sampling_distributions/gaussian.hpp"

namespace mppi
{
namespace sampling_distributions
{
typedef float colored_f32x4 __attribute__((ext_vector_type(4)));

/** floats of spectrum noise per (rollout, control): real and imaginary part of T + 1 frequencies */
__host__ __device__ inline int coloredSpectrumFloats(int num_timesteps)
{
  return 2 * (num_timesteps + 1);
}
/** k-steps (4 spectrum entries each) and 16-row time blocks of the basis table */
__host__ __device__ inline int coloredNumKSteps(int num_timesteps)
{
  // rounded up to whole groups of 4 k-steps (= one Philox quad per lane); the padding rows of the table are zero
  return (((coloredSpectrumFloats(num_timesteps) + 3) / 4) + 3) & ~3;
}
__host__ __device__ inline int coloredNumTBlocks(int num_timesteps)
{
  return (num_timesteps + 15) / 16;
}

/**
 * Host: the frequency weights and normalisation of the reference (colored_noise.cu:294-338), in float like there:
 *   freq[i] = i / N (fftfreq, colored_noise.cuh:27-37), cutoff = max(fmin, 1/N); every frequency below the cutoff takes
 *   the weight of the first one at or above it; weight = powf(freq, -beta/2);
 *   sigma = 2 sqrt(sum_{f=1}^{F-2} w^2 + (w_{F-1} (1 + N%2) / 2)^2) / N.
 * coeff: [C][F], sigma: [C]
 */
inline void coloredNoiseWeights(int T, int C, const float* exponents, float fmin, std::vector<float>& coeff,
                                std::vector<float>& sigma)
{
  const int N = 2 * T, F = N / 2 + 1;
  std::vector<float> freq(F);
  for (int i = 0; i < F; i++)
    freq[i] = i / (1.0f * N);
  const float cutoff = fmaxf(fmin, 1.0f / N);
  int first_ok = 0;
  while (first_ok < F && freq[first_ok] < cutoff)
    first_ok++;
  coeff.assign((size_t)C * F, 0.0f);
  sigma.assign(C, 0.0f);
  for (int c = 0; c < C; c++)
  {
    for (int i = 0; i < F; i++)
    {
      // below the cutoff: the weight of the first frequency that is not (if there is none the reference leaves
      // powf(freq, .) of the raw frequency, including powf(0, -beta/2) = inf at f = 0; fmin is never that large in use)
      const float fe = (i < first_ok && first_ok < F) ? freq[first_ok] : freq[i];
      coeff[(size_t)c * F + i] = powf(fe, -exponents[c] / 2.0f);
    }
    float s = 0.0f;
    for (int j = 1; j < F - 1; j++)
      s += coeff[(size_t)c * F + j] * coeff[(size_t)c * F + j];
    const float last = coeff[(size_t)c * F + F - 1] * ((1.0f + (N % 2)) / 2.0f);
    s += last * last;
    sigma[c] = 2.0f * sqrtf(s) / N;
  }
}

/**
 * Host: the basis table in MFMA A-fragment order, frag[((c*KS + ks)*NTB + tb)*64 + lane] = G_c[t = 16 tb + (lane & 15)]
 * [kk = 4 ks + (lane >> 4)] (zero outside the table), with
 *   G_c[t][2f]   =  w_c[f] m_f (cos(2 pi f t / N) - d_t cos(2 pi f s / N)) / (sigma_c * 2 * T)
 *   G_c[t][2f+1] = -w_c[f] m_f (sin(2 pi f t / N) - d_t sin(2 pi f s / N)) / (sigma_c * 2 * T)   for 0 < f < T, else 0
 * m_0 = m_T = 1, m_f = 2 otherwise (Hermitian inverse real DFT of length N = 2T, unnormalised like cuFFT C2R);
 * s = optimization_stride (rearrangeNoise's offset_t), d_t = offset_decay_rate == 0 ? 0 : powf(offset_decay_rate, t)
 * (colored_noise.cu:39-56).  The imaginary parts at f = 0 and at the Nyquist frequency do not enter a real inverse DFT.
 */
inline void buildColoredNoiseBasis(int T, int C, const float* exponents, float offset_decay_rate, float fmin,
                                   int optimization_stride, std::vector<float>& frag)
{
  const int N = 2 * T, F = T + 1;
  const int KS = coloredNumKSteps(T), NTB = coloredNumTBlocks(T);
  std::vector<float> coeff, sigma;
  coloredNoiseWeights(T, C, exponents, fmin, coeff, sigma);
  frag.assign((size_t)C * KS * NTB * 64, 0.0f);
  const int s = optimization_stride;
  const double two_pi_over_n = 6.283185307179586476925286766559 / (double)N;
  for (int c = 0; c < C; c++)
  {
    const float denom = sigma[c] * 2 * T;
    for (int t = 0; t < T; t++)
    {
      const float d_t = offset_decay_rate == 0.0f ? 0.0f : powf(offset_decay_rate, (float)t);
      for (int f = 0; f < F; f++)
      {
        const double m_f = (f == 0 || f == T) ? 1.0 : 2.0;
        const double w = (double)coeff[(size_t)c * F + f] * m_f / (double)denom;
        const double at = two_pi_over_n * (double)(((long long)f * t) % N);
        const double as = two_pi_over_n * (double)(((long long)f * s) % N);
        const double g_re = w * (cos(at) - (double)d_t * cos(as));
        const double g_im = (f == 0 || f == T) ? 0.0 : -w * (sin(at) - (double)d_t * sin(as));
        for (int part = 0; part < 2; part++)
        {
          const int kk = 2 * f + part;
          const int ks = kk >> 2, g = kk & 3, tb = t >> 4, m = t & 15;
          frag[(((size_t)c * KS + ks) * NTB + tb) * 64 + 16 * g + m] = (float)(part == 0 ? g_re : g_im);
        }
      }
    }
  }
}

template <class DYN_PARAMS_T>
class ColoredNoiseDistribution : public GaussianDistribution<DYN_PARAMS_T>
{
public:
  using PARENT = GaussianDistribution<DYN_PARAMS_T>;
  static const int CONTROL_DIM = PARENT::CONTROL_DIM;
  static constexpr bool IN_LOOP_DRAW = false;  ///< the rows are filled by the prologue GEMM
  static constexpr bool COLORED = true;
  static constexpr int MAX_TB = 16;            ///< time blocks (of 16 steps) accumulated per pass over the spectrum

  /* reference: ColoredNoiseParamsImpl, colored_noise/colored_noise.cuh:45-73 */
  float exponents_[CONTROL_DIM] = { 0.0f };
  float offset_decay_rate_ = 0.97f;
  float fmin_ = 0.0f;
  /** basis table in A-fragment order (device pointer, owned by the engine; rebuilt when T / stride / params change) */
  const float* basis_d_ = nullptr;

  ColoredNoiseDistribution(hipStream_t stream = 0) : PARENT(stream)
  {
  }

  /** Philox spectrum draw: entry kk of (rollout, control c) is element (kk >> 2) & 3 of quad ((kk >> 4) << 2) + (kk & 3) of
   *  stream 1 + c — each lane's quad feeds four consecutive k-steps of its own k-group, so no draw is wasted */
  __device__ __forceinline__ void initializeDistributions(const float* __restrict__ output, const float t_0,
                                                          const float dt, float* __restrict__ theta_d)
  {
    const int T = this->params_.num_timesteps;
    const int stride = PARENT::rowStride(T);
    const int bx = this->rolloutsPerBlock();
    const int tid_flat = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
    const int nwaves = (int)(blockDim.x * blockDim.y * blockDim.z) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid_flat >> 6);
    const int lane = tid_flat & 63;
    const int n = lane & 15, g = lane >> 4;
    const int row0 = (int)(blockIdx.x * bx);
    const int nrows = min(bx, this->params_.num_rollouts - row0);
    const int nz = (int)blockDim.z;
    const int RG = (bx + 15) >> 4;
    const int KS = coloredNumKSteps(T), NTB = coloredNumTBlocks(T), KK = coloredSpectrumFloats(T);
    const bool from_buffer = this->noise_source_ == NOISE_EPS_BUFFER;
    for (int unit = wave; unit < RG * CONTROL_DIM; unit += nwaves)
    {
      const int c = unit % CONTROL_DIM;
      const int rg = unit / CONTROL_DIM;
      const int row = 16 * rg + n;
      const bool valid = row < nrows;
      const uint32_t rollout = (uint32_t)(row0 + row + this->rollout_offset_);
      const float* __restrict__ zbuf =
          from_buffer ? this->eps_d_ + ((size_t)(row0 + (valid ? row : 0)) * CONTROL_DIM + c) * KK : nullptr;
      const float* __restrict__ basis = basis_d_ + (size_t)c * KS * NTB * 64 + lane;
      for (int tb0 = 0; tb0 < NTB; tb0 += MAX_TB)
      {
        colored_f32x4 acc[MAX_TB];
#pragma unroll
        for (int tb = 0; tb < MAX_TB; tb++)
          acc[tb] = colored_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
        /* A fragments of one group (4 k-steps x up to MAX_TB time blocks) are fetched one group AHEAD of their use:
         * the L2 round trip of group j + 1 overlaps the Philox draw and the 4 * NTB MFMAs of group j */
        float a_cur[4][MAX_TB], a_nxt[4][MAX_TB];
        auto fetch = [&](float (&a)[4][MAX_TB], const int ks4) {
#pragma unroll
          for (int e = 0; e < 4; e++)
          {
            const float* __restrict__ a_row = basis + (size_t)(ks4 + e) * NTB * 64;
#pragma unroll
            for (int tb = 0; tb < MAX_TB; tb++)
              a[e][tb] = (tb0 + tb < NTB) ? a_row[(tb0 + tb) * 64] : 0.0f;
          }
        };
        fetch(a_cur, 0);
        for (int ks4 = 0; ks4 < KS; ks4 += 4)
        {
          if (ks4 + 4 < KS)
            fetch(a_nxt, ks4 + 4);
          float zq[4];
          if (from_buffer)
          {
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
              const int kk = 4 * (ks4 + e) + g;
              zq[e] = (kk < KK) ? zbuf[kk] : 0.0f;
            }
          }
          else
          {
            mppi::rng::normal4(this->seed_, this->generation_, (uint32_t)(1 + c), rollout, (uint32_t)(ks4 + g), zq);
          }
#pragma unroll
          for (int e = 0; e < 4; e++)
#pragma unroll
            for (int tb = 0; tb < MAX_TB; tb++)
              if (tb0 + tb < NTB)
                acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[e][tb], zq[e], acc[tb], 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; e++)
#pragma unroll
            for (int tb = 0; tb < MAX_TB; tb++)
              a_cur[e][tb] = a_nxt[e][tb];
        }
        if (valid)
        {
#pragma unroll
          for (int tb = 0; tb < MAX_TB; tb++)
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
              const int t = 16 * (tb0 + tb) + 4 * g + i;
              if (t < T)
                for (int z = 0; z < nz; z++)
                  theta_d[(z * bx + row) * stride + t * CONTROL_DIM + c] = acc[tb][i];
            }
        }
      }
    }
  }
};

}  // namespace sampling_distributions
}  // namespace mppi

#endif
