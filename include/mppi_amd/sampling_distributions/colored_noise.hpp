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
 * genuine matrix-core work — and it runs INSIDE the rollout kernel's prologue: each wave takes 16 rollouts, draws z with
 * Philox in registers (B fragments), takes the pre-swizzled table (A fragments) from an LDS tile that the whole block
 * stages from L2 once, and writes the result straight into the block's LDS sample rows.  Nothing but the table
 * (T*(2T+2)*C floats, 0.64 MB at config 5, L2-resident) is read from memory and nothing is written to HBM.
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
  // rounded up to even: tiles are processed with an even compile-time width (coloredTiles<NTBP>); the padding block is zero
  return (((num_timesteps + 15) / 16) + 1) & ~1;
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
  /** long horizons: the prologue GEMM writes its tiles straight into the HBM rows (the accumulators are registers; only the
   *  table staging tiles need the LDS), the step loop reads them back through sampleRow() like the Gaussian sampler */
  static constexpr bool SUPPORTS_GLOBAL_ROWS = true;
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

  /** block-shared LDS: the double-buffered A-fragment staging tile (4 k-steps x min(MAX_TB, NTB) time blocks x 64 lanes).
   *  Layout of this class's LDS region: [slots x sample row][staging tile] (reference contract: Grd + Blk * slots bytes,
   *  utils/managed.cuh:104-111; the split is the class's own business) */
  __host__ __device__ inline int getGrdSharedSizeBytes() const
  {
    const int ntb = coloredNumTBlocks(this->params_.num_timesteps);  // even
    return 2 * 4 * (ntb < MAX_TB ? ntb : MAX_TB) * 64 * (int)sizeof(float);
  }

  /**
   * One (control, row-group pass, time-block chunk) of the prologue GEMM with a COMPILE-TIME number of time blocks NTBP
   * (even; the table's time-block count is padded to even with a zero block, which adds fma(0, z, acc) = acc):
   * the 4 * NTBP MFMAs of a tile are straight-line code, so their LDS reads are issued ahead of the matrix pipe.
   */
  template <int NTBP>
  __device__ __forceinline__ void coloredTiles(const float* __restrict__ basis, float* __restrict__ stage,
                                               float* __restrict__ theta_d, const float* __restrict__ zbuf,
                                               const int c, const int tb0, const int ntb, const int NTB, const int NG,
                                               const int KK, const int T, const int stride, const int bx, const int nz,
                                               const int row, const uint32_t rollout, const bool active,
                                               const bool valid, const bool from_buffer, const int tid_flat,
                                               const int nthreads, const int lane)
  {
    const int g = lane >> 4;
    constexpr int TILE4 = NTBP * 16;  // float4 per k-step of the staged tile
    colored_f32x4 acc[NTBP];
#pragma unroll
    for (int tb = 0; tb < NTBP; tb++)
      acc[tb] = colored_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    // tile j = table rows ks = 4j .. 4j+3, time blocks tb0 .. tb0+NTBP-1  ->  stage[j & 1][e][tb][lane], copied with the
    // gfx950 LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per wave instruction, no staging registers, no
    // ds_write pass); a tile is exactly NTBP KiB, the chunks are dealt round-robin to the waves of the block
    const int wave_u = __builtin_amdgcn_readfirstlane(tid_flat >> 6);
    const int nwaves = nthreads >> 6;
    auto stage_tile = [&](const int j) {
      float* dst_tile = stage + (size_t)(j & 1) * 4 * NTBP * 64;
      for (int chunk = wave_u; chunk < NTBP; chunk += nwaves)
      {
        const int i = chunk * 64 + lane;  // float4 index inside the tile
        const int e = i / TILE4, r = i - e * TILE4;
        const float* src = basis + ((size_t)(4 * j + e) * NTB + tb0) * 64 + 4 * r;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst_tile + chunk * 256), 16, 0, 0);
      }
    };
    // the spectrum entries of tile j: one Philox quad per lane (or four loads in replay mode)
    auto draw = [&](const int j, float (&zq)[4]) {
      if (from_buffer)
      {
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
          const int kk = 4 * (4 * j + e) + g;
          zq[e] = (kk < KK) ? zbuf[kk] : 0.0f;
        }
      }
      else
      {
        mppi::rng::normal4(this->seed_, this->generation_, (uint32_t)(1 + c), rollout, (uint32_t)(4 * j + g), zq);
      }
    };
    __syncthreads();  // the staging buffers may still be read by a slower wave of the previous pass
    stage_tile(0);
    float zq[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (active)
      draw(0, zq);
    __syncthreads();
    for (int j = 0; j < NG; j++)
    {
      if (j + 1 < NG)
        stage_tile(j + 1);
      if (active)
      {
        const float* __restrict__ a_tile = stage + (size_t)(j & 1) * 4 * NTBP * 64 + lane;
#if !defined(MPPI_COLORED_SIMPLE_READS)
        // The A fragments of k-step e + 1 are fetched from LDS while the matrix pipe works on k-step e.  Left to itself the
        // compiler places every ds_read directly in front of the two MFMAs that use it, followed by s_waitcnt lgkmcnt(0):
        // 2 * NTBP exposed LDS round trips per tile on a wave that is alone on its SIMD.  The empty asm statements pin the
        // order (values tied to VGPRs must have arrived; the "memory" clobber keeps the reads of the next row above it; the
        // tie on acc[0] keeps this row's MFMA chain below it): one exposed round trip per tile instead.
        float a_cur[NTBP], a_nxt[NTBP];
#pragma unroll
        for (int tb = 0; tb < NTBP; tb++)
          a_cur[tb] = a_tile[tb * 64];
#pragma unroll
        for (int tb = 0; tb < NTBP; tb++)
          asm volatile("" : "+v"(a_cur[tb]));
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
          if (e < 3)
          {
#pragma unroll
            for (int tb = 0; tb < NTBP; tb++)
              a_nxt[tb] = a_tile[((e + 1) * NTBP + tb) * 64];
            asm volatile("" : "+v"(acc[0]) : : "memory");
          }
#pragma unroll
          for (int tb = 0; tb < NTBP; tb++)
            acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[tb], zq[e], acc[tb], 0, 0, 0);
          if (e < 3)
          {
#pragma unroll
            for (int tb = 0; tb < NTBP; tb++)
            {
              asm volatile("" : "+v"(a_nxt[tb]));
              a_cur[tb] = a_nxt[tb];
            }
          }
        }
#else
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
          for (int tb = 0; tb < NTBP; tb++)
            acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_tile[(e * NTBP + tb) * 64], zq[e], acc[tb], 0, 0, 0);
#endif
        // the next tile's draw (~300 VALU instructions) does not depend on these MFMAs, so the scheduler is free to
        // interleave the two (measured: neutral with ROCm 7.2's hipcc, which keeps them apart)
        float zn[4];
        draw(min(j + 1, NG - 1), zn);
#pragma unroll
        for (int e = 0; e < 4; e++)
          zq[e] = zn[e];
      }
      __syncthreads();
    }
    if (valid)
    {
#pragma unroll
      for (int tb = 0; tb < NTBP; tb++)
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

  /**
   * Fills the block's sample rows with colored noise.  All waves of the block walk the table in lockstep — control by
   * control, 4 k-steps (one Philox quad per lane) at a time — so that every A fragment is fetched from L2 ONCE per
   * block: the threads stage the next tile global -> LDS (float4, coalesced) while the matrix cores work on the
   * current one, one barrier per tile.  Wave w owns the 16 rollouts of row group w (+ nwaves, ...) and keeps their
   * accumulators in registers; waves without a row group (the sampler / cost waves of the pipelined kernels) only help
   * with the staging.
   * Philox spectrum draw: entry kk of (rollout, control c) is element (kk >> 2) & 3 of quad ((kk >> 4) << 2) + (kk & 3)
   * of stream 1 + c — each lane's quad feeds four consecutive k-steps of its own k-group, so no draw is wasted.
   */
  __device__ __forceinline__ void initializeDistributions(const float* __restrict__ output, const float t_0,
                                                          const float dt, float* __restrict__ theta_d)
  {
    const int T = this->params_.num_timesteps;
    const int stride = PARENT::rowStride(T);
    const int bx = this->rolloutsPerBlock();
    const int nthreads = (int)(blockDim.x * blockDim.y * blockDim.z);
    const int tid_flat = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
    const int nwaves = nthreads >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid_flat >> 6);
    const int lane = tid_flat & 63;
    const int n = lane & 15;
    const int row0 = (int)(blockIdx.x * bx);
    const int nrows = min(bx, this->params_.num_rollouts - row0);
    const int nz = this->systemsPerBlock();
    const int RG = (bx + 15) >> 4;
    const int KS = coloredNumKSteps(T), NTB = coloredNumTBlocks(T), KK = coloredSpectrumFloats(T);
    const int NG = KS >> 2;  // tiles of 4 k-steps
    const bool from_buffer = this->noise_source_ == NOISE_EPS_BUFFER;
    // [slot rows][staging tiles] in LDS; with the rows in HBM (theta_d then points there) the staging tiles start the region
    float* __restrict__ stage = this->rows_global_d_ ? this->staging_lds_
                                                     : theta_d + (((size_t)bx * nz * stride + 3) & ~(size_t)3);  // 16-byte aligned
    for (int c = 0; c < CONTROL_DIM; c++)
    {
      const float* __restrict__ basis = basis_d_ + (size_t)c * KS * NTB * 64;
      for (int rg0 = 0; rg0 < RG; rg0 += nwaves)
      {
        const int rg = rg0 + wave;
        const bool active = rg < RG;  // wave-uniform
        const int row = 16 * rg + n;
        const bool valid = active && row < nrows;
        const uint32_t rollout = (uint32_t)(row0 + row + this->rollout_offset_);
        const float* __restrict__ zbuf =
            from_buffer ? this->eps_d_ + ((size_t)(row0 + (valid ? row : 0)) * CONTROL_DIM + c) * KK : nullptr;
        for (int tb0 = 0; tb0 < NTB; tb0 += MAX_TB)
        {
          const int ntb = min(MAX_TB, NTB - tb0);  // even: NTB and MAX_TB are
#define MPPI_COLORED_CASE(P)                                                                                           \
  case P / 2:                                                                                                          \
    coloredTiles<P>(basis, stage, theta_d, zbuf, c, tb0, ntb, NTB, NG, KK, T, stride, bx, nz, row, rollout, active,    \
                    valid, from_buffer, tid_flat, nthreads, lane);                                                     \
    break;
          switch ((ntb + 1) >> 1)  // block-uniform
          {
            MPPI_COLORED_CASE(2)
            MPPI_COLORED_CASE(4)
            MPPI_COLORED_CASE(6)
            MPPI_COLORED_CASE(8)
            MPPI_COLORED_CASE(10)
            MPPI_COLORED_CASE(12)
            MPPI_COLORED_CASE(14)
            default:
              coloredTiles<16>(basis, stage, theta_d, zbuf, c, tb0, ntb, NTB, NG, KK, T, stride, bx, nz, row, rollout,
                               active, valid, from_buffer, tid_flat, nthreads, lane);
          }
#undef MPPI_COLORED_CASE
        }
      }
    }
  }
};

}  // namespace sampling_distributions
}  // namespace mppi

#endif
