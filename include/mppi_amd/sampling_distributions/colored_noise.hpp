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

/* ====================================================================================================================
 * Radix-4 form (round 5), used whenever T is a multiple of 4 — every BASELINE horizon is.
 *
 * The dense table above spends 2 (2T+2) T multiply-adds per (rollout, control): an O(T^2) stand-in for the reference's FFT.
 * Two decimation steps of that FFT are applied in front of the GEMM: with T = 4P, N = 8P, w = exp(2 pi i / N) and the
 * weighted spectrum c_f = W_f (zr_f + i zi_f),
 *     x[4u + r] = Re sum_{f'=0}^{P}  D_r[f'] Omega^{f' u},     Omega = w^4 = exp(2 pi i / 2P),
 *     D_r[f'] = w^{f' r} ( c_f' + (-i)^r conj(c_{2P-f'}) + i^r c_{2P+f'} + (-1)^r conj(c_{4P-f'}) ),
 * i.e. the four time-residue classes r = t mod 4 are four inverse real DFTs of length 2P over the SAME basis: one table
 * A'[u][f'] = (cos, -sin)(pi f' u / P) for every class AND every control (the frequency weights now travel with the spectrum
 * operand), P x 2(P+1) entries instead of T x 2(T+1) per control — 26 KB instead of 2 x 321 KB at T = 200 — and a quarter of
 * the multiply-adds.  Per (f', rollout, control) a lane draws the eight spectrum entries of the four partner frequencies
 * (two Philox quads), forms the radix-4 butterfly and the three twiddle products in registers (~40 flops) and holds the
 * B operands of all four classes for two k-steps; the classes share every A fragment it fetches.  The offset term
 * -decay^t x[s] of rearrangeNoise is linear in ONE time sample and is applied to the finished rows (coloredOffsetPass).
 *
 * Table blob (one allocation, basis_d_): [A' fragments KS4 x NTB4 x 64][slot weights C x P1 x 8][twiddles P1 x 8][decay T],
 * P1 = P + 1 padded to a multiple of 8.
 *   A' fragment order: frag[(ks * NTB4 + tb) * 64 + lane] = A'[u = 16 tb + (lane & 15)][f' = 4 (ks >> 1) + (lane >> 4)],
 *                      part ks & 1 (0: cos, 1: -sin): lane group g owns frequency 4 q + g of k-step pair q — real and
 *                      imaginary part of one D_r[f'] land in the SAME lane, so no spectrum entry is drawn twice
 *   slot weights:      w[f'][2 slot + part] for the slots A = f', B = 2P - f', C = 2P + f', D = 4P - f': m_f w_c[f] / (sigma_c N)
 *                      with the conjugations of B and D folded into the sign of their imaginary weight, zero for the imaginary
 *                      parts at f = 0 and f = T, and zero for a slot that repeats a frequency another slot of the same f'
 *                      already carries (f' = 0: C = B = 2P;  f' = P: B = A = P and D = C = 3P)
 *   twiddles:          (cos, sin)(2 pi f' r / N) for r = 1, 2, 3
 * Philox spectrum of this form: quad 2 f' of stream 1 + c = (zr, zi) of slots A, B; quad 2 f' + 1 = slots C, D.
 * ==================================================================================================================== */
__host__ __device__ inline bool coloredRadix4(int num_timesteps)
{
  return (num_timesteps & 3) == 0 && num_timesteps >= 16;
}
/** the radix-4 form reads x[s] from the finished rows: the offset sample must be one of the T kept ones */
__host__ __device__ inline bool coloredUseRadix4(int num_timesteps, int optimization_stride)
{
  return coloredRadix4(num_timesteps) && optimization_stride >= 0 && optimization_stride < num_timesteps;
}
__host__ __device__ inline int coloredR4P1(int num_timesteps)
{
  return ((num_timesteps >> 2) + 1 + 7) & ~7;
}
/** k-steps: two per frequency quadruple 4 q + g, q = 0 .. ceil((P + 1) / 4) - 1, rounded up to whole tiles of 4 */
__host__ __device__ inline int coloredR4KSteps(int num_timesteps)
{
  const int pairs = ((num_timesteps >> 2) + 1 + 3) >> 2;
  return (2 * pairs + 3) & ~3;
}
__host__ __device__ inline int coloredR4TBlocks(int num_timesteps)
{
  return ((num_timesteps >> 2) + 15) >> 4;
}
__host__ __device__ inline size_t coloredR4FragFloats(int T)
{
  return (size_t)coloredR4KSteps(T) * coloredR4TBlocks(T) * 64;
}
__host__ __device__ inline size_t coloredR4BlobFloats(int T, int C)
{
  return coloredR4FragFloats(T) + (size_t)C * coloredR4P1(T) * 8 + (size_t)coloredR4P1(T) * 8 + (size_t)((T + 3) & ~3);
}

/** Host: the radix-4 table blob (layout above); every entry is evaluated in double and rounded to float once */
inline void buildColoredNoiseBasisRadix4(int T, int C, const float* exponents, float offset_decay_rate, float fmin,
                                         std::vector<float>& blob)
{
  const int N = 2 * T, F = T + 1, P = T / 4, P1 = coloredR4P1(T);
  const int KS = coloredR4KSteps(T), NTB = coloredR4TBlocks(T);
  std::vector<float> coeff, sigma;
  coloredNoiseWeights(T, C, exponents, fmin, coeff, sigma);
  blob.assign(coloredR4BlobFloats(T, C), 0.0f);
  const double pi = 3.14159265358979323846264338327950288;
  float* frag = blob.data();
  for (int ks = 0; ks < KS; ks++)
    for (int g = 0; g < 4; g++)
    {
      const int fp = 4 * (ks >> 1) + g;
      if (fp > P)
        continue;
      for (int u = 0; u < P; u++)
      {
        const double a = pi * (double)(((long long)fp * u) % (2 * P)) / (double)P;
        frag[((size_t)ks * NTB + (u >> 4)) * 64 + 16 * g + (u & 15)] = (float)((ks & 1) ? -sin(a) : cos(a));
      }
    }
  float* wslot = frag + coloredR4FragFloats(T);
  for (int c = 0; c < C; c++)
  {
    const double denom = (double)(sigma[c] * 2 * T);
    for (int fp = 0; fp <= P; fp++)
    {
      const int f_of_slot[4] = { fp, 2 * P - fp, 2 * P + fp, 4 * P - fp };
      for (int slot = 0; slot < 4; slot++)
      {
        const int f = f_of_slot[slot];
        bool repeated = false;  // a frequency an earlier slot of this f' carries already
        for (int e = 0; e < slot; e++)
          repeated = repeated || f_of_slot[e] == f;
        const double m_f = (f == 0 || f == T) ? 1.0 : 2.0;
        const double w = repeated ? 0.0 : (double)coeff[(size_t)c * F + f] * m_f / denom;
        const double conj_sign = (slot == 1 || slot == 3) ? -1.0 : 1.0;
        float* dst = wslot + ((size_t)c * P1 + fp) * 8 + 2 * slot;
        dst[0] = (float)w;
        dst[1] = (f == 0 || f == T) ? 0.0f : (float)(conj_sign * w);
      }
    }
  }
  float* twid = wslot + (size_t)C * P1 * 8;
  for (int fp = 0; fp <= P; fp++)
    for (int r = 1; r < 4; r++)
    {
      const double a = 2.0 * pi * (double)(((long long)fp * r) % N) / (double)N;
      twid[(size_t)fp * 8 + 2 * (r - 1)] = (float)cos(a);
      twid[(size_t)fp * 8 + 2 * (r - 1) + 1] = (float)sin(a);
    }
  float* decay = twid + (size_t)P1 * 8;
  for (int t = 0; t < T; t++)
    decay[t] = offset_decay_rate == 0.0f ? 0.0f : powf(offset_decay_rate, (float)t);
}

/**
 * The butterfly of one frequency quadruple: n[8] = (re, im) of the spectrum at the slots A, B, C, D, w[8] their weights,
 * tw[6] = (cos, sin) of the twiddles r = 1, 2, 3  ->  d[r][2] = (Re, Im) D_r.  Every operation is a single rounded fp32
 * add / multiply or an explicit fma; the CPU oracle performs the same sequence (oracle_colored.hpp: coloredButterfly).
 */
__host__ __device__ inline void coloredButterflyRadix4(const float (&n)[8], const float* __restrict__ w,
                                                       const float* __restrict__ tw, float (&d)[4][2])
{
  float p[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
    p[i] = w[i] * n[i];
  // A = p0 + i p1, B = p2 + i p3, C = p4 + i p5, D = p6 + i p7
  const float s0r = p[0] + p[6], s0i = p[1] + p[7];  // A + D
  const float s1r = p[0] - p[6], s1i = p[1] - p[7];  // A - D
  const float s2r = p[2] + p[4], s2i = p[3] + p[5];  // B + C
  const float s3r = p[3] - p[5], s3i = p[4] - p[2];  // i (C - B)
  const float c0r = s0r + s2r, c0i = s0i + s2i;
  const float c2r = s0r - s2r, c2i = s0i - s2i;
  const float c1r = s1r + s3r, c1i = s1i + s3i;
  const float c3r = s1r - s3r, c3i = s1i - s3i;
  d[0][0] = c0r;
  d[0][1] = c0i;
  const float cr[3] = { c1r, c2r, c3r }, ci[3] = { c1i, c2i, c3i };
#pragma unroll
  for (int r = 0; r < 3; r++)
  {
    const float tr = tw[2 * r], ti = tw[2 * r + 1];
    d[r + 1][0] = mppi::det::fma(cr[r], tr, -(ci[r] * ti));
    d[r + 1][1] = mppi::det::fma(cr[r], ti, ci[r] * tr);
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
    const int dense = 2 * 4 * (ntb < MAX_TB ? ntb : MAX_TB) * 64 * (int)sizeof(float);
    const int radix4 = 2 * 4 * MAX_TB4 * 64 * (int)sizeof(float);
    return dense > radix4 ? dense : radix4;
  }
  static constexpr int MAX_TB4 = 4;  ///< radix-4 form: time blocks (of 16 samples of ONE residue class) per pass

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
   * Radix-4 form of coloredTiles (see the block comment above buildColoredNoiseBasisRadix4): one (control, row group, chunk of
   * NTBP time blocks) with the accumulators of all four residue classes in registers — 4 x NTBP MFMA tiles share every A
   * fragment.  Tile j = k-steps 4j .. 4j+3 = the frequencies f' = 8j + g and 8j + 4 + g of lane group g, real and imaginary part.
   */
  template <int NTBP>
  __device__ __forceinline__ void coloredTilesRadix4(const float* __restrict__ basis, const float* __restrict__ wslot,
                                                     const float* __restrict__ twid, float* __restrict__ stage,
                                                     float* __restrict__ theta_d, const float* __restrict__ zbuf, const int c,
                                                     const int tb0, const int NTB, const int NG, const int T, const int stride,
                                                     const int bx, const int nz, const int row, const uint32_t rollout,
                                                     const bool active, const bool valid, const bool from_buffer,
                                                     const int tid_flat, const int nthreads, const int lane)
  {
    const int g = lane >> 4;
    const int P = T >> 2;
    constexpr int TILE4 = NTBP * 16;
    colored_f32x4 acc[4][NTBP];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int tb = 0; tb < NTBP; tb++)
        acc[r][tb] = colored_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
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
    // the B operands of tile j: zq[class][e], e = 2 h + part for the lane's two frequencies f' = 8 j + 4 h + g
    auto draw = [&](const int j, float (&zq)[4][4]) {
#pragma unroll
      for (int h = 0; h < 2; h++)
      {
        const int fp = 8 * j + 4 * h + g;
        float d[4][2];
        if (fp <= P)
        {
          float n[8];
          if (from_buffer)
          {
            const int fs[4] = { fp, 2 * P - fp, 2 * P + fp, 4 * P - fp };
#pragma unroll
            for (int slot = 0; slot < 4; slot++)
            {
              n[2 * slot] = zbuf[2 * fs[slot]];
              n[2 * slot + 1] = zbuf[2 * fs[slot] + 1];
            }
          }
          else
          {
            float q0[4], q1[4];
            mppi::rng::normal4(this->seed_, this->generation_, (uint32_t)(1 + c), rollout, (uint32_t)(2 * fp), q0);
            mppi::rng::normal4(this->seed_, this->generation_, (uint32_t)(1 + c), rollout, (uint32_t)(2 * fp + 1), q1);
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
              n[i] = q0[i];
              n[4 + i] = q1[i];
            }
          }
          coloredButterflyRadix4(n, wslot + (size_t)fp * 8, twid + (size_t)fp * 8, d);
        }
        else
        {
#pragma unroll
          for (int r = 0; r < 4; r++)
            d[r][0] = d[r][1] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
          zq[r][2 * h] = d[r][0];
          zq[r][2 * h + 1] = d[r][1];
        }
      }
    };
    __syncthreads();  // the staging buffers may still be read by a slower wave of the previous pass
    stage_tile(0);
    float zq[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int e = 0; e < 4; e++)
        zq[r][e] = 0.0f;
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
            asm volatile("" : "+v"(acc[0][0]) : : "memory");
          }
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int tb = 0; tb < NTBP; tb++)
              acc[r][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[tb], zq[r][e], acc[r][tb], 0, 0, 0);
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
        if (j + 1 < NG)
          draw(j + 1, zq);
      }
      __syncthreads();
    }
    if (valid)
    {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int tb = 0; tb < NTBP; tb++)
#pragma unroll
          for (int i = 0; i < 4; i++)
          {
            const int u = 16 * (tb0 + tb) + 4 * g + i;
            if (u < P)
              for (int z = 0; z < nz; z++)
                theta_d[(z * bx + row) * stride + (4 * u + r) * CONTROL_DIM + c] = acc[r][tb][i];
          }
    }
  }

  /** rearrangeNoise's offset (colored_noise.cu:39-56) on the finished rows: eps[t] = x[t] - decay^t x[s], one thread per
   *  (row, control); x[s] is read before the thread rewrites it */
  __device__ __forceinline__ void coloredOffsetPass(float* __restrict__ theta_d, const float* __restrict__ decay, const int T,
                                                    const int stride, const int rows, const int tid_flat, const int nthreads)
  {
    const int s = this->optimization_stride_;
    for (int job = tid_flat; job < rows * CONTROL_DIM; job += nthreads)
    {
      const int row = job / CONTROL_DIM, c = job - row * CONTROL_DIM;
      float* r = theta_d + (size_t)row * stride + c;
      const float xs = r[s * CONTROL_DIM];
      for (int t = 0; t < T; t++)
        r[t * CONTROL_DIM] = mppi::det::fma(-decay[t], xs, r[t * CONTROL_DIM]);
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
    const int stride = this->rowStrideNow();
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
    if (coloredUseRadix4(T, this->optimization_stride_))
    {
      const int KS4 = coloredR4KSteps(T), NTB4 = coloredR4TBlocks(T), P1 = coloredR4P1(T);
      const float* __restrict__ wslot_all = basis_d_ + coloredR4FragFloats(T);
      const float* __restrict__ twid = wslot_all + (size_t)CONTROL_DIM * P1 * 8;
      const float* __restrict__ decay = twid + (size_t)P1 * 8;
      for (int c = 0; c < CONTROL_DIM; c++)
        for (int rg0 = 0; rg0 < RG; rg0 += nwaves)
        {
          const int rg = rg0 + wave;
          const bool active = rg < RG;  // wave-uniform
          const int row = 16 * rg + n;
          const bool valid = active && row < nrows;
          const uint32_t rollout = (uint32_t)(row0 + row + this->rollout_offset_);
          const float* __restrict__ zbuf =
              from_buffer ? this->eps_d_ + ((size_t)(row0 + (valid ? row : 0)) * CONTROL_DIM + c) * KK : nullptr;
          for (int tb0 = 0; tb0 < NTB4; tb0 += MAX_TB4)
          {
#define MPPI_COLORED_R4_CASE(Q)                                                                                         \
  case Q:                                                                                                               \
    coloredTilesRadix4<Q>(basis_d_, wslot_all + (size_t)c * P1 * 8, twid, stage, theta_d, zbuf, c, tb0, NTB4, KS4 >> 2, T, \
                          stride, bx, nz, row, rollout, active, valid, from_buffer, tid_flat, nthreads, lane);          \
    break;
            switch (min(MAX_TB4, NTB4 - tb0))  // block-uniform
            {
              MPPI_COLORED_R4_CASE(1)
              MPPI_COLORED_R4_CASE(2)
              MPPI_COLORED_R4_CASE(3)
              default:
                coloredTilesRadix4<4>(basis_d_, wslot_all + (size_t)c * P1 * 8, twid, stage, theta_d, zbuf, c, tb0, NTB4,
                                      KS4 >> 2, T, stride, bx, nz, row, rollout, active, valid, from_buffer, tid_flat, nthreads,
                                      lane);
            }
#undef MPPI_COLORED_R4_CASE
          }
        }
      __syncthreads();  // every row is complete (rows in HBM: the barrier also waits for the block's stores)
      if (this->rows_global_d_)
        __threadfence_block();
      coloredOffsetPass(theta_d, decay, T, stride, nrows == bx ? bx * nz : 0, tid_flat, nthreads);
      if (nrows != bx)
      {  // a partial last block: only its valid rows of every system
        for (int z = 0; z < nz; z++)
          coloredOffsetPass(theta_d + (size_t)z * bx * stride, decay, T, stride, nrows, tid_flat, nthreads);
      }
      return;
    }
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
