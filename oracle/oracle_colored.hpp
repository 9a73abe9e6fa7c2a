/**
 * oracle_colored.hpp — CPU restatement of the colored-noise sampler.  TEST INFRASTRUCTURE ONLY (see oracle_core.hpp).
 *
 * Follows (paths relative to the reference's include/mppi/sampling_distributions/colored_noise/):
 *   fftfreq                                   colored_noise.cuh:27-37
 *   frequency weights, cutoff, sigma          colored_noise.cu:294-338   (the same code as powerlaw_psd_gaussian :69-113)
 *   configureFrequencyNoise                   colored_noise.cu:12-37     (scale re/im by the weight, im = 0 at f = 0 and
 *                                                                         at the last frequency when their count is odd)
 *   cufftExecC2R, length 2T, unnormalised     colored_noise.cu:358       (third-party cuFFT: restated from the definition
 *                                                                         of the inverse real DFT — parity unpinned
 *                                                                         beyond the definition)
 *   rearrangeNoise                            colored_noise.cu:39-56     ((x[t] - x[offset] decay^t) / (sigma 2T), first T)
 * and the python reference the authors cite, scripts/colored_noise.py:10-106 (same algorithm in numpy).
 *
 * Three flavours of the same map z (Gaussian spectrum, layout [K][C][T+1][2]) -> eps [K][T][C] (coloredNoiseRadix4: below):
 *   coloredNoiseDefinition  the pipeline above, step by step, accumulating in double: the DEFINITION
 *                           (checked against numpy.fft.irfft in tests/test_colored_noise.py)
 *   coloredNoiseGemm        eps = G z with the folded table G_c[T][2T+2] rounded to fp32 and one kk-ascending fp32 fma
 *                           chain per sample — the arithmetic the HIP engine's MFMA prologue performs; bit-exact
 *                           partner of the product, and within ~1e-6 of the definition.
 */
#ifndef MPPI_ORACLE_COLORED_HPP_
#define MPPI_ORACLE_COLORED_HPP_

#include <cmath>
#include <cstdint>
#include <vector>

#include "mppi_amd/det_math.h"
#include "oracle_rng.hpp"

namespace oracle
{
struct ColoredNoiseParams
{
  std::vector<float> exponents; /* [C] */
  float offset_decay_rate = 0.97f; /* colored_noise.cuh:49 */
  float fmin = 0.0f;
};

/** frequency weights [C][F] and sigma [C], float arithmetic as the reference's host code */
inline void coloredWeights(int T, int C, const ColoredNoiseParams& p, std::vector<float>& w, std::vector<float>& sigma)
{
  const int sample_num_timesteps = 2 * T;
  const int freq_size = sample_num_timesteps / 2 + 1;
  std::vector<float> sample_freq(freq_size);
  for (int i = 0; i < freq_size; i++)
    sample_freq[i] = i / (1.0f * sample_num_timesteps);
  const float cutoff_freq = fmaxf(p.fmin, 1.0f / sample_num_timesteps);
  w.assign((size_t)C * freq_size, 0.0f);
  /* the reference's loop: frequencies below the cutoff are overwritten with the first frequency at/above it */
  int smaller_index = 0;
  for (int i = 0; i < freq_size; i++)
  {
    if (sample_freq[i] < cutoff_freq)
    {
      smaller_index++;
    }
    else if (smaller_index < freq_size)
    {
      for (int j = 0; j < smaller_index; j++)
      {
        sample_freq[j] = sample_freq[smaller_index];
        for (int k = 0; k < C; k++)
          w[(size_t)k * freq_size + j] = powf(sample_freq[smaller_index], -p.exponents[k] / 2.0f);
      }
    }
    for (int j = 0; j < C; j++)
      w[(size_t)j * freq_size + i] = powf(sample_freq[i], -p.exponents[j] / 2.0f);
  }
  sigma.assign(C, 0.0f);
  for (int i = 0; i < C; i++)
  {
    for (int j = 1; j < freq_size - 1; j++)
      sigma[i] += w[(size_t)i * freq_size + j] * w[(size_t)i * freq_size + j];
    const float last = w[(size_t)i * freq_size + freq_size - 1] * ((1.0f + (sample_num_timesteps % 2)) / 2.0f);
    sigma[i] += last * last;
    sigma[i] = 2.0f * sqrtf(sigma[i]) / sample_num_timesteps;
  }
}

/** z: [K][C][T+1][2]  ->  eps: [K][T][C], the reference's pipeline with a double-precision inverse real DFT */
inline void coloredNoiseDefinition(int K, int T, int C, const ColoredNoiseParams& p, int offset_t, const float* z,
                                   float* eps)
{
  const int N = 2 * T, F = T + 1;
  std::vector<float> w, sigma;
  coloredWeights(T, C, p, w, sigma);
  const double two_pi_over_n = 6.283185307179586476925286766559 / (double)N;
  std::vector<double> re(F), im(F), x(N);
  for (int k = 0; k < K; k++)
    for (int c = 0; c < C; c++)
    {
      const float* zz = z + ((size_t)k * C + c) * F * 2;
      for (int f = 0; f < F; f++)
      { /* configureFrequencyNoise, in float like the kernel */
        re[f] = (double)(zz[2 * f] * w[(size_t)c * F + f]);
        if (f == 0 || (F % 2 == 1 && f == F - 1))
          im[f] = 0.0;
        else
          im[f] = (double)(zz[2 * f + 1] * w[(size_t)c * F + f]);
      }
      /* unnormalised C2R of length N: x[n] = X_0 + (-1)^n Re X_{N/2} + 2 sum_{0<f<N/2} Re(X_f e^{+2 pi i f n / N});
       * the imaginary part of the Nyquist bin does not enter a real inverse transform */
      for (int n = 0; n < N; n++)
      {
        double acc = re[0] + ((n & 1) ? -re[T] : re[T]);
        for (int f = 1; f < T; f++)
        {
          const double a = two_pi_over_n * (double)(((long long)f * n) % N);
          acc += 2.0 * (re[f] * cos(a) - im[f] * sin(a));
        }
        x[n] = acc;
      }
      for (int t = 0; t < T; t++)
      { /* rearrangeNoise */
        const float decayed_offset = p.offset_decay_rate == 0.0f ? 0.0f : powf(p.offset_decay_rate, (float)t);
        eps[((size_t)k * T + t) * C + c] =
            (float)((x[t] - x[offset_t] * (double)decayed_offset) / (double)(sigma[c] * 2 * T));
      }
    }
}

/** the folded table G[c][t][kk], kk = 2f + part, rounded to float once */
inline void coloredBasis(int T, int C, const ColoredNoiseParams& p, int offset_t, std::vector<float>& G)
{
  const int N = 2 * T, F = T + 1, KK = 2 * F;
  std::vector<float> w, sigma;
  coloredWeights(T, C, p, w, sigma);
  G.assign((size_t)C * T * KK, 0.0f);
  const double two_pi_over_n = 6.283185307179586476925286766559 / (double)N;
  for (int c = 0; c < C; c++)
  {
    const float denom = sigma[c] * 2 * T;
    for (int t = 0; t < T; t++)
    {
      const float d_t = p.offset_decay_rate == 0.0f ? 0.0f : powf(p.offset_decay_rate, (float)t);
      for (int f = 0; f < F; f++)
      {
        const double m_f = (f == 0 || f == T) ? 1.0 : 2.0;
        const double ww = (double)w[(size_t)c * F + f] * m_f / (double)denom;
        const double at = two_pi_over_n * (double)(((long long)f * t) % N);
        const double as = two_pi_over_n * (double)(((long long)f * offset_t) % N);
        G[((size_t)c * T + t) * KK + 2 * f] = (float)(ww * (cos(at) - (double)d_t * cos(as)));
        G[((size_t)c * T + t) * KK + 2 * f + 1] =
            (f == 0 || f == T) ? 0.0f : (float)(-ww * (sin(at) - (double)d_t * sin(as)));
      }
    }
  }
}

/** eps = G z as kk-ascending fp32 fma chains (the MFMA arithmetic of the engine) */
inline void coloredNoiseGemm(int K, int T, int C, const ColoredNoiseParams& p, int offset_t, const float* z, float* eps)
{
  const int F = T + 1, KK = 2 * F;
  std::vector<float> G;
  coloredBasis(T, C, p, offset_t, G);
  for (int k = 0; k < K; k++)
    for (int c = 0; c < C; c++)
    {
      const float* zz = z + ((size_t)k * C + c) * KK;
      for (int t = 0; t < T; t++)
      {
        const float* g = &G[((size_t)c * T + t) * KK];
        float acc = 0.0f;
        for (int kk = 0; kk < KK; kk++)
          acc = mppi::det::fma(g[kk], zz[kk], acc);
        eps[((size_t)k * T + t) * C + c] = acc;
      }
    }
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Radix-4 flavour (T a multiple of 4 and offset_t < T): the arithmetic of the engine's prologue since round 5
 * (include/mppi_amd/sampling_distributions/colored_noise.hpp).  Two decimation steps of the inverse FFT in front of a GEMM a
 * quarter the size: with T = 4P, N = 8P, w = exp(2 pi i / N), c_f = W_f (zr_f + i zi_f),
 *     x[4u + r] = Re sum_{f'=0}^{P} D_r[f'] exp(2 pi i f' u / 2P),
 *     D_r[f'] = w^{f' r} ( c_f' + (-i)^r conj(c_{2P-f'}) + i^r c_{2P+f'} + (-1)^r conj(c_{4P-f'}) ),
 * then eps[t] = x[t] - decay^t x[offset_t] (rearrangeNoise is linear in that one sample).  Restated here independently:
 * tables in double rounded once to float, the butterfly as single rounded fp32 operations in a fixed sequence, the GEMM as
 * one fp32 fma chain per sample in the engine's k order (k-step pair q, lane group g: f' = 4q + g; real part, then — in the
 * next k-step — imaginary part), the offset as one fma.  Within ~1e-6 of coloredNoiseDefinition like the dense flavour.
 * ------------------------------------------------------------------------------------------------------------------- */
inline bool coloredUseRadix4(int T, int offset_t)
{
  return (T & 3) == 0 && T >= 16 && offset_t >= 0 && offset_t < T;
}

struct ColoredRadix4Tables
{
  int P = 0, KS = 0;
  std::vector<float> basis;   /* [P][KS][4]: A'[u][ks][g] */
  std::vector<float> wslot;   /* [C][P+1][8] */
  std::vector<float> twid;    /* [P+1][6] */
  std::vector<float> decay;   /* [T] */
};

inline void coloredRadix4Tables(int T, int C, const ColoredNoiseParams& p, ColoredRadix4Tables& tb)
{
  const int N = 2 * T, F = T + 1, P = T / 4;
  const int pairs = (P + 1 + 3) / 4;
  const int KS = (2 * pairs + 3) & ~3;
  tb.P = P;
  tb.KS = KS;
  std::vector<float> w, sigma;
  coloredWeights(T, C, p, w, sigma);
  const double pi = 3.14159265358979323846264338327950288;
  tb.basis.assign((size_t)P * KS * 4, 0.0f);
  for (int u = 0; u < P; u++)
    for (int ks = 0; ks < KS; ks++)
      for (int g = 0; g < 4; g++)
      {
        const int fp = 4 * (ks >> 1) + g;
        if (fp > P)
          continue;
        const double a = pi * (double)(((long long)fp * u) % (2 * P)) / (double)P;
        tb.basis[((size_t)u * KS + ks) * 4 + g] = (float)((ks & 1) ? -sin(a) : cos(a));
      }
  tb.wslot.assign((size_t)C * (P + 1) * 8, 0.0f);
  for (int c = 0; c < C; c++)
  {
    const double denom = (double)(sigma[c] * 2 * T);
    for (int fp = 0; fp <= P; fp++)
    {
      const int fs[4] = { fp, 2 * P - fp, 2 * P + fp, 4 * P - fp };
      for (int slot = 0; slot < 4; slot++)
      {
        const int f = fs[slot];
        bool repeated = false; /* f' = 0: slots B and C are both 2P; f' = P: A = B = P and C = D = 3P — counted once */
        for (int e = 0; e < slot; e++)
          repeated = repeated || fs[e] == f;
        const double m_f = (f == 0 || f == T) ? 1.0 : 2.0;
        const double ww = repeated ? 0.0 : (double)w[(size_t)c * F + f] * m_f / denom;
        const double conj_sign = (slot == 1 || slot == 3) ? -1.0 : 1.0;
        float* dst = &tb.wslot[((size_t)c * (P + 1) + fp) * 8 + 2 * slot];
        dst[0] = (float)ww;
        dst[1] = (f == 0 || f == T) ? 0.0f : (float)(conj_sign * ww);
      }
    }
  }
  tb.twid.assign((size_t)(P + 1) * 6, 0.0f);
  for (int fp = 0; fp <= P; fp++)
    for (int r = 1; r < 4; r++)
    {
      const double a = 2.0 * pi * (double)(((long long)fp * r) % N) / (double)N;
      tb.twid[(size_t)fp * 6 + 2 * (r - 1)] = (float)cos(a);
      tb.twid[(size_t)fp * 6 + 2 * (r - 1) + 1] = (float)sin(a);
    }
  tb.decay.assign(T, 0.0f);
  for (int t = 0; t < T; t++)
    tb.decay[t] = p.offset_decay_rate == 0.0f ? 0.0f : powf(p.offset_decay_rate, (float)t);
}

/** the radix-4 butterfly + twiddles of one frequency quadruple: n[8] (re, im at slots A..D), w[8], tw[6] -> d[r][2] */
inline void coloredButterfly(const float* n, const float* w, const float* tw, float d[4][2])
{
  float q[8];
  for (int i = 0; i < 8; i++)
    q[i] = w[i] * n[i];
  const float s0r = q[0] + q[6], s0i = q[1] + q[7];
  const float s1r = q[0] - q[6], s1i = q[1] - q[7];
  const float s2r = q[2] + q[4], s2i = q[3] + q[5];
  const float s3r = q[3] - q[5], s3i = q[4] - q[2];
  const float cr[4] = { s0r + s2r, s1r + s3r, s0r - s2r, s1r - s3r };
  const float ci[4] = { s0i + s2i, s1i + s3i, s0i - s2i, s1i - s3i };
  d[0][0] = cr[0];
  d[0][1] = ci[0];
  for (int r = 1; r < 4; r++)
  {
    const float tr = tw[2 * (r - 1)], ti = tw[2 * (r - 1) + 1];
    d[r][0] = mppi::det::fma(cr[r], tr, -(ci[r] * ti));
    d[r][1] = mppi::det::fma(cr[r], ti, ci[r] * tr);
  }
}

inline void coloredNoiseRadix4(int K, int T, int C, const ColoredNoiseParams& p, int offset_t, const float* z, float* eps)
{
  ColoredRadix4Tables tb;
  coloredRadix4Tables(T, C, p, tb);
  const int P = tb.P, KS = tb.KS, KK = 2 * (T + 1);
  std::vector<float> zeta((size_t)4 * KS * 4);  /* [r][ks][g] */
  std::vector<float> x(T);
  for (int k = 0; k < K; k++)
    for (int c = 0; c < C; c++)
    {
      const float* zz = z + ((size_t)k * C + c) * KK;
      std::fill(zeta.begin(), zeta.end(), 0.0f);
      for (int fp = 0; fp <= P; fp++)
      {
        const int fs[4] = { fp, 2 * P - fp, 2 * P + fp, 4 * P - fp };
        float n[8], d[4][2];
        for (int slot = 0; slot < 4; slot++)
        {
          n[2 * slot] = zz[2 * fs[slot]];
          n[2 * slot + 1] = zz[2 * fs[slot] + 1];
        }
        coloredButterfly(n, &tb.wslot[((size_t)c * (P + 1) + fp) * 8], &tb.twid[(size_t)fp * 6], d);
        const int q = fp >> 2, g = fp & 3;
        for (int r = 0; r < 4; r++)
        {
          zeta[((size_t)r * KS + 2 * q) * 4 + g] = d[r][0];
          zeta[((size_t)r * KS + 2 * q + 1) * 4 + g] = d[r][1];
        }
      }
      for (int r = 0; r < 4; r++)
        for (int u = 0; u < P; u++)
        {
          const float* a = &tb.basis[(size_t)u * KS * 4];
          const float* b = &zeta[(size_t)r * KS * 4];
          float acc = 0.0f;
          for (int kk = 0; kk < KS * 4; kk++)
            acc = mppi::det::fma(a[kk], b[kk], acc);
          x[4 * u + r] = acc;
        }
      const float xs = x[offset_t];
      for (int t = 0; t < T; t++)
        eps[((size_t)k * T + t) * C + c] = mppi::det::fma(-tb.decay[t], xs, x[t]);
    }
}

/** what the engine's prologue computes for this (T, offset_t): the radix-4 flavour where it applies, else the dense table */
inline void coloredNoiseEngine(int K, int T, int C, const ColoredNoiseParams& p, int offset_t, const float* z, float* eps)
{
  if (coloredUseRadix4(T, offset_t))
    coloredNoiseRadix4(K, T, C, p, offset_t, z, eps);
  else
    coloredNoiseGemm(K, T, C, p, offset_t, z, eps);
}

/**
 * The engine's in-kernel spectrum draw: entry kk of (rollout k, control c) = element (kk >> 2) & 3 of Philox quad
 * ((kk >> 4) << 2) + (kk & 3), stream 1 + c (include/mppi_amd/sampling_distributions/colored_noise.hpp).
 * z out: [k_end - k_begin][C][T+1][2]
 */
inline void philoxSpectrum(uint64_t seed, uint32_t generation, int T, int C, int k_begin, int k_end, float* z)
{
  const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  const int KK = 2 * (T + 1);
  if ((T & 3) == 0 && T >= 16)
  {
    /* radix-4 form: quad 2 f' = (zr, zi) at the frequencies f' and 2P - f', quad 2 f' + 1 = at 2P + f' and 4P - f',
     * f' = 0 .. P.  A frequency two slots of one f' name (2P at f' = 0; P and 3P at f' = P) takes the FIRST slot's draw —
     * the later slot's weight is zero in the engine's table. */
    const int P = T / 4;
    for (int k = k_begin; k < k_end; k++)
      for (int c = 0; c < C; c++)
      {
        float* zz = z + ((size_t)(k - k_begin) * C + c) * KK;
        for (int fp = P; fp >= 0; fp--)  /* descending, slots D..A: the first slot of a repeated frequency is written last */
        {
          float n[8];
          for (int h = 0; h < 2; h++)
          {
            const uint32_t ctr[4] = { (uint32_t)(2 * fp + h), (uint32_t)k, generation, (uint32_t)(1 + c) };
            uint32_t x[4];
            philox4x32_10(ctr, key, x);
            boxMuller(x[0], x[1], &n[4 * h], &n[4 * h + 1]);
            boxMuller(x[2], x[3], &n[4 * h + 2], &n[4 * h + 3]);
          }
          const int fs[4] = { fp, 2 * P - fp, 2 * P + fp, 4 * P - fp };
          for (int slot = 3; slot >= 0; slot--)
          {
            zz[2 * fs[slot]] = n[2 * slot];
            zz[2 * fs[slot] + 1] = n[2 * slot + 1];
          }
        }
      }
    return;
  }
  for (int k = k_begin; k < k_end; k++)
    for (int c = 0; c < C; c++)
      for (int kk = 0; kk < KK; kk++)
      {
        const uint32_t quad = (uint32_t)(((kk >> 4) << 2) + (kk & 3));
        const uint32_t ctr[4] = { quad, (uint32_t)k, generation, (uint32_t)(1 + c) };
        uint32_t x[4];
        philox4x32_10(ctr, key, x);
        float n4[4];
        boxMuller(x[0], x[1], &n4[0], &n4[1]);
        boxMuller(x[2], x[3], &n4[2], &n4[3]);
        z[((size_t)(k - k_begin) * C + c) * KK + kk] = n4[(kk >> 2) & 3];
      }
}
}  // namespace oracle
#endif
