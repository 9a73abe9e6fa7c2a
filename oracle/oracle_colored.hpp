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
 * Two flavours of the same map z (Gaussian spectrum, layout [K][C][T+1][2]) -> eps [K][T][C]:
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

/**
 * The engine's in-kernel spectrum draw: entry kk of (rollout k, control c) = element (kk >> 2) & 3 of Philox quad
 * ((kk >> 4) << 2) + (kk & 3), stream 1 + c (include/mppi_amd/sampling_distributions/colored_noise.hpp).
 * z out: [k_end - k_begin][C][T+1][2]
 */
inline void philoxSpectrum(uint64_t seed, uint32_t generation, int T, int C, int k_begin, int k_end, float* z)
{
  const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  const int KK = 2 * (T + 1);
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
