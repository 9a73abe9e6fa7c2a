/**
 * oracle_rng.hpp — CPU restatement of the engine's counter-based Gaussian noise.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference fills control_samples_d_ with curandGenerateNormal (XORWOW, host API;
 * sampling_distributions/gaussian/gaussian.cu:380-394, controllers/controller.cu:192-207).  cuRAND is a closed
 * third-party library that is absent here and whose stream the reference itself never pins (its sampler tests are
 * statistical only, tests/sampling_distributions/colored_noise_tests.cu:98-209) — **RNG-stream parity with the
 * reference is unpinned**; parity is defined downstream of eps.  What IS pinned here is the engine's own generator:
 * Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11 — the generator behind
 * rocRAND/cuRAND's PHILOX4_32_10 type), restated from the paper and checked against the Random123 known-answer
 * vectors in tests/test_oracle_kat.py, followed by a Box-Muller transform written with det_math.
 *
 * Counter layout (shard-count invariant: depends only on the GLOBAL rollout index):
 *   element j = t*C + c of rollout k's row of eps[K][T][C];   quad = j / 4, lane = j % 4
 *   counter = { quad, k, generation, stream }   key = { lo32(seed), hi32(seed) }
 *   generation = number of generateSamples calls so far (cuRAND's advancing offset), stream = 0 for eps.
 *   (x0,x1) -> Box-Muller -> lanes 0,1;  (x2,x3) -> lanes 2,3.
 */
#ifndef MPPI_ORACLE_RNG_HPP_
#define MPPI_ORACLE_RNG_HPP_

#include <cstdint>
#include "mppi_amd/det_math.h"

namespace oracle
{
inline void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4])
{
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int round = 0; round < 10; round++)
  {
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

/** u in (0,1]: (x + 0.5) * 2^-32 evaluated as one fused op on the correctly rounded float(x). */
inline float u01_open_low(uint32_t x)
{
  return mppi::det::fma((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

inline void boxMuller(uint32_t xa, uint32_t xb, float* z0, float* z1)
{
  const float u1 = u01_open_low(xa);
  const float u2 = u01_open_low(xb);
  const float r = mppi::det::sqrt(-2.0f * mppi::det::log(u1));
  float s, c;
  mppi::det::sincos(MPPI_DET_TWO_PI * u2, &s, &c);
  *z0 = r * c;
  *z1 = r * s;
}

/** eps[K][T][C] for rollouts [k_begin, k_end) of a K-rollout problem */
inline void philoxNormal(uint64_t seed, uint32_t generation, uint32_t stream, int K, int T, int C, int k_begin,
                         int k_end, float* eps_out)
{
  const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  const int TC = T * C;
  for (int k = k_begin; k < k_end; k++)
  {
    for (int q = 0; q * 4 < TC; q++)
    {
      const uint32_t ctr[4] = { (uint32_t)q, (uint32_t)k, generation, stream };
      uint32_t x[4];
      philox4x32_10(ctr, key, x);
      float z[4];
      boxMuller(x[0], x[1], &z[0], &z[1]);
      boxMuller(x[2], x[3], &z[2], &z[3]);
      for (int l = 0; l < 4 && q * 4 + l < TC; l++)
        eps_out[(size_t)(k - k_begin) * TC + q * 4 + l] = z[l];
    }
  }
}
}  // namespace oracle
#endif
