/**
 * philox.h — counter-based Gaussian noise for the sampler: Philox4x32-10 + Box-Muller, host/device.
 *
 * Replaces the reference's curandGenerateNormal fill of control_samples_d_ (reference:
 * include/mppi/sampling_distributions/gaussian/gaussian.cu:380-394; generator created in
 * include/mppi/controllers/controller.cu:192-207).  Instead of materialising eps[K][T][C] in HBM and re-reading it, the
 * rollout kernel draws eps where it is consumed.  Philox4x32-10 is the generator behind rocRAND's
 * ROCRAND_RNG_PSEUDO_PHILOX4_32_10; the host-API rocRAND fill remains available as an alternative noise source
 * (mppi_config.noise_source = MPPI_NOISE_ROCRAND_HOST, see include/mppi_amd.h).
 *
 * The stream depends only on (seed, generation, GLOBAL element index), so any sharding of the K rollouts over GPUs
 * draws exactly the same noise — the property SURVEY.md §8e asks for.
 *   element j = t*C + c of rollout k's row;  quad q = j / 4, lane = j % 4
 *   counter = {q, k_global, generation, stream};  key = {lo32(seed), hi32(seed)}
 *   (x0,x1) -> Box-Muller -> lanes 0,1;  (x2,x3) -> lanes 2,3
 * Quads never straddle rollouts, so the 64 lanes of a wave (64 consecutive rollouts at the same t) need a new quad at
 * the same step: the draw is a wave-uniform branch inside the rollout loop.
 * Box-Muller uses det_math so host and device produce identical bits (tests/test_rng.py).
 */
#ifndef MPPI_AMD_PHILOX_H_
#define MPPI_AMD_PHILOX_H_

#include <stdint.h>
#include "mppi_amd/det_math.h"

namespace mppi
{
namespace rng
{
struct uint4_t
{
  uint32_t x, y, z, w;
};

MPPI_HD static inline uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

MPPI_HD static inline uint4_t philox4x32_10(uint4_t c, uint32_t k0, uint32_t k1)
{
#pragma unroll
  for (int i = 0; i < 10; i++)
  {
    const uint32_t hi0 = mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    uint4_t n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
    n.w = lo0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

/** (x + 0.5) * 2^-32 in (0, 1] */
MPPI_HD static inline float u01(uint32_t x)
{
  return det::fma((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

MPPI_HD static inline void box_muller(uint32_t xa, uint32_t xb, float* z0, float* z1)
{
  const float u1 = u01(xa);
  const float u2 = u01(xb);
#if defined(MPPI_EXPERIMENT_CHEAP_NORMAL)
  /* A/B ONLY (never in a product build): a draw of a third of the instructions — is the Cartpole dynamics wave waiting for its
   * sampler waves?  (tools/ab_kernels.py cartpole on `buildlib.py --variant cheap cartpole.hip -DMPPI_EXPERIMENT_CHEAP_NORMAL`) */
  *z0 = u1 - 0.5f;
  *z1 = u2 - 0.5f;
  return;
#endif
  const float r = det::sqrt(-2.0f * det::log(u1));
  float s, c;
  det::sincos(MPPI_DET_TWO_PI * u2, &s, &c);
  *z0 = r * c;
  *z1 = r * s;
}

/** four N(0,1) draws: quad `quad` of global rollout `rollout` */
MPPI_HD static inline void normal4(uint64_t seed, uint32_t generation, uint32_t stream, uint32_t rollout, uint32_t quad,
                                   float z[4])
{
  uint4_t c;
  c.x = quad;
  c.y = rollout;
  c.z = generation;
  c.w = stream;
  const uint4_t r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  box_muller(r.x, r.y, &z[0], &z[1]);
  box_muller(r.z, r.w, &z[2], &z[3]);
}
}  // namespace rng
}  // namespace mppi

#endif
