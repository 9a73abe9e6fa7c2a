/**
 * math_utils.hpp — small helpers with the reference's names (reference: include/mppi/utils/math_utils.h:15-110, 149-156, 738-747).
 */
#ifndef MPPI_AMD_PLUGIN_MATH_UTILS_HPP_
#define MPPI_AMD_PLUGIN_MATH_UTILS_HPP_

#include <hip/hip_runtime.h>
#include "mppi_amd/det_math.h"

#ifndef SQ
#define SQ(a) ((a) * (a))
#endif

namespace mppi
{
namespace math
{
/** reference: utils/math_utils.h nearest_multiple_4 */
inline __host__ __device__ int nearest_multiple_4(const int& a)
{
  return ((a + 3) / 4) * 4;
}
inline __host__ __device__ int int_ceil(const int& a, const int& b)
{
  return a == 0 ? a : (a - 1) / b + 1;
}
inline __host__ __device__ float clamp(float value, float min, float max)
{
  return fminf(fmaxf(value, min), max);
}
/** reference: utils/math_utils.h:744-747 — the float overload the dynamics base class resolves to */
inline __host__ __device__ float sign(float value)
{
  return value >= 0 ? 1 : -1;
}
/** reference: utils/math_utils.h:90-94 */
inline __host__ __device__ float linInterp(const float x, const float x_min, const float x_max, const float y_min,
                                           const float y_max)
{
  return (x - x_min) / (x_max - x_min) * (y_max - y_min) + y_min;
}
/** |r - centre line| in units of half the track width (reference: utils/math_utils.h:149-156) */
inline __host__ __device__ float normDistFromCenter(const float r, const float r_in, const float r_out)
{
  const float r_center = (r_in + r_out) / 2.0f;
  const float r_width = r_out - r_in;
  return fabsf(r - r_center) / (r_width * 0.5f);
}
}  // namespace math
}  // namespace mppi

/** reference: include/mppi/utils/angle_utils.cuh:21-27; evaluated with the bit-reproducible fmod of det_math.h */
namespace angle_utils
{
__host__ __device__ static inline float normalizeAngle(float angle)
{
  return mppi::det::normalizeAngle(angle);
}
/** the same value for |angle| < 1e7 rad, without the out-of-range test on the dependent chain (det_math.h) */
__host__ __device__ static inline float normalizeAngleBounded(float angle)
{
  return mppi::det::normalizeAngleBounded(angle);
}
__host__ __device__ static inline float shortestAngularDistance(float from, float to)
{
  return normalizeAngle(to - from);
}
}  // namespace angle_utils

#endif
