/**
 * math_utils.hpp — small helpers with the reference's names (reference: include/mppi/utils/math_utils.h:15-110, 738-747).
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
