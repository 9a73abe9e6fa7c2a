/**
 * wave_ops.hpp — wave64 cross-lane primitives for gfx950 used by the MFMA network forwards.
 *
 * The MFMA 16x16x4 fragment layouts (cdna_hip_programming.md §3) put a 16-wide column index in `lane & 15` and a
 * 4-wide group index g in `lane >> 4` (the four 16-lane ROWS of a wave).  A layer's output arrives in the D layout —
 * lane group g holds rows 4g + i, i = 0..3 — while the next layer's B operand wants rows 4s + g in k-step s: a 4x4
 * transpose between (lane group, register).  gfx950 has two swap instructions that do this without touching LDS:
 *   v_permlane32_swap a, b : a' = [a.lo32, b.lo32], b' = [a.hi32, b.hi32]       (swap a's upper half with b's lower half)
 *   v_permlane16_swap a, b : swap the odd 16-lane rows of a with the even rows of b
 * Two of each transpose the 4x4 (the classic 2x2-blocks-then-elements butterfly).
 * The reference does this re-layout through shared memory + __syncthreads (include/mppi/utils/nn_helpers/
 * fnn_helper.cu:470-480: curr_act/next_act swap after every layer).
 */
#ifndef MPPI_AMD_WAVE_OPS_HPP_
#define MPPI_AMD_WAVE_OPS_HPP_

#include <hip/hip_runtime.h>

namespace mppi
{
namespace wave
{
/**
 * In : lane group g = lane >> 4 holds v[i] = X[4g + i]   (i = 0..3; every column lane & 15 independently)
 * Out: lane group g holds v[s] = X[4s + g]
 */
__device__ inline void transpose4x4(float (&v)[4])
{
  unsigned a0 = __float_as_uint(v[0]), a1 = __float_as_uint(v[1]), a2 = __float_as_uint(v[2]),
           a3 = __float_as_uint(v[3]);
  // 2x2 blocks: groups {0,1} <-> {2,3} against registers {0,1} <-> {2,3}
  auto r02 = __builtin_amdgcn_permlane32_swap(a0, a2, false, false);
  auto r13 = __builtin_amdgcn_permlane32_swap(a1, a3, false, false);
  a0 = r02[0];
  a2 = r02[1];
  a1 = r13[0];
  a3 = r13[1];
  // inside each block: group parity against register parity
  auto r01 = __builtin_amdgcn_permlane16_swap(a0, a1, false, false);
  auto r23 = __builtin_amdgcn_permlane16_swap(a2, a3, false, false);
  v[0] = __uint_as_float(r01[0]);
  v[1] = __uint_as_float(r01[1]);
  v[2] = __uint_as_float(r23[0]);
  v[3] = __uint_as_float(r23[1]);
}
/**
 * v, one value per lane group g = lane >> 4 (per column lane & 15): returns (v_0 + v_1) + (v_2 + v_3) in all four groups.
 * v_permlane16_swap(a, a) leaves [row-pair's even value, row-pair's odd value] in both rows of a pair, v_permlane32_swap(c, c)
 * [lower half's value, upper half's value] in both halves: the two lanes of a pair add the same operands in the same order.
 */
__device__ inline float sumOverLaneGroups(const float v)
{
  const unsigned b = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  const float s01 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const unsigned c = __float_as_uint(s01);
  auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
/* ---- wave64 all-reduce without LDS: four DPP steps inside the rows of 16 lanes (xor 1, xor 2, half-row mirror, row mirror:
 * a butterfly — both lanes of a pair form the same commutative sum, so all 16 lanes of a row end with identical bits), then
 * the four row results through v_readlane in a fixed order.  ~20 issue slots against six dependent ds_bpermute round trips
 * (~0.3 us) for a __shfl_xor tree: the merge kernel is a chain of latencies, and its three reductions were 1.5 us of it
 * (in-kernel timers, round 4: profiles/r04_combine_timing_before.json); the block-softmin epilogue of the rollout kernels uses
 * them as well. */
template <int CTRL>
__device__ inline float dppMoveF(const float v)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ inline double dppMoveD(const double v)
{
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;  // quad_perm [1,0,3,2] / [2,3,0,1]

__device__ inline float readLaneF(const float v, const int l)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ inline double readLaneD(const double v, const int l)
{
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ inline float waveAllMin(float v)
{
  v = fminf(v, dppMoveF<DPP_XOR1>(v));
  v = fminf(v, dppMoveF<DPP_XOR2>(v));
  v = fminf(v, dppMoveF<DPP_HALF_MIRROR>(v));
  v = fminf(v, dppMoveF<DPP_MIRROR>(v));
  return fminf(fminf(readLaneF(v, 0), readLaneF(v, 16)), fminf(readLaneF(v, 32), readLaneF(v, 48)));
}
__device__ inline float waveAllSum(float v)
{
  v += dppMoveF<DPP_XOR1>(v);
  v += dppMoveF<DPP_XOR2>(v);
  v += dppMoveF<DPP_HALF_MIRROR>(v);
  v += dppMoveF<DPP_MIRROR>(v);
  return (readLaneF(v, 0) + readLaneF(v, 16)) + (readLaneF(v, 32) + readLaneF(v, 48));
}
__device__ inline double waveAllSum(double v)
{
  v += dppMoveD<DPP_XOR1>(v);
  v += dppMoveD<DPP_XOR2>(v);
  v += dppMoveD<DPP_HALF_MIRROR>(v);
  v += dppMoveD<DPP_MIRROR>(v);
  return (readLaneD(v, 0) + readLaneD(v, 16)) + (readLaneD(v, 32) + readLaneD(v, 48));
}

}  // namespace wave
}  // namespace mppi

#endif
