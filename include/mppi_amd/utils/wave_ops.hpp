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
}  // namespace wave
}  // namespace mppi

#endif
