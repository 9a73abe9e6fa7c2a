/**
 * kernarg_view.hpp — plugin objects read from the KERNARG SEGMENT inside the step loop, instead of living in SGPRs across it.
 *
 * The rollout kernels take their plugin objects by value (DESIGN.md §2 "kernel arguments by value"): the objects are the kernel's
 * argument block, in constant memory behind s[kernarg_ptr].  Written the plain way — `DYN_T* dynamics = &dynamics_obj;` — every
 * parameter the step loop reads is a loop-invariant scalar load; the compiler hoists all of them in front of the loop and keeps
 * them in SGPRs across it.  A wave has ~100 SGPRs.  The complete RACER model's argument block is 1560 B = 390 words, the Robust
 * kernels carry five plugin objects: round 5's code objects show 316-438 SPILLED SGPRs (written to VGPR lanes once, read back
 * with a v_readlane_b32 — a VALU issue slot plus hazard s_nops — at every use: 680 v_readlane + 456 s_nop in the 7740 instructions
 * of a pair of steps of the complete model's dynamics wave).
 *
 * kernargObject<T>(offset) hands out a pointer to argument `offset` that the optimiser cannot see through (the segment pointer
 * passes an empty volatile asm): taken INSIDE the loop, the loads that hang off it cannot be hoisted out of the loop, so each
 * step loads what it needs with s_load_dwordx{2,4,8,16} from the scalar cache (the whole block is resident after the first
 * step; SMEM instructions do not take VALU issue slots) into SGPRs that are free again at the end of the step.  The address space
 * survives the asm (address space 4 = constant): the loads stay scalar.
 *
 * The objects are never written by device code (they could not be: a by-value argument the kernel stored to would have been
 * copied to scratch by the front end — the kernels that use this have 0 B of scratch for it), so a const view is sufficient; the
 * plugin methods are non-const by the reference's signatures, hence the const_cast inside.
 *
 * Argument offsets: the AMDGPU HIP ABI lays the explicit arguments out in order, each at the next multiple of its alignment,
 * starting at offset 0 of the segment (hidden arguments follow the explicit ones) — KernargLayout<Ts...>::offset<I>().
 * tests/test_kernarg_layout.py checks the formula against the `.offset` fields the compiler wrote into the code object's
 * metadata for every kernel that uses it.
 */
#ifndef MPPI_AMD_ENGINE_KERNARG_VIEW_HPP_
#define MPPI_AMD_ENGINE_KERNARG_VIEW_HPP_

#include <hip/hip_runtime.h>

#include <cstddef>
#include <type_traits>

namespace mppi
{
namespace kernels
{
template <class... Ts>
struct KernargLayout
{
  template <int I>
  static constexpr size_t offset()
  {
    constexpr size_t sizes[] = { sizeof(Ts)... };
    constexpr size_t aligns[] = { alignof(Ts)... };
    size_t off = 0;
    for (int i = 0; i <= I; i++)
    {
      off = (off + aligns[i] - 1) / aligns[i] * aligns[i];
      if (i < I)
        off += sizes[i];
    }
    return off;
  }
};

typedef const __attribute__((address_space(4))) char* kernarg_ptr_t;

/** the kernarg segment's base, opaque to the optimiser from here on (see the file comment): call it where the reloads may start */
__device__ inline kernarg_ptr_t kernargBase()
{
  kernarg_ptr_t p = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

/** pointer to the kernel argument at byte `offset` of the segment `base` points at */
template <class T>
__device__ inline T* kernargObject(kernarg_ptr_t base, const size_t offset)
{
  return const_cast<T*>(reinterpret_cast<const T*>((const char*)(base + offset)));
}

/** does plugin class T ask for its read-only members to be re-read from the kernel's argument block at every step
 *  (plugin/dynamics.hpp: refreshStepInvariants)? */
template <class T, class = void>
struct refreshes_step_invariants : std::false_type
{
};
template <class T>
struct refreshes_step_invariants<T, std::void_t<decltype(T::MPPI_REFRESH_STEP_INVARIANTS)>>
  : std::integral_constant<bool, T::MPPI_REFRESH_STEP_INVARIANTS>
{
};

/** top of a step of a role loop: `obj` (the kernel's by-value argument at byte `offset` of the argument block, possibly carrying
 *  per-lane state in some of its members) takes its read-only members from the argument block again */
template <class T>
__device__ inline void refreshStepInvariants(T* obj, const size_t offset)
{
  if constexpr (refreshes_step_invariants<T>::value)
    obj->refreshStepInvariants(kernargObject<T>(kernargBase(), offset));
}

/** which role loops read their plugin objects from the kernarg segment per step (A/B: -DMPPI_KERNARG_RELOAD=0 restores the
 *  objects-in-SGPRs form everywhere) */
#if !defined(MPPI_KERNARG_RELOAD)
#define MPPI_KERNARG_RELOAD 1
#endif
}  // namespace kernels
}  // namespace mppi

#endif
