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
 * kernargObject<T>(base, offset) hands out a pointer to the argument at `offset` that the optimiser cannot see through (the
 * segment pointer passes an empty volatile asm, kernargBase()): taken INSIDE the loop, the loads that hang off it cannot be hoisted
 * out of the loop, so each step loads what it needs, where it needs it, with s_load_dwordx{2,4,8,16} from the scalar cache (the
 * whole block is resident after the first step; SMEM instructions do not take VALU issue slots) into SGPRs that are free again
 * afterwards.  The address space survives the asm (address space 4 = constant): the loads stay scalar.
 *
 * Three uses that do not work, all found the hard way in round 6 (profiles/r06_step_source_ab.json):
 *   - as the OBJECT the per-step methods run on (`DYN_T* dynamics = kernargObject<DYN_T>(...)`) when the class keeps per-lane
 *     state in members: the matrix-core network models and the four-lane RACER models load their weights into registers in
 *     initializeDynamics — members of the kernel's local copy, which the argument block knows nothing of (AutoRally's costs came
 *     out wrong by 3e8 ulp, a null pointer was dereferenced);
 *   - as a per-step COPY of the read-only members (`params_ = src->params_`): the copy is loaded at the top of the step and is
 *     live from there to every use — spilled within the step instead of across the loop (642 v_readlane per pair of steps
 *     against 680 before; the view proper: 220);
 *   - as a pointer MEMBER the kernels set (`step_src_`): writing any member of a by-value argument makes it a private copy, and a
 *     copy with run-time-indexed member arrays cannot be split into registers — the whole object moved to scratch memory in the
 *     blockDim.y > 1 kernels.
 * What works: the class routes the READS of its read-only members through a function, S(), that returns the argument block
 * behind the opaque pointer (dynamics/racer_dubins/racer_dubins_elevation.hpp); per-lane state stays in the object.  A stateless
 * class (the cost classes: kernarg_viewable<T>) may be viewed whole.
 *
 * What it buys, measured: nothing where registers are not short — the step loops are bound by the latency of
 * their dependent VALU chain, the v_readlanes sit in its shadow, and an s_load + s_waitcnt in the chain is slower than a
 * v_readlane beside it (the vanilla kernels got 6-18 % SLOWER with the view on; it is off for them).  It pays in the Robust
 * kernels of the RACER models, which are VGPR-starved: fewer SGPR-spill lanes let the register allocator keep the vector state
 * out of scratch (complete model 3399 -> 1565 us together with the helper-wave traits).  Hence the per-class switches:
 * MPPI_STEP_SOURCE in the RACER classes, MPPI_RMPPI_COST_VIEW, and the A/B macros below.
 *
 * Argument offsets: the AMDGPU HIP ABI lays the explicit arguments out in order, each at the next multiple of its alignment,
 * starting at offset 0 of the segment (hidden arguments follow the explicit ones) — KernargLayout<Ts...>::offset<I>().
 * tests/test_kernarg_layout.py checks the formula against the `.offset` fields the compiler wrote into the metadata of a probe
 * kernel's code object (tests/probes/kernarg_layout_probe.hip) with arguments of mixed size and alignment.
 */
#ifndef MPPI_AMD_ENGINE_KERNARG_VIEW_HPP_
#define MPPI_AMD_ENGINE_KERNARG_VIEW_HPP_

#include <hip/hip_runtime.h>

#include <cstddef>
#include <type_traits>
#include <utility>

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

/** Is plugin class T a pure parameter block on the device — no device method ever stores to a member (no per-lane state, no
 *  cached pointers)?  `static constexpr bool MPPI_KERNARG_VIEWABLE = true;` in the class.  A role loop may then run T's per-step
 *  methods on kernargObject<T>(...) itself — the argument block behind the opaque pointer — instead of on the kernel's copy (whose
 *  parameters would be held in SGPRs, or pinned to VGPRs, across the loop).  The in-tree cost classes are; the samplers
 *  (thread mapping, noise stream) and the network dynamics (weights in registers) are not. */
template <class T, class = void>
struct kernarg_viewable : std::false_type
{
};
template <class T>
struct kernarg_viewable<T, std::void_t<decltype(T::MPPI_KERNARG_VIEWABLE)>> : std::integral_constant<bool, T::MPPI_KERNARG_VIEWABLE>
{
};

/** do the plugin classes that have an S() read their read-only members from the kernarg segment (A/B: -DMPPI_KERNARG_RELOAD=0
 *  restores the parameters-in-SGPRs form everywhere) */
#if !defined(MPPI_KERNARG_RELOAD)
#define MPPI_KERNARG_RELOAD 1
#endif
/** S() source of the four-lane RACER dynamics classes (dynamics/racer_dubins/racer_dubins_elevation.hpp): 0 the object itself
 *  (product), 1 the argument block, 2 a copy in LDS — the A/B of round 6 */
#if !defined(MPPI_STEP_SOURCE_QUAD)
#define MPPI_STEP_SOURCE_QUAD 0
#endif
/** do the cost waves of the role-pipelined kernels run a kernarg_viewable cost class off the argument block (1), or on a copy
 *  pinned to VGPRs as in rounds 3-5 (0)?  A/B switch.  Measured (profiles/r06_c_*, r06_d_*): AutoRally-NN 175.3 -> 179.9 us per
 *  launch, the RACER models +1 %, Robust AutoRally 394 -> 402; only the kernel that spills VGPRs to scratch gains (Robust
 *  complete RACER 2953 -> 2704 us): the s_loads' results return out of order, every use waits for lgkmcnt(0) — LDS traffic
 *  included — and that costs more than the VGPRs the copy occupies.  Off. */
#if !defined(MPPI_COST_KERNARG_VIEW)
#define MPPI_COST_KERNARG_VIEW 0
#endif
}  // namespace kernels
}  // namespace mppi

#endif
