/**
 * parallel_utils.hpp — threadIdx -> (index, step) helpers the plugins use to stride their loops over the intra-rollout
 * lanes (reference: include/mppi/utils/parallel_utils.cuh:12-180, namespace mppi::p1, same names).
 *
 * MI355X note: the rollout kernels are compiled per block shape.  When a rollout owns a single lane (blockDim.y == 1)
 * the kernel tells the compiler so (__builtin_assume on the workgroup-size / workitem-id builtins these helpers read),
 * every `for (i = p_index; i < N; i += p_step)` loop in a plugin unrolls completely, the per-rollout arrays live in
 * VGPRs and lane_sync() below compiles to nothing.  With blockDim.y > 1 the same source runs the reference's
 * LDS + barrier scheme.
 */
#ifndef MPPI_AMD_PLUGIN_PARALLEL_UTILS_HPP_
#define MPPI_AMD_PLUGIN_PARALLEL_UTILS_HPP_

#include <hip/hip_runtime.h>

#include <type_traits>

namespace mppi
{
namespace p1  // one index and one step out of the block shape
{
/** same enumerators, same order as the reference (utils/parallel_utils.cuh:12-28): a plugin may name any of them */
enum class Parallel1Dir : int
{
  THREAD_X = 0,
  THREAD_Y,
  THREAD_Z,
  THREAD_XY,
  THREAD_YX,
  THREAD_XZ,
  THREAD_ZX,
  THREAD_YZ,
  THREAD_ZY,
  THREAD_XYZ,
  GLOBAL_X,
  GLOBAL_Y,
  GLOBAL_Z,
  NONE,
};

#define MPPI_AMD_TID_X ((int)__builtin_amdgcn_workitem_id_x())
#define MPPI_AMD_TID_Y ((int)__builtin_amdgcn_workitem_id_y())
#define MPPI_AMD_TID_Z ((int)__builtin_amdgcn_workitem_id_z())
#define MPPI_AMD_DIM_X ((int)__builtin_amdgcn_workgroup_size_x())
#define MPPI_AMD_DIM_Y ((int)__builtin_amdgcn_workgroup_size_y())
#define MPPI_AMD_DIM_Z ((int)__builtin_amdgcn_workgroup_size_z())

namespace detail
{
/** work-item id / workgroup size / workgroup id / grid size (in workgroups) along axis A (0 = x, 1 = y, 2 = z): the builtins the
 *  rollout kernels put their __builtin_assume on, so a shape-specialised kernel folds every one of them */
template <int A>
__device__ inline int tid()
{
  return A == 0 ? MPPI_AMD_TID_X : (A == 1 ? MPPI_AMD_TID_Y : MPPI_AMD_TID_Z);
}
template <int A>
__device__ inline int dim()
{
  return A == 0 ? MPPI_AMD_DIM_X : (A == 1 ? MPPI_AMD_DIM_Y : MPPI_AMD_DIM_Z);
}
template <int A>
__device__ inline int bid()
{
  return A == 0 ? (int)__builtin_amdgcn_workgroup_id_x()
                : (A == 1 ? (int)__builtin_amdgcn_workgroup_id_y() : (int)__builtin_amdgcn_workgroup_id_z());
}
template <int A>
__device__ inline int grid()
{
  return A == 0 ? (int)gridDim.x : (A == 1 ? (int)gridDim.y : (int)gridDim.z);
}
/** "fast axis first": THREAD_AB walks axis A fastest, then axis B (reference: parallel_utils.cuh:66-140) */
template <int A, int B>
__device__ inline void two(int& i, int& s)
{
  i = tid<A>() + dim<A>() * tid<B>();
  s = dim<A>() * dim<B>();
}
}  // namespace detail

/**
 * reference: parallel_utils.cuh:30-193 — there one explicit specialisation per direction (and a host branch returning
 * (0, 1): nothing on this path evaluates a plugin on the host, so the functions are device-only here).  All fourteen
 * directions of the enum are defined.
 */
template <Parallel1Dir P_DIR>
__device__ inline void getParallel1DIndex(int& p_index, int& p_step)
{
  using D = Parallel1Dir;
  if constexpr (P_DIR == D::THREAD_X || P_DIR == D::THREAD_Y || P_DIR == D::THREAD_Z)
  {
    constexpr int A = (int)P_DIR - (int)D::THREAD_X;
    p_index = detail::tid<A>();
    p_step = detail::dim<A>();
  }
  else if constexpr (P_DIR == D::THREAD_XY)
    detail::two<0, 1>(p_index, p_step);
  else if constexpr (P_DIR == D::THREAD_YX)
    detail::two<1, 0>(p_index, p_step);
  else if constexpr (P_DIR == D::THREAD_XZ)
    detail::two<0, 2>(p_index, p_step);
  else if constexpr (P_DIR == D::THREAD_ZX)
    detail::two<2, 0>(p_index, p_step);
  else if constexpr (P_DIR == D::THREAD_YZ)
    detail::two<1, 2>(p_index, p_step);
  else if constexpr (P_DIR == D::THREAD_ZY)
    detail::two<2, 1>(p_index, p_step);
  else if constexpr (P_DIR == D::THREAD_XYZ)
  {
    p_index = MPPI_AMD_TID_X + MPPI_AMD_DIM_X * (MPPI_AMD_TID_Y + MPPI_AMD_DIM_Y * MPPI_AMD_TID_Z);
    p_step = MPPI_AMD_DIM_X * MPPI_AMD_DIM_Y * MPPI_AMD_DIM_Z;
  }
  else if constexpr (P_DIR == D::GLOBAL_X || P_DIR == D::GLOBAL_Y || P_DIR == D::GLOBAL_Z)
  {  // grid-stride over one axis (parallel_utils.cuh:142-180)
    constexpr int A = (int)P_DIR - (int)D::GLOBAL_X;
    p_index = detail::tid<A>() + detail::dim<A>() * detail::bid<A>();
    p_step = detail::grid<A>() * detail::dim<A>();
  }
  else
  {
    static_assert(P_DIR == D::NONE, "unknown Parallel1Dir");
    p_index = 0;
    p_step = 1;
  }
}

/**
 * N elements of a2 (from off2) into a1 (from off1), strided over the lanes P_DIR names; 16- and 8-byte moves when count
 * and both offsets allow (reference: parallel_utils.cuh:195-222 loadArrayParallel<P_DIR, T>(…, N): the offsets, not the
 * addresses, decide — the arrays themselves are assumed aligned, as there).
 */
template <Parallel1Dir P_DIR = Parallel1Dir::THREAD_Y, class T = float>
__device__ inline void loadArrayParallel(T* __restrict__ a1, const int off1, const T* __restrict__ a2, const int off2, const int N)
{
  int p_index, p_step;
  getParallel1DIndex<P_DIR>(p_index, p_step);
  struct alignas(4 * sizeof(T)) Quad
  {
    T v[4];
  };
  struct alignas(2 * sizeof(T)) Pair
  {
    T v[2];
  };
  if (sizeof(Quad) <= 16 && ((N | off1 | off2) & 3) == 0)
  {
    for (int i = p_index; i < N / 4; i += p_step)
      reinterpret_cast<Quad*>(a1 + off1)[i] = reinterpret_cast<const Quad*>(a2 + off2)[i];
  }
  else if (sizeof(Pair) <= 16 && ((N | off1 | off2) & 1) == 0)
  {
    for (int i = p_index; i < N / 2; i += p_step)
      reinterpret_cast<Pair*>(a1 + off1)[i] = reinterpret_cast<const Pair*>(a2 + off2)[i];
  }
  else
  {
    for (int i = p_index; i < N; i += p_step)
      a1[off1 + i] = a2[off2 + i];
  }
}

/** compile-time count (reference: parallel_utils.cuh:224-228).  Element-wise on purpose: with one lane per rollout the loop
 *  unrolls into register moves, which a reinterpret_cast to a wider type would force through memory. */
template <int N, Parallel1Dir P_DIR = Parallel1Dir::THREAD_Y, class T = float>
__device__ inline void loadArrayParallel(T* __restrict__ a1, const int off1, const T* __restrict__ a2, const int off2)
{
  int p_index, p_step;
  getParallel1DIndex<P_DIR>(p_index, p_step);
  for (int i = p_index; i < N; i += p_step)
  {
    a1[off1 + i] = a2[off2 + i];
  }
}
}  // namespace p1

namespace p2  // two indices and two steps (reference: parallel_utils.cuh:231-345)
{
enum class Parallel2Dir : int
{
  THREAD_XY = 0,
  THREAD_XZ,
  THREAD_YZ,
  THREAD_YX,
  THREAD_ZX,
  THREAD_ZY,
  NONE
};

template <Parallel2Dir P_DIR>
__device__ inline void getParallel2DIndex(int& p1_index, int& p2_index, int& p1_step, int& p2_step)
{
  using D = Parallel2Dir;
  // first axis, second axis of every direction, in the enum's order
  constexpr int first[6] = { 0, 0, 1, 1, 2, 2 };
  constexpr int second[6] = { 1, 2, 2, 0, 0, 1 };
  if constexpr (P_DIR == D::NONE)
  {
    p1_index = p2_index = 0;
    p1_step = p2_step = 1;
  }
  else
  {
    constexpr int A = first[(int)P_DIR], B = second[(int)P_DIR];
    p1_index = p1::detail::tid<A>();
    p1_step = p1::detail::dim<A>();
    p2_index = p1::detail::tid<B>();
    p2_step = p1::detail::dim<B>();
  }
}
}  // namespace p2

/**
 * Barrier between the phases of one rollout step.  The lanes of a rollout exchange data through LDS only when
 * blockDim.y > 1; a rollout that owns one lane needs no barrier at all, and the branch folds away at compile time in
 * the single-lane kernels (see the note at the top of this file).  Block-uniform by construction.
 */
__device__ inline void lane_sync()
{
  if (__builtin_amdgcn_workgroup_size_y() != 1)
  {
    __syncthreads();
  }
}

/**
 * May the device methods of plugin class T run on the ROLE-SEPARATED rollout kernels (engine/rollout_pipeline_kernel.hpp,
 * rmppi_pipeline_kernel.hpp)?  There the waves of a block have different jobs — sampler waves, a dynamics wave (or four), cost
 * waves — and only the wave whose job it is calls a plugin's per-step methods: a block barrier (__syncthreads()) inside
 * Dynamics::step / computeDynamics, Cost::computeRunningCost or a sampler's per-step methods would wait for waves that never
 * arrive, i.e. hang the GPU.  The reference's own Dynamics::step has two such barriers (dynamics/dynamics.cu:138,140) and a
 * good part of its models call __syncthreads() themselves, so the engine ASSUMES BARRIERS unless the class says otherwise:
 *
 *     static constexpr bool MPPI_BARRIER_FREE_STEP = true;   // in the plugin class
 *
 * declares that no device method the rollout kernels call per step contains a block barrier (mppi::lane_sync() does not
 * count: it is a barrier only when blockDim.y > 1, and the role-separated kernels give every rollout one lane).  A class without the
 * declaration is a class with barriers: it runs on the fused kernels (engine/rollout_kernel.hpp), where every thread of the
 * block reaches every plugin call, as in the reference.  The declaration is deliberately NOT made in the CRTP bases
 * (plugin/dynamics.hpp, plugin/cost.hpp) — a user's class would inherit it; the in-tree models make it themselves.
 * Consumers: controllers_templated.hpp chooses PIPELINE from it; ModelT refuses PIPELINE = true / a replicated-lane form
 * without it, at registration (mppi_register_model_checked) and again at mppi_create.
 */
template <class T, class = void>
struct barrier_free_step : std::false_type
{
};
template <class T>
struct barrier_free_step<T, std::void_t<decltype(T::MPPI_BARRIER_FREE_STEP)>> : std::integral_constant<bool, T::MPPI_BARRIER_FREE_STEP>
{
};
}  // namespace mppi

#endif
