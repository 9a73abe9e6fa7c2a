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

namespace mppi
{
namespace p1
{
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
  NONE,
};

template <Parallel1Dir P_DIR>
inline __device__ void getParallel1DIndex(int& p_index, int& p_step);

#define MPPI_AMD_TID_X ((int)__builtin_amdgcn_workitem_id_x())
#define MPPI_AMD_TID_Y ((int)__builtin_amdgcn_workitem_id_y())
#define MPPI_AMD_TID_Z ((int)__builtin_amdgcn_workitem_id_z())
#define MPPI_AMD_DIM_X ((int)__builtin_amdgcn_workgroup_size_x())
#define MPPI_AMD_DIM_Y ((int)__builtin_amdgcn_workgroup_size_y())
#define MPPI_AMD_DIM_Z ((int)__builtin_amdgcn_workgroup_size_z())

template <>
inline __device__ void getParallel1DIndex<Parallel1Dir::THREAD_X>(int& p_index, int& p_step)
{
  p_index = MPPI_AMD_TID_X;
  p_step = MPPI_AMD_DIM_X;
}
template <>
inline __device__ void getParallel1DIndex<Parallel1Dir::THREAD_Y>(int& p_index, int& p_step)
{
  p_index = MPPI_AMD_TID_Y;
  p_step = MPPI_AMD_DIM_Y;
}
template <>
inline __device__ void getParallel1DIndex<Parallel1Dir::THREAD_Z>(int& p_index, int& p_step)
{
  p_index = MPPI_AMD_TID_Z;
  p_step = MPPI_AMD_DIM_Z;
}
template <>
inline __device__ void getParallel1DIndex<Parallel1Dir::THREAD_XY>(int& p_index, int& p_step)
{
  p_index = MPPI_AMD_TID_X + MPPI_AMD_DIM_X * MPPI_AMD_TID_Y;
  p_step = MPPI_AMD_DIM_X * MPPI_AMD_DIM_Y;
}
template <>
inline __device__ void getParallel1DIndex<Parallel1Dir::THREAD_YX>(int& p_index, int& p_step)
{
  p_index = MPPI_AMD_TID_Y + MPPI_AMD_DIM_Y * MPPI_AMD_TID_X;
  p_step = MPPI_AMD_DIM_Y * MPPI_AMD_DIM_X;
}
template <>
inline __device__ void getParallel1DIndex<Parallel1Dir::THREAD_XZ>(int& p_index, int& p_step)
{
  p_index = MPPI_AMD_TID_X + MPPI_AMD_DIM_X * MPPI_AMD_TID_Z;
  p_step = MPPI_AMD_DIM_X * MPPI_AMD_DIM_Z;
}
template <>
inline __device__ void getParallel1DIndex<Parallel1Dir::THREAD_XYZ>(int& p_index, int& p_step)
{
  p_index = MPPI_AMD_TID_X + MPPI_AMD_DIM_X * (MPPI_AMD_TID_Y + MPPI_AMD_DIM_Y * MPPI_AMD_TID_Z);
  p_step = MPPI_AMD_DIM_X * MPPI_AMD_DIM_Y * MPPI_AMD_DIM_Z;
}
template <>
inline __device__ void getParallel1DIndex<Parallel1Dir::NONE>(int& p_index, int& p_step)
{
  p_index = 0;
  p_step = 1;
}

/** N-float copy strided over the intra-rollout lanes (reference: parallel_utils.cuh loadArrayParallel<N>) */
template <int N, Parallel1Dir P_DIR = Parallel1Dir::THREAD_Y>
inline __device__ void loadArrayParallel(float* __restrict__ a1, const int off1, const float* __restrict__ a2,
                                         const int off2)
{
  int p_index, p_step;
  getParallel1DIndex<P_DIR>(p_index, p_step);
  for (int i = p_index; i < N; i += p_step)
  {
    a1[off1 + i] = a2[off2 + i];
  }
}
}  // namespace p1

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
}  // namespace mppi

#endif
