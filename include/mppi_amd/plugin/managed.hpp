/**
 * managed.hpp — base of every plugin object (reference: include/mppi/utils/managed.cuh:54-138, same member names).
 *
 * A plugin lives on the host and as a byte copy on the device (GPUSetup), exactly as in the reference.  Differences,
 * on purpose: no virtual functions and no logger member, so the object is trivially copyable and the device copy is a
 * well-defined memcpy; errors are returned (hipError_t) instead of exit()ing (reference: utils/gpu_err_chk.cuh:32-40).
 */
#ifndef MPPI_AMD_PLUGIN_MANAGED_HPP_
#define MPPI_AMD_PLUGIN_MANAGED_HPP_

#include <hip/hip_runtime.h>

namespace mppi
{
class Managed
{
public:
  hipStream_t stream_ = 0;  ///< stream the object's copies are enqueued on
  bool GPUMemStatus_ = false;

  void bindToStream(hipStream_t stream)
  {
    stream_ = stream;
  }
  /** LDS bytes requested once per block / once per rollout slot (reference: managed.cuh:104-111) */
  __device__ __host__ int getGrdSharedSizeBytes() const
  {
    return SHARED_MEM_REQUEST_GRD_BYTES;
  }
  __device__ __host__ int getBlkSharedSizeBytes() const
  {
    return SHARED_MEM_REQUEST_BLK_BYTES;
  }

protected:
  template <class T>
  static hipError_t GPUSetup(T* host_ptr, T** device_ptr)
  {
    hipError_t e = hipMalloc((void**)device_ptr, sizeof(T));
    if (e != hipSuccess)
      return e;
    e = hipMemcpyAsync(*device_ptr, host_ptr, sizeof(T), hipMemcpyHostToDevice, host_ptr->stream_);
    if (e != hipSuccess)
      return e;
    host_ptr->GPUMemStatus_ = true;
    return hipStreamSynchronize(host_ptr->stream_);
  }
  int SHARED_MEM_REQUEST_GRD_BYTES = 0;
  int SHARED_MEM_REQUEST_BLK_BYTES = 0;
};

/** reference: core/mppi_common.cu:1557-1564 calcClassSharedMemSize — Grd + Blk * (slots), each rounded to 16 B */
template <class T>
__host__ __device__ inline int calcClassSharedMemSize(const T* obj, int slots)
{
  const int grd = ((obj->getGrdSharedSizeBytes() + 15) / 16) * 16;
  const int blk = ((obj->getBlkSharedSizeBytes() + 15) / 16) * 16;
  return grd + blk * slots;
}
}  // namespace mppi
#endif
