// What does it cost to let rollout kernel i+1 start (launch ramp, LDS allocation, the first mu-independent work) while merge
// kernel i is still running?  Stand-ins: R = 256 workgroups x 256 threads with 100 KB of LDS each (one per CU), busy for
// ~20 us after an optional spin on a device flag; C = 2 small workgroups busy for ~4 us that raise the flag at their end.
//   (1) one stream, R C R C ...                                  — today's structure: two in-order boundaries per iteration
//   (2) two streams, every kernel waits for its predecessor through an event — the price of a cross-queue dependency
//   (3) overlap: R_i and C_i in stream i & 1, R_{i+1} in the other stream behind an event on R_i, spinning on C_i's flag
// Build: hipcc --offload-arch=gfx950 -O3 two_queue.hip -o two_queue ; run on the GPU box.  Prints us per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x)                                                                  \
  do                                                                           \
  {                                                                            \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess)                                                      \
    {                                                                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                           \
      return 1;                                                                \
    }                                                                          \
  } while (0)

__device__ inline void busy(unsigned long long ticks)  // 100 MHz wall clock
{
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks)
    __builtin_amdgcn_s_sleep(1);
}

__global__ void __launch_bounds__(256) R(const unsigned* flag, unsigned need, unsigned long long work, float* sink)
{
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;  // the launch ramp of a 100 KB-LDS block
  __syncthreads();
  if (flag && threadIdx.x == 0)
  {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need && wall_clock64() - t0 < 200000000ull)
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
    busy(work);
  __syncthreads();
  if (sink && lds[threadIdx.x] < 0.0f)
    *sink = 1.0f;
}

__global__ void __launch_bounds__(256) C(unsigned* flag, unsigned seq, unsigned long long work)
{
  if ((threadIdx.x & 63) == 0)
    busy(work);
  __syncthreads();
  if (flag && blockIdx.x == 0 && threadIdx.x == 0)
  {
    __threadfence();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main()
{
  const int N = 400;
  const unsigned long long WR = 2000, WC = 400;  // 20 us, 4 us
  const size_t LDS = 100 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(R), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
  hipStream_t s[2];
  CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  unsigned* flag;
  CK(hipMalloc((void**)&flag, 64));
  std::vector<hipEvent_t> ev(2 * N + 2);
  for (auto& e : ev)
    CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  float ms;
  for (int rep = 0; rep < 2; rep++)
  {
    // (1) one stream
    CK(hipMemset(flag, 0, 64));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, s[0]));
    for (int i = 0; i < N; i++)
    {
      hipLaunchKernelGGL(R, dim3(256), dim3(256), LDS, s[0], (const unsigned*)nullptr, 0u, WR, (float*)nullptr);
      hipLaunchKernelGGL(C, dim3(2), dim3(256), 0, s[0], (unsigned*)nullptr, 0u, WC);
    }
    CK(hipEventRecord(t1, s[0]));
    CK(hipEventSynchronize(t1));
    CK(hipEventElapsedTime(&ms, t0, t1));
    printf("(1) one stream, in-order          : %.2f us per iteration (kernels alone: %.1f)\n", ms * 1e3 / N, 24.0);
    // (2) two streams, every kernel behind an event on its predecessor
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, s[0]));
    CK(hipEventRecord(ev[2 * N], s[0]));
    hipEvent_t last = ev[2 * N];
    for (int i = 0; i < N; i++)
    {
      CK(hipStreamWaitEvent(s[0], last, 0));
      hipLaunchKernelGGL(R, dim3(256), dim3(256), LDS, s[0], (const unsigned*)nullptr, 0u, WR, (float*)nullptr);
      CK(hipEventRecord(ev[2 * i], s[0]));
      CK(hipStreamWaitEvent(s[1], ev[2 * i], 0));
      hipLaunchKernelGGL(C, dim3(2), dim3(256), 0, s[1], (unsigned*)nullptr, 0u, WC);
      CK(hipEventRecord(ev[2 * i + 1], s[1]));
      last = ev[2 * i + 1];
    }
    CK(hipStreamWaitEvent(s[0], last, 0));
    CK(hipEventRecord(t1, s[0]));
    CK(hipEventSynchronize(t1));
    CK(hipEventElapsedTime(&ms, t0, t1));
    printf("(2) two streams, events everywhere: %.2f us per iteration\n", ms * 1e3 / N);
    // (3) overlap
    CK(hipMemset(flag, 0, 64));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, s[0]));
    CK(hipEventRecord(ev[2 * N], s[0]));
    CK(hipStreamWaitEvent(s[1], ev[2 * N], 0));
    last = ev[2 * N];
    for (int i = 0; i < N; i++)
    {
      hipStream_t st = s[i & 1];
      CK(hipStreamWaitEvent(st, last, 0));  // R_i starts when R_{i-1} is done (and, in order, after C_{i-2})
      hipLaunchKernelGGL(R, dim3(256), dim3(256), LDS, st, (const unsigned*)flag, (unsigned)i, WR, (float*)nullptr);  // spins for C_{i-1}
      CK(hipEventRecord(ev[2 * i], st));
      last = ev[2 * i];
      hipLaunchKernelGGL(C, dim3(2), dim3(256), 0, st, flag, (unsigned)(i + 1), WC);
    }
    CK(hipEventRecord(ev[2 * N + 1], s[(N - 1) & 1]));
    CK(hipStreamWaitEvent(s[0], ev[2 * N + 1], 0));
    CK(hipEventRecord(t1, s[0]));
    CK(hipEventSynchronize(t1));
    CK(hipEventElapsedTime(&ms, t0, t1));
    unsigned f = 0;
    CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
    printf("(3) overlap R_{i+1} with C_i      : %.2f us per iteration (flag %u of %d)\n", ms * 1e3 / N, f, N);
  }
  return 0;
}
