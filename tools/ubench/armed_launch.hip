// What would an "armed" mppi_compute_control save?  The rollout launch of a call costs the host ~3.4 us of
// enqueue and the device ~3 us from doorbell to first wave before a single rollout step runs.  Armed: the launch is enqueued
// BEFORE the call (behind the previous call's control phase), its 256 blocks — one per CU, 100 KB of LDS each, like the Cartpole
// rollout kernel — are resident and poll a go word in a fine-grained device allocation the host writes through the PCIe BAR; the
// call itself is {write 0.5 KB of inputs, fence, write go}.  Measured here, per call: host t0 -> every block has seen go and read
// the inputs -> flag back on the host, against the same kernel launched the ordinary way at t0.
// hipcc --offload-arch=gfx950 -O3 -o armed_launch tools/ubench/armed_launch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

constexpr int BLOCKS = 256, THREADS = 256, LDS_BYTES = 100 * 1024, N_IN = 128;

// armed != 0: wait for inbox[N_IN] (the go word) == seq first.  Then every block reads the inputs, takes a ticket; the last one
// hands the sum back and raises the host flag.
__global__ void __launch_bounds__(THREADS) rolloutStandIn(const float* inbox, const unsigned* go, int armed, unsigned seq,
                                                          unsigned* ticket, float* out, unsigned* flag, unsigned* relay)
{
  extern __shared__ float lds[];
  if (armed)
  {
    // armed == 1: every wave polls; 2: one wave per block polls, the others wait at the barrier; 3: as 2, block 0 alone polls the
    // host's word and passes it on through an ordinary device word (relay) the other blocks poll
    __shared__ int expired;
    if (threadIdx.x == 0)
      expired = 0;
    __syncthreads();
    if (armed == 1 || threadIdx.x < 64)
    {
      const unsigned long long t0 = wall_clock64();
      const unsigned* word = (armed == 3 && blockIdx.x != 0) ? relay : go;
      while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq)
      {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 100000000ull)  // 1 s: nobody called
        {
          expired = 1;
          break;
        }
      }
      if (armed == 3 && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(relay, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (expired)
      return;
  }
  float s = 0.0f;
  for (int i = threadIdx.x; i < N_IN; i += THREADS)
    s += inbox[i];
  lds[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    float t = 0.0f;
    for (int i = 0; i < THREADS; i++)
      t += lds[i];
    __threadfence();
    if (atomicAdd(ticket, 1u) == (unsigned)BLOCKS * seq - 1u)
    {
      *out = t;
      __threadfence_system();
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

int main()
{
  int large_bar = 0;
  hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
  float* inbox = nullptr;
  if (!large_bar || hipExtMallocWithFlags((void**)&inbox, (N_IN + 16) * 4, hipDeviceMallocFinegrained) != hipSuccess)
  {
    printf("no large BAR / fine-grained allocation: nothing to measure\n");
    return 0;
  }
  unsigned* go = reinterpret_cast<unsigned*>(inbox + N_IN);
  unsigned *flag_h, *flag_dev, *ticket;
  float *out_h, *out_dev;
  hipHostMalloc((void**)&flag_h, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostGetDevicePointer((void**)&flag_dev, flag_h, 0);
  hipHostMalloc((void**)&out_h, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostGetDevicePointer((void**)&out_dev, out_h, 0);
  hipMalloc((void**)&ticket, 8);
  hipMemset(ticket, 0, 8);
  unsigned* relay = ticket + 1;
  *flag_h = 0;
  *go = 0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(rolloutStandIn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  unsigned seq = 0;
  for (int armed = 0; armed < 4; armed++)
  {
    double total = 0, best = 1e9;
    int bad = 0;
    for (int it = 0; it < 1200; it++)
    {
      hipStreamSynchronize(st);
      ++seq;
      if (armed)
      {
        hipLaunchKernelGGL(rolloutStandIn, dim3(BLOCKS), dim3(THREADS), LDS_BYTES, st, inbox, go, armed, seq, ticket, out_dev, flag_dev, relay);
        const auto w = std::chrono::steady_clock::now();  // the plant's time between two calls: the blocks become resident
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w).count() < 25.0)
          ;
      }
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N_IN; i++)
        inbox[i] = (float)(it + i);
#if defined(__x86_64__)
      _mm_sfence();
#endif
      if (armed)
      {
        __atomic_store_n(go, seq, __ATOMIC_RELEASE);
#if defined(__x86_64__)
        _mm_sfence();
#endif
      }
      else
        hipLaunchKernelGGL(rolloutStandIn, dim3(BLOCKS), dim3(THREADS), LDS_BYTES, st, inbox, go, 0, seq, ticket, out_dev, flag_dev, relay);
      while (__atomic_load_n(flag_h, __ATOMIC_ACQUIRE) != seq)
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0)
        {
          printf("no flag after 2 s (armed %d, call %d)\n", armed, it);
          return 1;
        }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (*out_h != (float)N_IN * it + (float)(N_IN * (N_IN - 1) / 2))
        bad++;
      if (it >= 200)
      {
        total += us;
        best = us < best ? us : best;
      }
    }
    printf("%-28s inputs written -> all %d blocks ran -> flag back: mean %.2f us, best %.2f us, wrong sums %d\n",
           armed == 0 ? "ordinary launch" : armed == 1 ? "armed, every wave polls" : armed == 2 ? "armed, one wave per block" :
                                                                                        "armed, block 0 + relay word",
           BLOCKS, total / 1000, best, bad);
  }
  return 0;
}
