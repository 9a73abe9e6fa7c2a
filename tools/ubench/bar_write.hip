// Can the host write straight into device memory (large BAR) instead of handing inputs over through mapped HOST memory + an
// ingest kernel?  hipExtMallocWithFlags(hipDeviceMallocFinegrained / Uncached): does the pointer accept CPU stores, does a kernel
// launched right after see them, and what does {write 2 KB + launch + flag back} cost against {mapped host memory + launch}?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <csetjmp>
#include <csignal>

static sigjmp_buf jb;
static void onsegv(int) { siglongjmp(jb, 1); }

__global__ void consume(const float* in, int n, float* sum_out, unsigned* flag, unsigned seq)
{
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 64)
    s += in[i];
  for (int o = 32; o > 0; o >>= 1)
    s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0)
  {
    *sum_out = s;
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int main()
{
  const int n = 512;
  float *host_mapped, *host_mapped_dev, *out_h, *out_dev;
  unsigned *flag_h, *flag_dev;
  hipHostMalloc((void**)&host_mapped, n * 4, hipHostMallocMapped);
  hipHostGetDevicePointer((void**)&host_mapped_dev, host_mapped, 0);
  hipHostMalloc((void**)&out_h, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostGetDevicePointer((void**)&out_dev, out_h, 0);
  hipHostMalloc((void**)&flag_h, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostGetDevicePointer((void**)&flag_dev, flag_h, 0);
  *flag_h = 0;
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const unsigned flags[2] = { hipDeviceMallocFinegrained, hipDeviceMallocUncached };
  const char* names[2] = { "hipDeviceMallocFinegrained", "hipDeviceMallocUncached" };
  float* vram[2] = { nullptr, nullptr };
  bool ok[2] = { false, false };
  for (int k = 0; k < 2; k++)
  {
    if (hipExtMallocWithFlags((void**)&vram[k], n * 4, flags[k]) != hipSuccess)
    {
      printf("%s: allocation refused\n", names[k]);
      (void)hipGetLastError();
      continue;
    }
    signal(SIGSEGV, onsegv);
    signal(SIGBUS, onsegv);
    if (sigsetjmp(jb, 1) == 0)
    {
      volatile float* p = vram[k];
      p[0] = 1.0f;  // CPU store to the device allocation
      ok[k] = true;
    }
    signal(SIGSEGV, SIG_DFL);
    signal(SIGBUS, SIG_DFL);
    printf("%s: CPU store %s\n", names[k], ok[k] ? "accepted" : "faulted (not host-accessible)");
  }
  unsigned seq = 0;
  auto run = [&](const char* what, float* host_ptr, const float* dev_ptr) {
    double best = 1e9, total = 0;
    int bad = 0;
    for (int it = 0; it < 2200; it++)
    {
      hipStreamSynchronize(st);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < n; i++)
        host_ptr[i] = (float)(it + i);
      ++seq;
      hipLaunchKernelGGL(consume, dim3(1), dim3(64), 0, st, dev_ptr, n, out_dev, flag_dev, seq);
      while (__atomic_load_n(flag_h, __ATOMIC_ACQUIRE) != seq)
        ;
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      const float want = (float)n * it + (float)(n * (n - 1) / 2);
      if (*out_h != want)
        bad++;
      if (it >= 200)
      {
        total += us;
        best = us < best ? us : best;
      }
    }
    printf("%-34s write 2 KB + launch + flag back: mean %.2f us, best %.2f us, wrong sums %d\n", what, total / 2000, best, bad);
  };
  run("mapped host memory", host_mapped, host_mapped_dev);
  for (int k = 0; k < 2; k++)
    if (ok[k])
      run(names[k], vram[k], vram[k]);
  return 0;
}
