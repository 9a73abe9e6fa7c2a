// Does the kernel-argument PRELOAD of gfx950 (the command processor writes the first dwords of the kernarg segment into user SGPRs
// at wave launch: .amdhsa_user_sgpr_kernarg_preload_length) shorten the front of a launch whose first memory operation hangs off a
// pointer in the arguments — the streamed merge's record loads of the Cartpole rollout kernel?  Ordinarily the
// wave's first act is s_load(kernarg) + s_waitcnt: one memory round trip before the first useful load can even be issued.
// Two builds of the same file, 256 blocks x 256 threads like the rollout kernel, 2000 back-to-back launches:
//   hipcc --offload-arch=gfx950 -O3 -o kp_base tools/ubench/kernarg_preload.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 -o kp_pre tools/ubench/kernarg_preload.hip
// Prints: us per launch (HIP events) and, from s_memtime inside the kernel, entry -> first dependent load returned (mean over blocks;
// the "100 MHz" figure in the output line is wrong: s_memtime counts core clocks, ~2.4 GHz).
// RESULT (MI355X): 723 ticks without, 648 with the preload = 0.30 -> 0.27 us — the argument block's s_load is a ~35 ns matter (the
// command processor has the segment in the scalar cache's reach before the first wave runs), not the 0.9 us round trip the
// in-kernel timers of the instrumented rollout kernel suggested.  Tried on rolloutPipelineKernel all the same (copies of the
// three scalars in front of the plugin objects, the record loads hoisted above every use of the argument block, cartpole.hip
// compiled with -amdgpu-kernarg-preload-count=4): 24.25 us per iteration against 23.75 — the hoisted loads had to be issued by
// all four waves (a branch on the role ends the entry block) and doubled the cache-line requests of the launch's front.  Not kept.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Plugin
{
  float w[96];  // the plugin objects behind the scalars, like the rollout kernels' (not preloaded)
};

__global__ void __launch_bounds__(256) standIn(const float* records, int n, float scale, unsigned long long* stamps, float* out,
                                               Plugin plugin)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int lane = threadIdx.x;
  float v = records[(lane * 104) % n];  // first useful load: address from the arguments
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  v = v * scale + plugin.w[17];
  if (lane == 0)
    stamps[blockIdx.x] = t1 - t0;
  if (v == 12345.678f)
    out[lane] = v;
}

int main()
{
  const int N = 256 * 104, LAUNCHES = 2000;
  float *records, *out;
  unsigned long long* stamps;
  hipMalloc(&records, N * sizeof(float));
  hipMalloc(&out, 256 * sizeof(float));
  hipMalloc(&stamps, 256 * sizeof(unsigned long long));
  hipMemset(records, 0, N * sizeof(float));
  Plugin plugin{};
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int pass = 0; pass < 4; pass++)
  {
    hipEventRecord(e0, s);
    for (int i = 0; i < LAUNCHES; i++)
      hipLaunchKernelGGL(standIn, dim3(256), dim3(256), 0, s, records, N, 1.0f + i, stamps, out, plugin);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), stamps, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto t : h)
      mean += (double)t;
    mean /= 256.0;
    printf("pass %d: %.3f us per launch; entry -> first dependent load back: %.0f ticks (100 MHz: %.2f us)\n", pass,
           1000.0 * ms / LAUNCHES, mean, mean / 100.0);
  }
  return 0;
}
