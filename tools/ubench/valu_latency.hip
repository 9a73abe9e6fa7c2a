// Dependent-issue latency of fp32 VALU instructions on one wave (gfx950): NCHAIN independent fma chains interleaved.
// Build: hipcc --offload-arch=gfx950 -O3 valu_latency.hip -o valu_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCHAIN>
__global__ void chains(float* out, int iters, float a, float b)
{
  float v[NCHAIN];
#pragma unroll
  for (int c = 0; c < NCHAIN; c++)
    v[c] = threadIdx.x * 0.001f + c;
  long long t0 = clock64();
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int c = 0; c < NCHAIN; c++)
        v[c] = __builtin_fmaf(v[c], a, b);
  }
  long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int c = 0; c < NCHAIN; c++)
    s += v[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0)
    ((long long*)(out + 64))[0] = t1 - t0;
}
template <int NCHAIN>
void run(const char* name)
{
  float* d;
  hipMalloc(&d, 1024);
  const int iters = 2000;
  chains<NCHAIN><<<1, 64>>>(d, iters, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  chains<NCHAIN><<<1, 64>>>(d, iters, 0.999f, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long cyc;
  hipMemcpy(&cyc, d + 64, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * NCHAIN;
  printf("%s: %d chains: %.2f ns per instruction (wall), %.2f clock64 ticks per instruction, %.1f ns per dependent step\n", name,
         NCHAIN, ms * 1e6 / n, (double)cyc / n, ms * 1e6 / (iters * 16.0));
}
int main()
{
  run<1>("fma");
  run<2>("fma");
  run<3>("fma");
  run<4>("fma");
  run<8>("fma");
  return 0;
}
