// Does a wave64 VALU instruction cost fewer issue cycles when only the low 32 / 16 lanes are active (EXEC narrowed)?  gfx950
// executes a wave64 fp32 instruction as four passes over a SIMD16; if passes whose lanes are all inactive were skipped, a
// latency-bound rollout (one dynamics wave per block: T steps x ~83 dependent-or-not instructions x ~2 ns) could be split over
// narrower waves.  One wave, NCHAIN interleaved fma chains, lanes >= W branch around the loop.
// Build: hipcc --offload-arch=gfx950 -O3 exec_width.hip -o exec_width ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCHAIN>
__global__ void chains(float* out, int iters, float a, float b, int width)
{
  float v[NCHAIN];
#pragma unroll
  for (int c = 0; c < NCHAIN; c++)
    v[c] = threadIdx.x * 0.001f + c;
  long long t0 = 0, t1 = 0;
  if ((int)threadIdx.x < width)
  {
    t0 = clock64();
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int c = 0; c < NCHAIN; c++)
          v[c] = __builtin_fmaf(v[c], a, b);
    }
    t1 = clock64();
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < NCHAIN; c++)
    s += v[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0)
    ((long long*)(out + 64))[0] = t1 - t0;
}
template <int NCHAIN>
void run(int width)
{
  float* d;
  (void)hipMalloc(&d, 1024);
  const int iters = 4000;
  chains<NCHAIN><<<1, 64>>>(d, iters, 0.999f, 0.001f, width);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  chains<NCHAIN><<<1, 64>>>(d, iters, 0.999f, 0.001f, width);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long cyc;
  (void)hipMemcpy(&cyc, d + 64, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * NCHAIN;
  printf("active lanes %2d, %d chains: %.3f ns per instruction (wall), %.3f clock64 ticks per instruction\n", width, NCHAIN,
         ms * 1e6 / n, (double)cyc / n);
  (void)hipFree(d);
}
int main()
{
  for (int w : { 64, 48, 32, 16, 1 })
  {
    run<1>(w);
    run<4>(w);
  }
  return 0;
}
