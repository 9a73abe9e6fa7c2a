// Can a lone wave issue independent VALU instructions while its MFMA executes?  (gfx950, v_mfma_f32_16x16x4_f32, 8 passes)
// Per loop iteration: 8 x { one MFMA into accumulator chain A (or alternating A/B), then N independent v_fma_f32 }, N = 0..12.
// Prints ns per {MFMA + N fma} group for one wave alone on its SIMD, and for 2 waves of one block on the SAME SIMD (waves 0, 4).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int N, bool TWO_CHAINS, bool DEP_VALU>
__global__ void k(float* out, int iters, float a, float b)
{
  f4 c0 = { 0, 0, 0, 0 }, c1 = { 0, 0, 0, 0 };
  float x[12];
  for (int i = 0; i < 12; i++)
    x[i] = threadIdx.x * 0.001f + i;
  float wa = a + threadIdx.x, wb = b;
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int g = 0; g < 8; g++)
    {
      if (TWO_CHAINS && (g & 1))
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(wa), "v"(wb));
      else
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(wa), "v"(wb));
#pragma unroll
      for (int j = 0; j < N; j++)
      {
        if (DEP_VALU)  // one dependent chain
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
        else
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
      }
    }
  }
  float s = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
  for (int i = 0; i < 12; i++)
    s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, bool TWO, bool DEP>
float run(int threads, float* d)
{
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<N, TWO, DEP><<<1, threads>>>(d, 100, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<N, TWO, DEP><<<1, threads>>>(d, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (iters * 8.0f);  // ns per group
}

template <int N>
void row(float* d)
{
  printf("N=%2d  1 wave: one chain %6.1f ns  two chains %6.1f ns  (dep. valu chain %6.1f)   | 5 waves (2 on SIMD0): %6.1f ns\n", N,
         run<N, false, false>(64, d), run<N, true, false>(64, d), run<N, true, true>(64, d), run<N, true, false>(320, d));
}

int main()
{
  float* d;
  hipMalloc(&d, 4096 * 4);
  row<0>(d);
  row<1>(d);
  row<2>(d);
  row<4>(d);
  row<6>(d);
  row<7>(d);
  row<8>(d);
  row<10>(d);
  row<12>(d);
  return 0;
}
