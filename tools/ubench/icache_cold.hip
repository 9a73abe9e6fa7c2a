// Is the front of a launch bound by INSTRUCTION FETCH?  The Cartpole rollout kernel spends ~3 us before the first sample of a launch
// exists, on code that runs once per launch (prologue + the sampler waves' first trip: ~1500 instructions of straight-line code).
// If the instruction cache started cold at every launch, that code would run at the speed of its misses.  Here: a straight-line
// body of N fused multiply-adds with distinct literal constants (the compiler emits a v_mov + a v_fmamk per FMA: 2 N instructions,
// 8 bytes each, four independent chains), executed three times per launch by wave 0 of every block; s_memtime around each pass.
// Pass 1 = cold (if the cache is cold at launch), passes 2 / 3 = warm.  1000 back-to-back launches of the same kernel, like the bench.
// RESULT (MI355X): pass 1 = pass 2 = pass 3 at every size (16 KB of code: 15468 / 15480 / 15477 ticks) — no cold start between
// back-to-back launches of one kernel; 8.0 cycles per FMA = 4.0 per instruction, the issue rate of a lone wave64 — and
// tools/ubench/ifetch_rate.hip: 4-byte and 8-byte encodings issue at the same 4.04 cycles, alone or with three other waves
// fetching beside them.  The front of the launch is not a fetch problem.
//   hipcc --offload-arch=gfx950 -O3 -o icache_cold tools/ubench/icache_cold.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N>
__device__ inline void body(float& a, float& b, float& c, float& d)
{
#pragma unroll
  for (int i = 0; i < N; i += 4)
  {
    a = fmaf(a, 1.0001f + 1e-6f * i, 0.5f + 1e-5f * i);
    b = fmaf(b, 1.0002f + 1e-6f * i, 0.25f + 1e-5f * i);
    c = fmaf(c, 1.0003f + 1e-6f * i, 0.125f + 1e-5f * i);
    d = fmaf(d, 1.0004f + 1e-6f * i, 0.0625f + 1e-5f * i);
  }
}

template <int N>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* stamps, int passes, float x0)
{
  float a = x0, b = x0 + 1.0f, c = x0 + 2.0f, d = x0 + 3.0f;
  const int wave = threadIdx.x >> 6;
  if (wave == 0)
  {
#pragma nounroll
    for (int p = 0; p < passes; p++)
    {
      // (the stamps take the four chains as operands: the clock reads stay where they are written)
      unsigned long long t0, t1;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
      body<N>(a, b, c, d);
      asm volatile("s_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
      if (threadIdx.x == 0 && p < 4)
        stamps[4 * blockIdx.x + p] = t1 - t0;
    }
  }
  if (a + b + c + d == 1234.5f)
    out[threadIdx.x] = a;
}

template <int N>
void run(float* out, unsigned long long* stamps, hipStream_t s)
{
  for (int pass = 0; pass < 2; pass++)
  {
    for (int i = 0; i < 1000; i++)
      hipLaunchKernelGGL(k<N>, dim3(256), dim3(256), 0, s, out, stamps, 3, 1.0f + i);
    hipStreamSynchronize(s);
    std::vector<unsigned long long> h(4 * 256);
    hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double m[3] = { 0, 0, 0 };
    for (int b = 0; b < 256; b++)
      for (int p = 0; p < 3; p++)
        m[p] += (double)h[4 * b + p] / 256.0;
    printf("N = %5d instructions (%5.1f KB): pass 1 %7.0f ticks, pass 2 %7.0f, pass 3 %7.0f  (ticks per instruction: %.2f / %.2f / %.2f)\n", N,
           N * 8 / 1024.0, m[0], m[1], m[2], m[0] / N, m[1] / N, m[2] / N);
  }
}

int main()
{
  float* out;
  unsigned long long* stamps;
  hipMalloc(&out, 256 * sizeof(float));
  hipMalloc(&stamps, 4 * 256 * sizeof(unsigned long long));
  hipStream_t s;
  hipStreamCreate(&s);
  run<128>(out, stamps, s);
  run<512>(out, stamps, s);
  run<1024>(out, stamps, s);
  run<2048>(out, stamps, s);
  return 0;
}
