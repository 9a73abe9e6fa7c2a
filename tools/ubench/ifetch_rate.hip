// Instruction FETCH rate of one wave on gfx950, by encoding size (the companion of tools/ubench/icache_cold.hip): does a lone wave
// per SIMD — the rollout kernels' dynamics waves — pay for 8-byte encodings (VOP3, VOP3P: v_fma_f32, v_pk_fma_f32) what it does
// not pay for 4-byte ones?  .rept blocks of inline asm, four independent accumulators: E32 = v_fmac_f32_e32 (4 bytes), E64 =
// v_fma_f32 (VOP3, 8 bytes), MIX = alternating; with 1, 2 or 4 waves of the block (one per SIMD) running the body at the same time
// — the role waves of a block share the CU's instruction cache port.
// RESULT (MI355X, 2048 instructions = 8 / 16 KB, third pass): E32 4.04, MIX 4.04, E64 4.04-4.65 cycles per instruction with one
// wave, 4.04-4.05 with two or four: the encoding size does not matter, neither do the neighbours.
//   hipcc --offload-arch=gfx950 -O3 -o ifetch_rate tools/ubench/ifetch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define E32_4 "v_fmac_f32_e32 %0, %4, %5\n v_fmac_f32_e32 %1, %4, %5\n v_fmac_f32_e32 %2, %4, %5\n v_fmac_f32_e32 %3, %4, %5\n"
#define E64_4 "v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n"
#define MIX_4 "v_fmac_f32_e32 %0, %4, %5\n v_fma_f32 %1, %4, %5, %1\n v_fmac_f32_e32 %2, %4, %5\n v_fma_f32 %3, %4, %5, %3\n"

template <int KIND, int REPT>
__device__ inline void body(float& a, float& b, float& c, float& d, float x, float y)
{
  static_assert(REPT == 64 || REPT == 512, "");
  if (KIND == 0 && REPT == 64)
    asm volatile(".rept 64\n" E32_4 ".endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
  if (KIND == 1 && REPT == 64)
    asm volatile(".rept 64\n" E64_4 ".endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
  if (KIND == 2 && REPT == 64)
    asm volatile(".rept 64\n" MIX_4 ".endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
  if (KIND == 0 && REPT == 512)
    asm volatile(".rept 512\n" E32_4 ".endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
  if (KIND == 1 && REPT == 512)
    asm volatile(".rept 512\n" E64_4 ".endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
  if (KIND == 2 && REPT == 512)
    asm volatile(".rept 512\n" MIX_4 ".endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
}

template <int KIND, int REPT>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* stamps, int active_waves, float x0)
{
  float a = x0, b = x0 + 1.0f, c = x0 + 2.0f, d = x0 + 3.0f;
  const float x = 1.0f + 1e-7f * x0, y = 1e-3f;
  const int wave = threadIdx.x >> 6;
  if (wave < active_waves)
  {
#pragma nounroll
    for (int p = 0; p < 3; p++)
    {
      unsigned long long t0, t1;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
      body<KIND, REPT>(a, b, c, d, x, y);
      asm volatile("s_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
      if (threadIdx.x == 0)
        stamps[4 * blockIdx.x + p] = t1 - t0;
    }
  }
  if (a + b + c + d == 1234.5f)
    out[threadIdx.x] = a;
}

template <int KIND, int REPT>
void run(const char* name, float* out, unsigned long long* stamps, hipStream_t s)
{
  for (int waves = 1; waves <= 4; waves *= 2)
  {
    for (int i = 0; i < 200; i++)
      hipLaunchKernelGGL((k<KIND, REPT>), dim3(256), dim3(256), 0, s, out, stamps, waves, 1.0f + i);
    hipStreamSynchronize(s);
    std::vector<unsigned long long> h(4 * 256);
    hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double m[3] = { 0, 0, 0 };
    for (int b = 0; b < 256; b++)
      for (int p = 0; p < 3; p++)
        m[p] += (double)h[4 * b + p] / 256.0;
    const int n = 4 * REPT;
    printf("%-4s %5d instructions, %d wave(s) per CU: %7.0f / %7.0f / %7.0f ticks per pass = %.2f cycles per instruction (pass 3)\n", name, n,
           waves, m[0], m[1], m[2], m[2] / n);
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
  run<0, 64>("E32", out, stamps, s);
  run<1, 64>("E64", out, stamps, s);
  run<2, 64>("MIX", out, stamps, s);
  run<0, 512>("E32", out, stamps, s);
  run<1, 512>("E64", out, stamps, s);
  run<2, 512>("MIX", out, stamps, s);
  return 0;
}
