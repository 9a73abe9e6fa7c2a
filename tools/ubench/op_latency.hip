// Per-opcode dependent latency and independent issue interval of one wave alone on its SIMD (gfx950), measured with
// inline asm so the compiler cannot reorder or fold anything.  For each op: a chain of 8 x 64 dependent instances
// (latency) and 8 interleaved independent chains (issue interval).  Times from s_memtime (constant 100 MHz) are converted
// with the wall clock of a known fma chain; printed in ns and in "fma-latency units".
// Build: hipcc --offload-arch=gfx950 -O3 op_latency.hip -o op_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// DEP: one register chain.  IND: 8 registers round robin.
#define KERNEL_PAIR(NAME, DEP_ASM, IND_ASM)                                                                          \
  __global__ void NAME##_dep(float* out, int iters, float a, float b)                                                \
  {                                                                                                                    \
    float x0 = threadIdx.x * 0.001f + 1.0f;                                                                            \
    float x1 = x0, x2 = x0, x3 = x0, x4 = x0, x5 = x0, x6 = x0, x7 = x0;                                               \
    for (int i = 0; i < iters; i++)                                                                                    \
    {                                                                                                                  \
      asm volatile(REP64(DEP_ASM) : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "vcc");                                     \
    }                                                                                                                  \
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                                          \
  }                                                                                                                    \
  __global__ void NAME##_ind(float* out, int iters, float a, float b)                                                \
  {                                                                                                                    \
    float x0 = threadIdx.x * 0.001f + 1.0f;                                                                            \
    float x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;                   \
    for (int i = 0; i < iters; i++)                                                                                    \
    {                                                                                                                  \
      asm volatile(REP8(IND_ASM)                                                                                       \
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)                    \
                   : "v"(a), "v"(b)                                                                                    \
                   : "vcc");                                                                                           \
    }                                                                                                                  \
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                                          \
  }

// operand numbering: DEP: %0 = x0, %1 = x1, %2 = a, %3 = b.  IND: %0..%7 = x0..x7, %8 = a, %9 = b.
#define IND8(op_fmt_0, op_fmt_1, op_fmt_2, op_fmt_3, op_fmt_4, op_fmt_5, op_fmt_6, op_fmt_7)                            \
  op_fmt_0 op_fmt_1 op_fmt_2 op_fmt_3 op_fmt_4 op_fmt_5 op_fmt_6 op_fmt_7

KERNEL_PAIR(fma, "v_fma_f32 %0, %0, %2, %3\n",
            IND8("v_fma_f32 %0, %0, %8, %9\n", "v_fma_f32 %1, %1, %8, %9\n", "v_fma_f32 %2, %2, %8, %9\n",
                 "v_fma_f32 %3, %3, %8, %9\n", "v_fma_f32 %4, %4, %8, %9\n", "v_fma_f32 %5, %5, %8, %9\n",
                 "v_fma_f32 %6, %6, %8, %9\n", "v_fma_f32 %7, %7, %8, %9\n"))
KERNEL_PAIR(mul, "v_mul_f32 %0, %0, %2\n",
            IND8("v_mul_f32 %0, %0, %8\n", "v_mul_f32 %1, %1, %8\n", "v_mul_f32 %2, %2, %8\n", "v_mul_f32 %3, %3, %8\n",
                 "v_mul_f32 %4, %4, %8\n", "v_mul_f32 %5, %5, %8\n", "v_mul_f32 %6, %6, %8\n", "v_mul_f32 %7, %7, %8\n"))
KERNEL_PAIR(add_lit, "v_add_f32 %0, 0x40490fdb, %0\n",
            IND8("v_add_f32 %0, 0x40490fdb, %0\n", "v_add_f32 %1, 0x40490fdb, %1\n", "v_add_f32 %2, 0x40490fdb, %2\n",
                 "v_add_f32 %3, 0x40490fdb, %3\n", "v_add_f32 %4, 0x40490fdb, %4\n", "v_add_f32 %5, 0x40490fdb, %5\n",
                 "v_add_f32 %6, 0x40490fdb, %6\n", "v_add_f32 %7, 0x40490fdb, %7\n"))
KERNEL_PAIR(fmac_lit, "v_fmac_f32 %0, 0x3f7fbe77, %0\n",
            IND8("v_fmac_f32 %0, 0x3f7fbe77, %0\n", "v_fmac_f32 %1, 0x3f7fbe77, %1\n", "v_fmac_f32 %2, 0x3f7fbe77, %2\n",
                 "v_fmac_f32 %3, 0x3f7fbe77, %3\n", "v_fmac_f32 %4, 0x3f7fbe77, %4\n", "v_fmac_f32 %5, 0x3f7fbe77, %5\n",
                 "v_fmac_f32 %6, 0x3f7fbe77, %6\n", "v_fmac_f32 %7, 0x3f7fbe77, %7\n"))
KERNEL_PAIR(trunc, "v_trunc_f32 %0, %0\n",
            IND8("v_trunc_f32 %0, %0\n", "v_trunc_f32 %1, %1\n", "v_trunc_f32 %2, %2\n", "v_trunc_f32 %3, %3\n",
                 "v_trunc_f32 %4, %4\n", "v_trunc_f32 %5, %5\n", "v_trunc_f32 %6, %6\n", "v_trunc_f32 %7, %7\n"))
KERNEL_PAIR(rcp, "v_rcp_f32 %0, %0\n",
            IND8("v_rcp_f32 %0, %0\n", "v_rcp_f32 %1, %1\n", "v_rcp_f32 %2, %2\n", "v_rcp_f32 %3, %3\n",
                 "v_rcp_f32 %4, %4\n", "v_rcp_f32 %5, %5\n", "v_rcp_f32 %6, %6\n", "v_rcp_f32 %7, %7\n"))
KERNEL_PAIR(bfi, "v_bfi_b32 %0, %2, %0, %3\n",
            IND8("v_bfi_b32 %0, %8, %0, %9\n", "v_bfi_b32 %1, %8, %1, %9\n", "v_bfi_b32 %2, %8, %2, %9\n",
                 "v_bfi_b32 %3, %8, %3, %9\n", "v_bfi_b32 %4, %8, %4, %9\n", "v_bfi_b32 %5, %8, %5, %9\n",
                 "v_bfi_b32 %6, %8, %6, %9\n", "v_bfi_b32 %7, %8, %7, %9\n"))
// compare -> vcc -> select: the pair as the compiler emits it (with the s_nop it places in between)
KERNEL_PAIR(cmp_sel, "v_cmp_lt_f32 vcc, %0, %2\ns_nop 1\nv_cndmask_b32 %0, %0, %3, vcc\n",
            IND8("v_cmp_lt_f32 vcc, %0, %8\ns_nop 1\nv_cndmask_b32 %0, %0, %9, vcc\n",
                 "v_cmp_lt_f32 vcc, %1, %8\ns_nop 1\nv_cndmask_b32 %1, %1, %9, vcc\n",
                 "v_cmp_lt_f32 vcc, %2, %8\ns_nop 1\nv_cndmask_b32 %2, %2, %9, vcc\n",
                 "v_cmp_lt_f32 vcc, %3, %8\ns_nop 1\nv_cndmask_b32 %3, %3, %9, vcc\n",
                 "v_cmp_lt_f32 vcc, %4, %8\ns_nop 1\nv_cndmask_b32 %4, %4, %9, vcc\n",
                 "v_cmp_lt_f32 vcc, %5, %8\ns_nop 1\nv_cndmask_b32 %5, %5, %9, vcc\n",
                 "v_cmp_lt_f32 vcc, %6, %8\ns_nop 1\nv_cndmask_b32 %6, %6, %9, vcc\n",
                 "v_cmp_lt_f32 vcc, %7, %8\ns_nop 1\nv_cndmask_b32 %7, %7, %9, vcc\n"))
// packed fp32 fma on a register pair (x0:x1 must be consecutive: use one 64-bit operand instead)
__global__ void pkfma_dep(float* out, int iters, float a, float b)
{
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x = { threadIdx.x * 0.001f + 1.0f, 2.0f }, aa = { a, a }, bb = { b, b };
  for (int i = 0; i < iters; i++)
    asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(x) : "v"(aa), "v"(bb));
  out[threadIdx.x] = x.x + x.y;
}
__global__ void pkfma_ind(float* out, int iters, float a, float b)
{
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x0 = { threadIdx.x * 0.001f + 1.0f, 2.0f }, aa = { a, a }, bb = { b, b };
  f2 x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; i++)
    asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\n"
                      "v_pk_fma_f32 %3, %3, %8, %9\nv_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\n"
                      "v_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                 : "v"(aa), "v"(bb));
  const f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[threadIdx.x] = s.x + s.y;
}
// SALU op between VALU ops, LDS read latency
__global__ void lds_dep(float* out, int iters, float a, float b)
{
  __shared__ int idx[64];
  idx[threadIdx.x] = (threadIdx.x * 4) & 255;
  __syncthreads();
  int p = threadIdx.x * 4;
  for (int i = 0; i < iters; i++)
    asm volatile(REP64("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(p)::"memory");
  out[threadIdx.x] = (float)p;
}

typedef void (*kern_t)(float*, int, float, float);
static double timeKernel(kern_t k, int iters)
{
  float* d;
  hipMalloc(&d, 4096);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, iters, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double best = 1e30;
  for (int r = 0; r < 3; r++)
  {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, iters, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best)
      best = ms;
  }
  hipFree(d);
  return best * 1e6;  // ns
}
static void report(const char* name, kern_t dep, kern_t ind, int per_iter_dep, int per_iter_ind)
{
  const int iters = 4000;
  const double base = timeKernel(dep, 0);  // launch overhead
  const double td = (timeKernel(dep, iters) - base) / ((double)iters * per_iter_dep);
  double ti = 0;
  if (ind)
    ti = (timeKernel(ind, iters) - base) / ((double)iters * per_iter_ind);
  printf("%-10s dependent %.2f ns/op   independent %.2f ns/op\n", name, td, ti);
}
int main()
{
  report("fma", fma_dep, fma_ind, 64, 64);
  report("mul", mul_dep, mul_ind, 64, 64);
  report("add_lit", add_lit_dep, add_lit_ind, 64, 64);
  report("fmac_lit", fmac_lit_dep, fmac_lit_ind, 64, 64);
  report("trunc", trunc_dep, trunc_ind, 64, 64);
  report("rcp", rcp_dep, rcp_ind, 64, 64);
  report("bfi", bfi_dep, bfi_ind, 64, 64);
  report("cmp+sel", cmp_sel_dep, cmp_sel_ind, 64, 64);
  report("pk_fma", pkfma_dep, pkfma_ind, 64, 64);
  report("lds_read", lds_dep, nullptr, 64, 64);
  return 0;
}
