/**
 * Probe of tests/test_kernarg_layout.py: a kernel with by-value arguments of awkward sizes and alignments (4-, 8- and 16-byte
 * aligned structs, a trailing int — the shapes of the rollout kernels' argument lists).  The host program prints what
 * mppi::kernels::KernargLayout says the argument offsets are; the test compares them with the `.offset` fields the compiler
 * wrote into the code object's metadata for this very kernel.
 */
#include <cstdio>
#include "mppi_amd/engine/kernarg_view.hpp"

struct A4
{
  float a[5];
};
struct B8
{
  float* p;
  float b[3];
};
struct alignas(16) C16
{
  float c[6];
};
struct D4
{
  int d[7];
};

__global__ void kernargLayoutProbeKernel(A4 a, B8 b, C16 c, D4 d, const int n, float* out)
{
  const mppi::kernels::kernarg_ptr_t base = mppi::kernels::kernargBase();
  using L = mppi::kernels::KernargLayout<A4, B8, C16, D4, int, float*>;
  const A4* pa = mppi::kernels::kernargObject<A4>(base, L::offset<0>());
  const B8* pb = mppi::kernels::kernargObject<B8>(base, L::offset<1>());
  const C16* pc = mppi::kernels::kernargObject<C16>(base, L::offset<2>());
  const D4* pd = mppi::kernels::kernargObject<D4>(base, L::offset<3>());
  // through the view and through the named arguments: the same values (run on the GPU by test_gpu_ops.py)
  out[0] = pa->a[4] - a.a[4];
  out[1] = pb->b[2] - b.b[2];
  out[2] = pc->c[5] - c.c[5];
  out[3] = (float)(pd->d[6] - d.d[6]);
  out[4] = pa->a[4] + pb->b[2] + pc->c[5] + (float)pd->d[6] + (float)n;
}

int main(int argc, char** argv)
{
  using L = mppi::kernels::KernargLayout<A4, B8, C16, D4, int, float*>;
  printf("%zu %zu %zu %zu %zu %zu\n", L::offset<0>(), L::offset<1>(), L::offset<2>(), L::offset<3>(), L::offset<4>(), L::offset<5>());
  if (argc > 1)
  {  // on a GPU: launch and check
    float* out_d = nullptr;
    if (hipMalloc((void**)&out_d, 8 * sizeof(float)) != hipSuccess)
      return 3;
    A4 a{ { 1, 2, 3, 4, 5.5f } };
    B8 b{ out_d, { 6, 7, 8.25f } };
    C16 c{ { 9, 10, 11, 12, 13, 14.125f } };
    D4 d{ { 1, 2, 3, 4, 5, 6, 77 } };
    hipLaunchKernelGGL(kernargLayoutProbeKernel, dim3(1), dim3(64), 0, 0, a, b, c, d, 1000, out_d);
    float out[5] = { -1, -1, -1, -1, -1 };
    if (hipMemcpy(out, out_d, sizeof(out), hipMemcpyDeviceToHost) != hipSuccess)
      return 4;
    printf("%g %g %g %g %g\n", out[0], out[1], out[2], out[3], out[4]);
    return (out[0] == 0 && out[1] == 0 && out[2] == 0 && out[3] == 0 && out[4] == 5.5f + 8.25f + 14.125f + 77.0f + 1000.0f) ? 0 : 5;
  }
  return 0;
}
