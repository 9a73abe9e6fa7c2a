// Probe for include/mppi_amd/plugin/parallel_utils.hpp: every Parallel1Dir / Parallel2Dir direction on a (4, 3, 2) block in a
// (2, 3, 2) grid against the index arithmetic written out on the host, and loadArrayParallel's runtime-count form over
// counts and offsets that select its 16-byte, 8-byte and element-wise paths.  Built and run by tests/test_parallel_utils.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mppi_amd/plugin/parallel_utils.hpp"

using mppi::p1::Parallel1Dir;
using mppi::p2::Parallel2Dir;
constexpr int N1 = 14, N2 = 7;

template <int D>
__device__ void probe1(int* out)
{
  int i = -1, s = -1;
  mppi::p1::getParallel1DIndex<(Parallel1Dir)D>(i, s);
  out[2 * D] = i;
  out[2 * D + 1] = s;
  if constexpr (D + 1 < N1)
    probe1<D + 1>(out);
}
template <int D>
__device__ void probe2(int* out)
{
  int a = -1, b = -1, sa = -1, sb = -1;
  mppi::p2::getParallel2DIndex<(Parallel2Dir)D>(a, b, sa, sb);
  out[4 * D] = a;
  out[4 * D + 1] = b;
  out[4 * D + 2] = sa;
  out[4 * D + 3] = sb;
  if constexpr (D + 1 < N2)
    probe2<D + 1>(out);
}

__global__ void probeKernel(int* out)
{
  const int threads = blockDim.x * blockDim.y * blockDim.z;
  const int block = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int t = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  int* o = out + (size_t)(block * threads + t) * (2 * N1 + 4 * N2);
  probe1<0>(o);
  probe2<0>(o + 2 * N1);
}

__global__ void copyKernel(float* dst, const float* src, int off1, int off2, int n)
{
  mppi::p1::loadArrayParallel<Parallel1Dir::THREAD_XY>(dst, off1, src, off2, n);
  mppi::p1::loadArrayParallel<5, Parallel1Dir::THREAD_ZY>(dst, 200, src, 3);
}

#define CHECK(x)                                                            \
  do                                                                        \
  {                                                                         \
    hipError_t e = (x);                                                     \
    if (e != hipSuccess)                                                    \
    {                                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main()
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
  {
    fprintf(stderr, "no HIP device\n");
    return 1;
  }
  const dim3 grid(2, 3, 2), block(4, 3, 2);
  const int threads = 24, blocks = 12, per = 2 * N1 + 4 * N2;
  int* out_d;
  CHECK(hipMalloc((void**)&out_d, sizeof(int) * threads * blocks * per));
  hipLaunchKernelGGL(probeKernel, grid, block, 0, 0, out_d);
  std::vector<int> out(threads * blocks * per);
  CHECK(hipMemcpy(out.data(), out_d, sizeof(int) * out.size(), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int bz = 0; bz < 2; bz++)
    for (int by = 0; by < 3; by++)
      for (int bx = 0; bx < 2; bx++)
        for (int tz = 0; tz < 2; tz++)
          for (int ty = 0; ty < 3; ty++)
            for (int tx = 0; tx < 4; tx++)
            {
              const int t[3] = { tx, ty, tz }, d[3] = { 4, 3, 2 }, b[3] = { bx, by, bz }, g[3] = { 2, 3, 2 };
              const int blk = bx + 2 * (by + 3 * bz), thr = tx + 4 * (ty + 3 * tz);
              const int* o = &out[(size_t)(blk * threads + thr) * per];
              auto two = [&](int A, int B, int& i, int& s) { i = t[A] + d[A] * t[B]; s = d[A] * d[B]; };
              int want[2 * N1];
              for (int a = 0; a < 3; a++)
              {
                want[2 * a] = t[a];
                want[2 * a + 1] = d[a];
                want[2 * (10 + a)] = t[a] + d[a] * b[a];
                want[2 * (10 + a) + 1] = g[a] * d[a];
              }
              // THREAD_XY, YX, XZ, ZX, YZ, ZY = enumerators 3..8
              const int pairs[6][2] = { { 0, 1 }, { 1, 0 }, { 0, 2 }, { 2, 0 }, { 1, 2 }, { 2, 1 } };
              for (int k = 0; k < 6; k++)
                two(pairs[k][0], pairs[k][1], want[2 * (3 + k)], want[2 * (3 + k) + 1]);
              want[2 * 9] = thr;
              want[2 * 9 + 1] = threads;
              want[2 * 13] = 0;
              want[2 * 13 + 1] = 1;
              for (int k = 0; k < 2 * N1; k++)
                if (o[k] != want[k])
                {
                  if (bad++ < 10)
                    fprintf(stderr, "p1 dir %d %s: got %d want %d (block %d thread %d)\n", k / 2, k % 2 ? "step" : "index", o[k],
                            want[k], blk, thr);
                }
              // Parallel2Dir: XY, XZ, YZ, YX, ZX, ZY, NONE
              const int p2[6][2] = { { 0, 1 }, { 0, 2 }, { 1, 2 }, { 1, 0 }, { 2, 0 }, { 2, 1 } };
              for (int k = 0; k < 7; k++)
              {
                const int* q = o + 2 * N1 + 4 * k;
                const int w[4] = { k < 6 ? t[p2[k][0]] : 0, k < 6 ? t[p2[k][1]] : 0, k < 6 ? d[p2[k][0]] : 1, k < 6 ? d[p2[k][1]] : 1 };
                for (int j = 0; j < 4; j++)
                  if (q[j] != w[j] && bad++ < 10)
                    fprintf(stderr, "p2 dir %d item %d: got %d want %d\n", k, j, q[j], w[j]);
              }
            }
  // loadArrayParallel: (count, off1, off2) choosing the 16-byte, the 8-byte and the element-wise path
  const int cases[5][3] = { { 16, 4, 8 }, { 16, 2, 4 }, { 15, 0, 0 }, { 12, 4, 1 }, { 0, 0, 0 } };
  float *src_d, *dst_d;
  CHECK(hipMalloc((void**)&src_d, sizeof(float) * 256));
  CHECK(hipMalloc((void**)&dst_d, sizeof(float) * 256));
  std::vector<float> src(256), dst(256);
  for (int i = 0; i < 256; i++)
    src[i] = 1.0f + i;
  CHECK(hipMemcpy(src_d, src.data(), sizeof(float) * 256, hipMemcpyHostToDevice));
  for (auto& c : cases)
  {
    CHECK(hipMemset(dst_d, 0, sizeof(float) * 256));
    hipLaunchKernelGGL(copyKernel, dim3(1), dim3(2, 3, 2), 0, 0, dst_d, src_d, c[1], c[2], c[0]);
    CHECK(hipMemcpy(dst.data(), dst_d, sizeof(float) * 256, hipMemcpyDeviceToHost));
    for (int i = 0; i < 256; i++)
    {
      float want = 0.0f;
      if (i >= c[1] && i < c[1] + c[0])
        want = src[c[2] + i - c[1]];
      if (i >= 200 && i < 205)
        want = src[3 + i - 200];
      if (dst[i] != want && bad++ < 10)
        fprintf(stderr, "copy (n %d, off %d <- %d): dst[%d] = %g want %g\n", c[0], c[1], c[2], i, dst[i], want);
    }
  }
  if (bad)
  {
    fprintf(stderr, "%d mismatches\n", bad);
    return 2;
  }
  printf("PARALLEL OK: %d threads x %d directions, %d copy cases\n", threads * blocks, N1 + N2, 5);
  return 0;
}
