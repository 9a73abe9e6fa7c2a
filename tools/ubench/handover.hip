// Host <-> device hand-over primitives on MI355X: what a mppi_compute_control call pays around its kernels.
//  A  empty kernel + hipStreamSynchronize
//  B  empty kernel that stores a sequence number to host-mapped pinned memory; the host spins on it
//  C  H2D copy (1 KB, pinned) + kernel + D2H copy (4 KB, pinned) + hipStreamSynchronize        (what computeControl does)
//  D  kernel reads 1 KB from host-mapped memory, writes 4 KB + flag to host-mapped memory; the host spins
//  E  as D with three dependent kernels (rollout, combine, finalize stand-ins)
// Build: hipcc --offload-arch=gfx950 -O3 handover.hip -o handover.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

__global__ void empty_k()
{
}
__global__ void flag_k(volatile unsigned* flag, unsigned seq)
{
  if (threadIdx.x == 0)
  {
    __atomic_store_n((unsigned*)flag, seq, __ATOMIC_RELEASE);
  }
}
__global__ void copy_k(const float* in, float* out, int n_in, int n_out)
{
  float s = 0.f;
  for (int i = threadIdx.x; i < n_in; i += blockDim.x)
    s += in[i];
  for (int i = threadIdx.x; i < n_out; i += blockDim.x)
    out[i] = s + i;
}
__global__ void copy_flag_k(const float* in, float* out, int n_in, int n_out, volatile unsigned* flag, unsigned seq)
{
  float s = 0.f;
  for (int i = threadIdx.x; i < n_in; i += blockDim.x)
    s += in[i];
  for (int i = threadIdx.x; i < n_out; i += blockDim.x)
    out[i] = s + i;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    __threadfence_system();
    __atomic_store_n((unsigned*)flag, seq, __ATOMIC_RELEASE);
  }
}

static double now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  float *in_h, *out_h, *in_d, *out_d, *in_m, *out_m;
  unsigned* flag_h;
  hipHostMalloc((void**)&in_h, 4096, hipHostMallocDefault);
  hipHostMalloc((void**)&out_h, 16384, hipHostMallocDefault);
  hipHostMalloc((void**)&in_m, 4096, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostMalloc((void**)&out_m, 16384, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostMalloc((void**)&flag_h, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipMalloc((void**)&in_d, 4096);
  hipMalloc((void**)&out_d, 16384);
  float *in_md, *out_md;
  unsigned* flag_d;
  hipHostGetDevicePointer((void**)&in_md, in_m, 0);
  hipHostGetDevicePointer((void**)&out_md, out_m, 0);
  hipHostGetDevicePointer((void**)&flag_d, flag_h, 0);
  *flag_h = 0;
  const int N = 2000;
  unsigned seq = 0;
  for (int rep = 0; rep < 2; rep++)
  {
    double t0 = now_us();
    for (int i = 0; i < N; i++)
    {
      empty_k<<<1, 64, 0, s>>>();
      hipStreamSynchronize(s);
    }
    double a = (now_us() - t0) / N;
    t0 = now_us();
    for (int i = 0; i < N; i++)
    {
      ++seq;
      flag_k<<<1, 64, 0, s>>>(flag_d, seq);
      while (__atomic_load_n(flag_h, __ATOMIC_ACQUIRE) != seq)
      {
      }
    }
    double b = (now_us() - t0) / N;
    t0 = now_us();
    for (int i = 0; i < N; i++)
    {
      hipMemcpyAsync(in_d, in_h, 1024, hipMemcpyHostToDevice, s);
      copy_k<<<1, 256, 0, s>>>(in_d, out_d, 256, 1024);
      hipMemcpyAsync(out_h, out_d, 4096, hipMemcpyDeviceToHost, s);
      hipStreamSynchronize(s);
    }
    double c = (now_us() - t0) / N;
    t0 = now_us();
    for (int i = 0; i < N; i++)
    {
      ++seq;
      in_m[0] = (float)i;
      copy_flag_k<<<1, 256, 0, s>>>(in_md, out_md, 256, 1024, flag_d, seq);
      while (__atomic_load_n(flag_h, __ATOMIC_ACQUIRE) != seq)
      {
      }
    }
    double d = (now_us() - t0) / N;
    t0 = now_us();
    for (int i = 0; i < N; i++)
    {
      ++seq;
      in_m[0] = (float)i;
      copy_k<<<256, 256, 0, s>>>(in_md, out_d, 256, 1024);
      copy_k<<<2, 256, 0, s>>>(out_d, in_d, 256, 256);
      copy_flag_k<<<1, 256, 0, s>>>(in_d, out_md, 256, 1024, flag_d, seq);
      while (__atomic_load_n(flag_h, __ATOMIC_ACQUIRE) != seq)
      {
      }
    }
    double e = (now_us() - t0) / N;
    t0 = now_us();
    for (int i = 0; i < N; i++)
    {
      hipMemcpyAsync(in_d, in_h, 1024, hipMemcpyHostToDevice, s);
      copy_k<<<256, 256, 0, s>>>(in_d, out_d, 256, 1024);
      copy_k<<<2, 256, 0, s>>>(out_d, in_d, 256, 256);
      copy_k<<<1, 256, 0, s>>>(in_d, out_d, 256, 1024);
      hipMemcpyAsync(out_h, out_d, 4096, hipMemcpyDeviceToHost, s);
      hipStreamSynchronize(s);
    }
    double f = (now_us() - t0) / N;
    if (rep == 1)
      printf("A empty+sync %.1f us | B flag spin %.1f us | C h2d+kernel+d2h+sync %.1f us | D mapped in/out + spin %.1f us | "
             "E 3 kernels mapped + spin %.1f us | F 3 kernels copies + sync %.1f us\n", a, b, c, d, e, f);
  }
  return 0;
}
