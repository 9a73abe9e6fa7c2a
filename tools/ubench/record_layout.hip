// The streamed merge's first trip (rolloutPipelineKernel, STREAM_MERGE): every block's two sampler waves read, of all 256 block
// records of the previous launch, the 8-byte tail and one 16-byte column quad — lane = record, records 416 B apart, i.e. every
// lane of every load instruction touches its own cache line: 2 waves x 8 instructions x 64 lines per CU, 131k line requests
// per launch for 12 KB of useful data per wave.  Is the first trip's memory round trip bound by that request count rather than
// by latency?  Same reads against a TRANSPOSED copy ([quad][record][4 floats], tails [record][2]): a load instruction's 64 lanes
// read 1 KB / 512 B of contiguous memory (8 / 4 lines).
//   hipcc --offload-arch=gfx950 -O3 -o record_layout tools/ubench/record_layout.hip
// A writer kernel (256 blocks, one record each: the previous launch's epilogues, other CUs / XCDs) runs before every reader launch.
// Prints per layout: reader kernel time (HIP events, writer excluded by measuring it alone too) and the in-kernel entry -> data-back
// time of sampler wave 0 (s_memtime ticks, mean over blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int RECORDS = 256, TC = 100, PS = 104, QUADS = TC / 4;

__global__ void __launch_bounds__(256) writer(float* rows, float* quads_t, float* tails_t, float seed)
{
  const int b = blockIdx.x, l = threadIdx.x;
  if (l < PS)
    rows[(size_t)b * PS + l] = seed + b + 0.001f * l;
  if (l < TC)
    quads_t[((size_t)(l / 4) * RECORDS + b) * 4 + (l & 3)] = seed + b + 0.001f * l;
  if (l < 2)
    tails_t[2 * b + l] = seed + b + 0.001f * (TC + l);
}

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool TRANSPOSED>
__global__ void __launch_bounds__(256) reader(const float* rows, const float* quads_t, const float* tails_t, float* out,
                                              unsigned long long* stamps)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.0f;
  if (wave == 0 || wave == 3)  // the two sampler waves; their first trips: quad 0 and quad 1
  {
    const int q = wave == 0 ? 0 : 1;
    f2 t[4];
    f4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
      const int b = lane + 64 * i;
      if (TRANSPOSED)
      {
        t[i] = *reinterpret_cast<const f2*>(tails_t + 2 * b);
        v[i] = *reinterpret_cast<const f4*>(quads_t + ((size_t)q * RECORDS + b) * 4);
      }
      else
      {
        t[i] = *reinterpret_cast<const f2*>(rows + (size_t)b * PS + TC);
        v[i] = *reinterpret_cast<const f4*>(rows + (size_t)b * PS + 4 * q);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 4; i++)
      acc += t[i].x + t[i].y + v[i].x + v[i].y + v[i].z + v[i].w;
    if (lane == 0 && wave == 0)
      stamps[blockIdx.x] = t1 - t0;
  }
  __syncthreads();
  if (acc == 1234.5f)
    out[threadIdx.x] = acc;
}

int main()
{
  float *rows, *quads_t, *tails_t, *out;
  unsigned long long* stamps;
  hipMalloc(&rows, RECORDS * PS * sizeof(float));
  hipMalloc(&quads_t, QUADS * RECORDS * 4 * sizeof(float));
  hipMalloc(&tails_t, RECORDS * 2 * sizeof(float));
  hipMalloc(&out, 256 * sizeof(float));
  hipMalloc(&stamps, 256 * sizeof(unsigned long long));
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int N = 1000;
  auto run = [&](int what, const char* name) {
    for (int pass = 0; pass < 3; pass++)
    {
      hipEventRecord(e0, s);
      for (int i = 0; i < N; i++)
      {
        hipLaunchKernelGGL(writer, dim3(RECORDS), dim3(256), 0, s, rows, quads_t, tails_t, (float)i);
        if (what == 1)
          hipLaunchKernelGGL(reader<false>, dim3(256), dim3(256), 0, s, rows, quads_t, tails_t, out, stamps);
        if (what == 2)
          hipLaunchKernelGGL(reader<true>, dim3(256), dim3(256), 0, s, rows, quads_t, tails_t, out, stamps);
      }
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(256);
      hipMemcpy(h.data(), stamps, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double mean = 0, mx = 0;
      for (auto t : h)
      {
        mean += (double)t;
        mx = t > mx ? (double)t : mx;
      }
      printf("%-28s pass %d: %.3f us per writer+reader pair; wave 0 entry -> data back: mean %.0f max %.0f ticks\n", name, pass,
             1000.0 * ms / N, mean / 256.0, mx);
    }
  };
  run(0, "writer alone");
  run(1, "rows (lane = record, 416 B)");
  run(2, "transposed copy");
  run(1, "rows (lane = record, 416 B)");
  run(2, "transposed copy");
  return 0;
}
