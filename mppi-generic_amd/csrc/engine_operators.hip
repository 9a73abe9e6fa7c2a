/**
 * engine_operators.hip — kernel-level operators and probes of the C ABI.
 * Part of the implementation of include/mppi_amd.h; see engine_internal.hpp for how the engine is divided and
 * engine_core.hip for the references its logic follows.
 */
#include "engine_internal.hpp"

/* ---------------------------------------------------------------- kernel-level operators ------------------------- */
mppi_status mppi_rollout_costs(mppi_handle h, const float* x0, int stride)
{
  CHECK_HANDLE(h);
  if (!x0)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(h->x0_d, x0, sizeof(float) * h->D * h->S, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->mean_d, h->control_h.data(), sizeof(float) * h->TC, hipMemcpyHostToDevice, h->stream));
  if (h->D == 2)
    HIP_TRY(h, hipMemcpyAsync(h->mean_d + h->TC, h->nominal_control_h.data(), sizeof(float) * h->TC,
                              hipMemcpyHostToDevice, h->stream));
  MPPI_TRY(launchRollout(h, 0, stride));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

mppi_status mppi_enforce_constraints(mppi_handle h, const float* state, float* u)
{
  if (!h)
    return MPPI_ERR_INVALID_ARG;
  if (!u)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_enforce_constraints: null control");
  {
    // host path: no handle lock, no stream — a control publication from the state-estimator thread never queues behind the
    // rollouts of a computeControl in flight (the reference clamps on the host too, controller.cuh:329-345)
    std::lock_guard<std::mutex> params_lock(h->params_mu);
    if (h->model->hostEnforceConstraints(u))
      return MPPI_OK;
  }
  // the plugin overrides enforceConstraints(): a zero-length model step on the device returns the constrained control
  std::vector<float> x(h->S, 0.0f);
  if (state)
    std::copy(state, state + h->S, x.begin());
  return mppi_model_step(h, x.data(), u, 0.0f, 1);
}

mppi_status mppi_model_step(mppi_handle h, float* x, float* u, float dt, int enforce)
{
  CHECK_HANDLE_HOST(h);
  if (!x || !u)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  MPPI_TRY(modelStepInPlace(h, x, u, dt, enforce));
  return MPPI_OK;
}

namespace
{
struct DevBuf
{
  float* p = nullptr;
  ~DevBuf()
  {
    if (p)
      (void)hipFree(p);
  }
  hipError_t alloc(size_t n)
  {
    return hipMalloc((void**)&p, n * sizeof(float));
  }
};
mppi_status opFail(const char* what, hipError_t e)
{
  g_create_error = std::string(what) + ": " + hipGetErrorString(e);
  return MPPI_ERR_HIP;
}
}  // namespace
#define OP_TRY(expr)                  \
  do                                  \
  {                                   \
    hipError_t e__ = (expr);          \
    if (e__ != hipSuccess)            \
      return opFail(#expr, e__);      \
  } while (0)

static mppi_status opDevice(int device)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
  {
    g_create_error = "no HIP device visible (this library has no CPU path)";
    return MPPI_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n)
    return MPPI_ERR_INVALID_ARG;
  OP_TRY(hipSetDevice(device));
  return MPPI_OK;
}

mppi_status mppi_norm_exp(float* costs, int K, float lambda_inv, float baseline, int device)
{
  if (!costs || K <= 0)
    return MPPI_ERR_INVALID_ARG;
  MPPI_TRY(opDevice(device));
  DevBuf d;
  OP_TRY(d.alloc(K));
  OP_TRY(hipMemcpy(d.p, costs, sizeof(float) * K, hipMemcpyHostToDevice));
  // reference: norm_exp_kernel_parallelization_ = 64 (controller.cuh:64) -> grid ceil(K/64) x 64
  hipLaunchKernelGGL(kernels::normExpKernel, dim3((K + 63) / 64), dim3(64), 0, 0, K, d.p, lambda_inv, baseline);
  OP_TRY(hipGetLastError());
  OP_TRY(hipMemcpy(costs, d.p, sizeof(float) * K, hipMemcpyDeviceToHost));
  return MPPI_OK;
}

mppi_status mppi_compute_weights(float* costs, int K, float lambda_inv, float* out2, int device)
{
  if (!costs || !out2 || K <= 0)
    return MPPI_ERR_INVALID_ARG;
  MPPI_TRY(opDevice(device));
  DevBuf d, o;
  OP_TRY(d.alloc(K));
  OP_TRY(o.alloc(2));
  OP_TRY(hipMemcpy(d.p, costs, sizeof(float) * K, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(kernels::computeWeightsKernel, dim3(1), dim3(kernels::COMBINE_THREADS), 0, 0, K, d.p, lambda_inv,
                     o.p);
  OP_TRY(hipGetLastError());
  OP_TRY(hipMemcpy(costs, d.p, sizeof(float) * K, hipMemcpyDeviceToHost));
  OP_TRY(hipMemcpy(out2, o.p, sizeof(float) * 2, hipMemcpyDeviceToHost));
  return MPPI_OK;
}

mppi_status mppi_weighted_reduction(const float* weights, const float* v, float normalizer, int K, int T, int C,
                                    float* u_out, int device)
{
  if (!weights || !v || !u_out || K <= 0 || T <= 0 || C <= 0)
    return MPPI_ERR_INVALID_ARG;
  MPPI_TRY(opDevice(device));
  DevBuf w, vd, u;
  const size_t TC = (size_t)T * C;
  OP_TRY(w.alloc(K));
  OP_TRY(vd.alloc((size_t)K * TC));
  OP_TRY(u.alloc(TC));
  OP_TRY(hipMemcpy(w.p, weights, sizeof(float) * K, hipMemcpyHostToDevice));
  OP_TRY(hipMemcpy(vd.p, v, sizeof(float) * K * TC, hipMemcpyHostToDevice));
  OP_TRY(hipMemset(u.p, 0, sizeof(float) * TC));
  const int per_block = 32;
  hipLaunchKernelGGL(kernels::weightedReductionKernel, dim3((K + per_block - 1) / per_block), dim3(256), 0, 0, w.p,
                     vd.p, u.p, normalizer, (int)TC, K, per_block);
  OP_TRY(hipGetLastError());
  OP_TRY(hipMemcpy(u_out, u.p, sizeof(float) * TC, hipMemcpyDeviceToHost));
  return MPPI_OK;
}

mppi_status mppi_compute_weights_reference_order(float* costs, int K, float lambda, float* stats8, int device)
{
  if (!costs || !stats8 || K <= 0 || !(lambda > 0.0f))
    return MPPI_ERR_INVALID_ARG;
  MPPI_TRY(opDevice(device));
  DevBuf c, w, st;
  OP_TRY(c.alloc(K));
  OP_TRY(w.alloc(K));
  OP_TRY(st.alloc(kernels::STATS_STRIDE));
  OP_TRY(hipMemcpy(c.p, costs, sizeof(float) * K, hipMemcpyHostToDevice));
  OP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernels::exactWeightsKernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kernels::EXACT_WEIGHTS_LDS_BYTES));
  kernels::ExactWeightsArgs a{};
  a.num_rollouts = K;
  a.costs_d = c.p;
  a.weights_d = w.p;
  a.stats_out_d = st.p;
  a.lambda = lambda;
  a.lambda_inv = (float)(1.0 / (double)lambda);
  hipLaunchKernelGGL(kernels::exactWeightsKernel, dim3(1), dim3(kernels::COMBINE_THREADS), kernels::EXACT_WEIGHTS_LDS_BYTES,
                     0, a);
  OP_TRY(hipGetLastError());
  OP_TRY(hipMemcpy(costs, w.p, sizeof(float) * K, hipMemcpyDeviceToHost));
  OP_TRY(hipMemcpy(stats8, st.p, sizeof(float) * kernels::STATS_STRIDE, hipMemcpyDeviceToHost));
  return MPPI_OK;
}

mppi_status mppi_weighted_reduction_reference_order(const float* weights, const float* v, float normalizer, int K, int T,
                                                    int C, int sum_stride, int fma, float* u_out, int device)
{
  if (!weights || !v || !u_out || K <= 0 || T <= 0 || C <= 0 || sum_stride <= 0)
    return MPPI_ERR_INVALID_ARG;
  MPPI_TRY(opDevice(device));
  DevBuf w, vd, u, st, inter;
  const int TC = T * C;
  const int cells = (K - 1) / sum_stride + 1;
  OP_TRY(w.alloc(K));
  OP_TRY(vd.alloc((size_t)K * TC));
  OP_TRY(u.alloc(TC));
  OP_TRY(st.alloc(kernels::STATS_STRIDE));
  OP_TRY(inter.alloc((size_t)cells * TC));
  float sth[kernels::STATS_STRIDE] = { 0.0f, normalizer };
  OP_TRY(hipMemcpy(w.p, weights, sizeof(float) * K, hipMemcpyHostToDevice));
  OP_TRY(hipMemcpy(vd.p, v, sizeof(float) * (size_t)K * TC, hipMemcpyHostToDevice));
  OP_TRY(hipMemcpy(st.p, sth, sizeof(sth), hipMemcpyHostToDevice));
  const dim3 grid((TC + 63) / 64, (cells + kernels::COMBINE_THREADS / 64 - 1) / (kernels::COMBINE_THREADS / 64), 1);
  if (fma)
    hipLaunchKernelGGL(kernels::exactReductionCellsKernel<1>, grid, dim3(kernels::COMBINE_THREADS), 0, 0, w.p, vd.p, st.p, TC,
                       K, sum_stride, cells, inter.p);
  else
    hipLaunchKernelGGL(kernels::exactReductionCellsKernel<0>, grid, dim3(kernels::COMBINE_THREADS), 0, 0, w.p, vd.p, st.p, TC,
                       K, sum_stride, cells, inter.p);
  hipLaunchKernelGGL(kernels::exactReductionFinalKernel, dim3((TC + 63) / 64, 1), dim3(64), 0, 0, inter.p, TC, cells, u.p);
  OP_TRY(hipGetLastError());
  OP_TRY(hipMemcpy(u_out, u.p, sizeof(float) * TC, hipMemcpyDeviceToHost));
  return MPPI_OK;
}


__global__ void philoxNormalKernel(uint64_t seed, uint32_t generation, int TC, int k_begin, int k_end, float* out)
{
  const int qpr = (TC + 3) / 4;  // quads per rollout row
  const int nq = (k_end - k_begin) * qpr;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x)
  {
    const int r = i / qpr, q = i - r * qpr;
    float z[4];
    mppi::rng::normal4(seed, generation, 0u, (uint32_t)(k_begin + r), (uint32_t)q, z);
    for (int l = 0; l < 4 && q * 4 + l < TC; l++)
      out[(size_t)r * TC + q * 4 + l] = z[l];
  }
}

mppi_status mppi_philox_normal(uint64_t seed, uint32_t generation, int K, int T, int C, int k_begin, int k_end,
                               float* eps_out, int device)
{
  if (!eps_out || K <= 0 || T <= 0 || C <= 0 || k_begin < 0 || k_end > K || k_begin >= k_end)
    return MPPI_ERR_INVALID_ARG;
  MPPI_TRY(opDevice(device));
  const size_t n = (size_t)(k_end - k_begin) * T * C;
  DevBuf d;
  OP_TRY(d.alloc(n));
  hipLaunchKernelGGL(philoxNormalKernel, dim3(256), dim3(256), 0, 0, seed, generation, T * C, k_begin, k_end, d.p);
  OP_TRY(hipGetLastError());
  OP_TRY(hipMemcpy(eps_out, d.p, sizeof(float) * n, hipMemcpyDeviceToHost));
  return MPPI_OK;
}

__global__ void detEvalKernel(int func, const float* x, float* y, int n)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
  {
    float r = 0.0f;
    switch (func)
    {
      case 0: r = mppi::det::sin(x[i]); break;
      case 1: r = mppi::det::cos(x[i]); break;
      case 2: r = mppi::det::exp(x[i]); break;
      case 3: r = mppi::det::log(x[i]); break;
      case 4: r = mppi::det::tanh(x[i]); break;
      case 5: r = mppi::det::atan(x[i]); break;
      case 6: r = mppi::det::normalizeAngle(x[i]); break;
      case 7: r = mppi::det::sigmoid(x[i]); break;
      case 8: r = mppi::det::sqrt(x[i]); break;
      case 9: r = 1.0f / x[i]; break;
      case 10:  // packed pair path: element i is evaluated together with its neighbour i ^ 1
      {
        float ra, rb;
        const int j = (i ^ 1) < n ? (i ^ 1) : i;
        mppi::det::tanh2(x[i & ~1], x[(i & ~1) + 1 < n ? (i & ~1) + 1 : i], &ra, &rb);
        r = (i & 1) && (j != i) ? rb : ra;
        break;
      }
      case 11:
      {
        float v[4] = { x[i], x[i] * 0.5f, -x[i], x[i] + 1.0f };
        mppi::det::sigmoid_n<4>(v);
        r = v[0] + v[1] + v[2] + v[3];
        break;
      }
      case 12: r = mppi::det::tan(x[i]); break;
      case 13: r = mppi::det::asin(x[i]); break;
    }
    y[i] = r;
  }
}

extern "C++" {
template <int NC>
__global__ void texture2dQueryKernel(mppi::texture::TwoDTextureHelper<1, NC> helper, const float* __restrict__ points, int n,
                                     int frame, float* __restrict__ out)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
  {
    const float pt[3] = { points[3 * i], points[3 * i + 1], points[3 * i + 2] };
    float r[NC];
    if (frame == 0)
      helper.queryTexture(0, pt, r);
    else if (frame == 1)
      helper.queryTextureAtMapPose(0, pt, r);
    else
      helper.queryTextureAtWorldPose(0, pt, r);
    for (int ch = 0; ch < NC; ch++)
      out[(size_t)i * NC + ch] = r[ch];
  }
}

template <int NC>
static mppi_status texture2dQuery(const float* data, int width, int height, const mppi_texture2d_params* p,
                                  const float* points, int n, int frame, float* out)
{
  DevBuf dd, dp, dout;
  const size_t texels = (size_t)width * height * NC;
  OP_TRY(dd.alloc(texels));
  OP_TRY(dp.alloc((size_t)3 * n));
  OP_TRY(dout.alloc((size_t)n * NC));
  OP_TRY(hipMemcpy(dd.p, data, sizeof(float) * texels, hipMemcpyHostToDevice));
  OP_TRY(hipMemcpy(dp.p, points, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
  mppi::texture::TwoDTextureHelper<1, NC> helper;
  mppi::texture::TextureParams2D& t = helper.textures_[0];
  t.data = dd.p;
  t.width = width;
  t.height = height;
  t.use = 1;
  t.address_mode[0] = p->address_mode[0];
  t.address_mode[1] = p->address_mode[1];
  t.filter_mode = p->filter_mode;
  memcpy(t.border_color, p->border_color, sizeof(t.border_color));
  memcpy(t.origin, p->origin, sizeof(t.origin));
  memcpy(t.rotations, p->rotations, sizeof(t.rotations));
  memcpy(t.resolution, p->resolution, sizeof(t.resolution));
  hipLaunchKernelGGL((texture2dQueryKernel<NC>), dim3((n + 255) / 256), dim3(256), 0, 0, helper, dp.p, n, frame, dout.p);
  OP_TRY(hipGetLastError());
  OP_TRY(hipMemcpy(out, dout.p, sizeof(float) * n * NC, hipMemcpyDeviceToHost));
  return MPPI_OK;
}
}  // extern "C++"

mppi_status mppi_texture2d_query(const float* data, int width, int height, int channels, const mppi_texture2d_params* p,
                                 const float* points, int n, int frame, float* out, int device)
{
  if (!data || !p || !points || !out || n <= 0 || width < 2 || height < 2 || frame < 0 || frame > 2 ||
      (channels != 1 && channels != 4))
    return MPPI_ERR_INVALID_ARG;
  MPPI_TRY(opDevice(device));
  return channels == 1 ? texture2dQuery<1>(data, width, height, p, points, n, frame, out) :
                         texture2dQuery<4>(data, width, height, p, points, n, frame, out);
}

extern "C++" {
__global__ void boundaryProbeKernel(int* sink)
{
  if (sink && threadIdx.x == 1024)
    *sink = 0;
}
}
mppi_status mppi_measure_launch_boundary(int device, int n, float* us_per_launch)
{
  if (n <= 0 || !us_per_launch)
    return MPPI_ERR_INVALID_ARG;
  MPPI_TRY(opDevice(device));
  hipStream_t s = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e == hipSuccess)
    e = hipEventCreate(&a);
  if (e == hipSuccess)
    e = hipEventCreate(&b);
  // the launches are replayed from a graph: enqueued one by one the host is the bottleneck (~3 us per launch), which is not
  // what separates two kernels of an iteration whose launches were queued long before the first one finished
  float ms = 0.0f;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  if (e == hipSuccess)
    e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (e == hipSuccess)
  {
    for (int i = 0; i < n; i++)
      hipLaunchKernelGGL(boundaryProbeKernel, dim3(256), dim3(64), 0, s, (int*)nullptr);
    e = hipStreamEndCapture(s, &graph);
  }
  if (e == hipSuccess)
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e == hipSuccess)
    e = hipGraphLaunch(exec, s);  // warm-up replay
  if (e == hipSuccess)
    e = hipStreamSynchronize(s);
  if (e == hipSuccess)
    e = hipEventRecord(a, s);
  if (e == hipSuccess)
    e = hipGraphLaunch(exec, s);
  if (e == hipSuccess)
    e = hipEventRecord(b, s);
  if (e == hipSuccess)
    e = hipEventSynchronize(b);
  if (e == hipSuccess)
    e = hipEventElapsedTime(&ms, a, b);
  if (exec)
    (void)hipGraphExecDestroy(exec);
  if (graph)
    (void)hipGraphDestroy(graph);
  if (a)
    (void)hipEventDestroy(a);
  if (b)
    (void)hipEventDestroy(b);
  if (s)
    (void)hipStreamDestroy(s);
  if (e != hipSuccess)
    return opFail("mppi_measure_launch_boundary", e);
  *us_per_launch = ms * 1e3f / (float)n;
  return MPPI_OK;
}

extern "C++" {
/** one wave per workgroup, N_CHAIN v_fmac_f32 (4-byte encoding) per loop trip on eight independent accumulators, inside ONE
 *  asm statement (between separate statements the compiler puts an s_nop 0 after every dependent v_fmac, and a wave alone on
 *  its SIMD pays an issue slot for it: the first version of this probe measured 3.4 ns per fmac + nop pair): what a lone
 *  wave pays per instruction (~1.9 ns) */
#define MPPI_PROBE_8 \
  "v_fmac_f32_e32 %0, %8, %9\nv_fmac_f32_e32 %1, %8, %9\nv_fmac_f32_e32 %2, %8, %9\nv_fmac_f32_e32 %3, %8, %9\n" \
  "v_fmac_f32_e32 %4, %8, %9\nv_fmac_f32_e32 %5, %8, %9\nv_fmac_f32_e32 %6, %8, %9\nv_fmac_f32_e32 %7, %8, %9\n"
#define MPPI_PROBE_64 MPPI_PROBE_8 MPPI_PROBE_8 MPPI_PROBE_8 MPPI_PROBE_8 MPPI_PROBE_8 MPPI_PROBE_8 MPPI_PROBE_8 MPPI_PROBE_8
template <int N_CHAIN>
__global__ void __launch_bounds__(64) issueProbeKernel(float* sink, int trips, float a, float b)
{
  static_assert(N_CHAIN == 256, "four blocks of 64 per trip");
  float x0 = (float)threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < trips; i++)
    asm volatile(MPPI_PROBE_64 MPPI_PROBE_64 MPPI_PROBE_64 MPPI_PROBE_64
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                 : "v"(a), "v"(b));
  const float x = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (sink && x == 123.456f)
    *sink = x;
}
#undef MPPI_PROBE_64
#undef MPPI_PROBE_8
}
mppi_status mppi_measure_issue_interval(int device, float* ns_per_instruction)
{
  if (!ns_per_instruction)
    return MPPI_ERR_INVALID_ARG;
  MPPI_TRY(opDevice(device));
  hipStream_t s = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e == hipSuccess)
    e = hipEventCreate(&a);
  if (e == hipSuccess)
    e = hipEventCreate(&b);
  // two chain lengths, differenced: launch ramp, loop overhead and the tail fall out.  The probes follow a ~10 ms warm-up
  // launch on the same stream with no host synchronisation in between: short kernels after an idle gap run below the
  // sustained clock (first version of this probe: 3.4 ns instead of 1.9).
  constexpr int CHAIN = 256;
  const int trips[2] = { 256, 1280 };
  hipEvent_t c = nullptr;
  if (e == hipSuccess)
    e = hipEventCreate(&c);
  float best = 1e30f;
  for (int rep = 0; rep < 3 && e == hipSuccess; rep++)
  {
    hipLaunchKernelGGL((issueProbeKernel<CHAIN>), dim3(256), dim3(64), 0, s, (float*)nullptr, 20000, 0.999f, 1e-3f);
    e = hipEventRecord(a, s);
    hipLaunchKernelGGL((issueProbeKernel<CHAIN>), dim3(256), dim3(64), 0, s, (float*)nullptr, trips[0], 0.999f, 1e-3f);
    if (e == hipSuccess)
      e = hipEventRecord(b, s);
    hipLaunchKernelGGL((issueProbeKernel<CHAIN>), dim3(256), dim3(64), 0, s, (float*)nullptr, trips[1], 0.999f, 1e-3f);
    if (e == hipSuccess)
      e = hipEventRecord(c, s);
    if (e == hipSuccess)
      e = hipEventSynchronize(c);
    float t0 = 0.0f, t1 = 0.0f;
    if (e == hipSuccess)
      e = hipEventElapsedTime(&t0, a, b);
    if (e == hipSuccess)
      e = hipEventElapsedTime(&t1, b, c);
    if (e == hipSuccess && t1 - t0 < best)
      best = t1 - t0;
  }
  if (c)
    (void)hipEventDestroy(c);
  if (a)
    (void)hipEventDestroy(a);
  if (b)
    (void)hipEventDestroy(b);
  if (s)
    (void)hipStreamDestroy(s);
  if (e != hipSuccess)
    return opFail("mppi_measure_issue_interval", e);
  *ns_per_instruction = best * 1e6f / (float)((trips[1] - trips[0]) * CHAIN);
  return MPPI_OK;
}

mppi_status mppi_det_eval(int func, const float* x, float* y, int n, int device)
{
  if (!x || !y || n <= 0)
    return MPPI_ERR_INVALID_ARG;
  MPPI_TRY(opDevice(device));
  DevBuf dx, dy;
  OP_TRY(dx.alloc(n));
  OP_TRY(dy.alloc(n));
  OP_TRY(hipMemcpy(dx.p, x, sizeof(float) * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(detEvalKernel, dim3(256), dim3(256), 0, 0, func, dx.p, dy.p, n);
  OP_TRY(hipGetLastError());
  OP_TRY(hipMemcpy(y, dy.p, sizeof(float) * n, hipMemcpyDeviceToHost));
  return MPPI_OK;
}
