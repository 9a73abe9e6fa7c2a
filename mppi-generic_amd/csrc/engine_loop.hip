/**
 * engine_loop.hip — device-resident loop (mppi_optimize), timing, kernel choice.
 * Part of the implementation of include/mppi_amd.h; see engine_internal.hpp for how the engine is divided and
 * engine_core.hip for the references its logic follows.
 */
#include "engine_internal.hpp"

/* ---------------------------------------------------------------- device-resident loop --------------------------- */
mppi_status mppi_upload_state(mppi_handle h, const float* x0)
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
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->external_iteration = 0;
  return MPPI_OK;
}

mppi_status mppi_get_optimal_control(mppi_handle h, float* u_out)
{
  CHECK_HANDLE(h);
  if (!u_out)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(u_out, h->mean_d, sizeof(float) * h->D * h->TC, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

mppi_status mppi_optimize(mppi_handle h, int n, int synchronize)
{
  CHECK_HANDLE(h);
  if (n < 0)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_optimize: negative iteration count");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  PendingRecordsGuard guard{ h };
  for (int i = 0; i < n; i++)  // opt_iter of mppi_controller.cu:160: std_dev_decay^i shapes iteration i of this call
    MPPI_TRY(iteration(h, i, h->last_stride));
  MPPI_TRY(flushMerge(h));  // streamed merge: the last iteration's records become mean_d / stats_d here
  if (synchronize)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

mppi_status mppi_time_iterations(mppi_handle h, int n, float* ms_total, float* ms_rollout)
{
  CHECK_HANDLE(h);
  if (n <= 0 || !ms_total)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_time_iterations: bad arguments");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  PendingRecordsGuard guard{ h };
  // pass 1: whole iterations between two events.  Sharded handle whose exchange is driven by the caller (no library
  // communicator): the iteration is not the library's to time — *ms_total = 0 and only the kernel pass below runs.
  *ms_total = 0.0f;
  if (!exchangeActive(h) || h->p2p_ready || (h->comm && g_ncclAllGather))
  {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipEventRecord(h->ev_a, h->stream));
    for (int i = 0; i < n; i++)
      MPPI_TRY(iteration(h, 0, h->last_stride));
    MPPI_TRY(flushMerge(h));  // the n-th iteration's merge belongs to the n timed iterations
    HIP_TRY(h, hipEventRecord(h->ev_b, h->stream));
    HIP_TRY(h, hipEventSynchronize(h->ev_b));
    HIP_TRY(h, hipEventElapsedTime(ms_total, h->ev_a, h->ev_b));
  }
  if (ms_rollout)
  {
    // pass 2: the rollout kernel alone, n launches back to back between two events (events around every single launch
    // would add ~3 us of event packets per launch to a ~25 us kernel; this way the figure agrees with the kernel's
    // duration in a rocprofv3 --kernel-trace).  The merge launches are left out: the rollouts do not depend on them here.
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipEventRecord(h->ev_a, h->stream));
    for (int i = 0; i < n; i++)
    {
      MPPI_TRY(launchRollout(h, 0, h->last_stride));
      if (streamMergeApplies(h))
      {  // the kernel that is timed is the one iterations run: it merges the previous launch's records in its sampler waves
        h->pending_records_d = h->partials_d;
        std::swap(h->partials_d, h->partials_alt_d);
      }
    }
    HIP_TRY(h, hipEventRecord(h->ev_b, h->stream));
    HIP_TRY(h, hipEventSynchronize(h->ev_b));
    float sum = 0.0f;
    HIP_TRY(h, hipEventElapsedTime(&sum, h->ev_a, h->ev_b));
    // leave the handle as an iteration would: merged records, updated mean
    if (h->pending_records_d)
      MPPI_TRY(flushMerge(h));
    else if (!exchangeActive(h))
      MPPI_TRY(launchCombine(h, h->partials_d, h->num_blocks, 1, nullptr, h->cfg.num_rollouts));
    else
      MPPI_TRY(launchCombine(h, h->partials_d, h->num_blocks, 0, h->send_d, h->K_local));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *ms_rollout = sum;
  }
  return MPPI_OK;
}

mppi_status mppi_choose_kernel(mppi_handle h, int num_evaluations, int* chosen_variant, float* fused_ms, float* pipeline_ms)
{
  CHECK_HANDLE(h);
  if (num_evaluations <= 0)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_choose_kernel: num_evaluations must be > 0");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const bool pipe_ok = h->cfg.controller != MPPI_CONTROLLER_ROBUST &&
                       ((h->model->supportsPipeline() && h->bx == 64 && h->by == 1) ||
                        h->model->supportsPipelineFold(h->bx, h->by, h->bz) || h->model->supportsPipelineRep(h->bx, h->by, h->bz)) &&
                       h->model->rolloutSharedBytes(h->bx, h->by, h->bz, h->cfg.num_timesteps, h->D, true) <= MAX_LDS_BYTES;
  const bool fused_ok = h->cfg.controller == MPPI_CONTROLLER_ROBUST ||
                        (h->model->supportsShape(h->bx, h->by, h->bz) &&
                         h->model->rolloutSharedBytes(h->bx, h->by, h->bz, h->cfg.num_timesteps, h->D, false) <= MAX_LDS_BYTES);
  float t_ms[2] = { INFINITY, INFINITY };  // [0] fused, [1] pipeline
  const bool was = h->pipeline;
  const uint32_t generation = h->generation;
  /* What is compared is what an iteration costs with either structure: on a handle that merges on its own (no exchange) the
   * fused kernel needs a merge launch behind every rollout launch, the pipelined one may merge the previous records in its
   * sampler waves (one launch per iteration) — timing the bare rollout kernels would hold ~3-5 us per iteration against the
   * pipeline on small problems.  The trial iterations overwrite mean_d / stats_d: both are saved and put back.  A K-sharded
   * handle's merge needs its peers and is the same launch for both structures: there the rollout kernels alone are timed. */
  const bool whole_iterations = !exchangeActive(h) && !tsallisActive(h);
  PendingRecordsGuard guard{ h };
  float* saved_d = nullptr;
  const size_t mean_floats = (size_t)h->D * h->TC, stats_floats = (size_t)h->D * kernels::STATS_STRIDE;
  if (whole_iterations)
  {
    HIP_TRY(h, hipMalloc((void**)&saved_d, sizeof(float) * (mean_floats + stats_floats)));
    (void)hipMemcpyAsync(saved_d, h->mean_d, sizeof(float) * mean_floats, hipMemcpyDeviceToDevice, h->stream);
    (void)hipMemcpyAsync(saved_d + mean_floats, h->stats_d, sizeof(float) * stats_floats, hipMemcpyDeviceToDevice, h->stream);
  }
  auto trial = [&]() -> mppi_status { return whole_iterations ? iteration(h, 0, h->last_stride) : launchRollout(h, 0, h->last_stride); };
  auto restore = [&]() {
    if (!saved_d)
      return;
    (void)hipMemcpyAsync(h->mean_d, saved_d, sizeof(float) * mean_floats, hipMemcpyDeviceToDevice, h->stream);
    (void)hipMemcpyAsync(h->stats_d, saved_d + mean_floats, sizeof(float) * stats_floats, hipMemcpyDeviceToDevice, h->stream);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(saved_d);
    saved_d = nullptr;
  };
  for (int v = 0; v < 2; v++)
  {
    if ((v == 0 && !fused_ok) || (v == 1 && !pipe_ok))
      continue;
    h->pipeline = v == 1;
    mppi_status st = trial();  // warm-up (code object load, LDS attribute)
    if (st == MPPI_OK && whole_iterations)
      st = flushMerge(h);
    if (st == MPPI_OK)
      st = hipStreamSynchronize(h->stream) == hipSuccess ? MPPI_OK : MPPI_ERR_HIP;
    if (st == MPPI_OK && hipEventRecord(h->ev_a, h->stream) != hipSuccess)
      st = MPPI_ERR_HIP;
    for (int i = 0; st == MPPI_OK && i < num_evaluations; i++)
      st = trial();
    if (st == MPPI_OK && whole_iterations)
      st = flushMerge(h);  // the last iteration's merge belongs to the timed iterations
    if (st == MPPI_OK && (hipEventRecord(h->ev_b, h->stream) != hipSuccess || hipEventSynchronize(h->ev_b) != hipSuccess ||
                          hipEventElapsedTime(&t_ms[v], h->ev_a, h->ev_b) != hipSuccess))
      st = MPPI_ERR_HIP;
    if (st != MPPI_OK)
    {
      h->pipeline = was;
      h->generation = generation;
      restore();
      return st == MPPI_ERR_HIP ? fail(h, st, "mppi_choose_kernel: HIP error while timing the rollout kernels") : st;
    }
    t_ms[v] /= (float)num_evaluations;
  }
  restore();
  h->generation = generation;  // the trial launches do not advance the noise stream
  if (!fused_ok && !pipe_ok)
  {
    h->pipeline = was;
    return fail(h, MPPI_ERR_LDS_OVERFLOW, "mppi_choose_kernel: neither kernel structure fits this configuration");
  }
  h->pipeline = t_ms[1] < t_ms[0];
  if (chosen_variant)
    *chosen_variant = h->pipeline ? MPPI_KERNEL_PIPELINE : MPPI_KERNEL_FUSED;
  if (fused_ms)
    *fused_ms = t_ms[0];
  if (pipeline_ms)
    *pipeline_ms = t_ms[1];
  return MPPI_OK;
}

/** diagnostics: the host-side stamps of the last low-latency Vanilla mppi_compute_control (see mppi_handle_s::host_stamps_us) */
mppi_status mppi_debug_host_stamps(mppi_handle h, double* out8)
{
  CHECK_HANDLE_HOST(h);
  if (!out8)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  std::copy(h->host_stamps_us, h->host_stamps_us + 8, out8);
  return MPPI_OK;
}

mppi_status mppi_synchronize(mppi_handle h)
{
  CHECK_HANDLE(h);
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}
