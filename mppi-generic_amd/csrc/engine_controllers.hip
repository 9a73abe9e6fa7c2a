/**
 * engine_controllers.hip — mppi_compute_control (Vanilla / Colored, Tube, Robust), hand-over, getters, slide.
 * Part of the implementation of include/mppi_amd.h; see engine_internal.hpp for how the engine is divided and
 * engine_core.hip for the references its logic follows.
 */
#include "engine_internal.hpp"

/* ---------------------------------------------------------------- control loop ----------------------------------- */
mppi_status mppi_set_nominal_control(mppi_handle h, const float* u)
{
  CHECK_HANDLE(h);
  if (!u)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_nominal_control: null");
  std::copy(u, u + h->control_h.size(), h->control_h.begin());
  if (h->D == 2)  // Tube: both trajectories; RMPPI: nominal_control_trajectory_ = init_control_traj (:33)
    std::copy(u, u + h->control_h.size(), h->nominal_control_h.begin());
  return MPPI_OK;
}

mppi_status mppi_inject_noise(mppi_handle h, const float* eps, int n_iters)
{
  CHECK_HANDLE(h);
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (n_iters <= 0 || !eps)
  {
    h->noise_source = h->cfg.noise_source == MPPI_NOISE_INJECTED ? MPPI_NOISE_PHILOX_FUSED : h->cfg.noise_source;
    return MPPI_OK;
  }
  const size_t n = (size_t)n_iters * epsFloatsPerIteration(h);
  if (n_iters != h->n_eps_iters)
  {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->eps_d)
      HIP_TRY(h, hipFree(h->eps_d));
    h->eps_d = nullptr;
    HIP_TRY(h, hipMalloc((void**)&h->eps_d, n * sizeof(float)));
    h->n_eps_iters = n_iters;
  }
  HIP_TRY(h, hipMemcpyAsync(h->eps_d, eps, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->noise_source = MPPI_NOISE_INJECTED;
  h->generation = 0;
  return MPPI_OK;
}

static mppi_status uploadTube(mppi_handle h, const float* x0_actual)
{
  // both initial states and both nominal controls through the pinned input block: one copy
  float* in = h->in_pin_h;
  std::copy(x0_actual, x0_actual + h->S, in);
  std::copy(h->tube_x_h.begin(), h->tube_x_h.end(), in + h->S);
  float* mean = in + (h->mean_d - h->in_block_d);
  std::copy(h->control_h.begin(), h->control_h.end(), mean);
  std::copy(h->nominal_control_h.begin(), h->nominal_control_h.end(), mean + h->TC);
  const size_t n = (size_t)(h->mean_d - h->in_block_d) + 2 * (size_t)h->TC;
  HIP_TRY(h, hipMemcpyAsync(h->in_block_d, in, sizeof(float) * n, hipMemcpyHostToDevice, h->stream));
  return MPPI_OK;
}

/** stats of system z from the floats the merge kernel wrote */
void parseStats(mppi_handle h, const float* st)
{
  mppi_system_stats* sys[2] = { &h->stats_h.real_sys, &h->stats_h.nominal_sys };
  if (h->cfg.controller == MPPI_CONTROLLER_ROBUST)  // system 0 is the NOMINAL one there (robust_mppi_controller.cu:637-640)
    std::swap(sys[0], sys[1]);
  for (int z = 0; z < h->D; z++)
  {
    const float* s = st + z * kernels::STATS_STRIDE;
    sys[z]->baseline = s[0];
    sys[z]->normalizer = s[1];
    sys[z]->free_energy_mean = s[2];
    sys[z]->free_energy_variance = s[3];
    sys[z]->free_energy_modified_variance = s[4];
    if (s[6] != 0.0f)  // combineKernel gave up waiting for a peer's record (P2P exchange)
      h->exchange_failed = true;
  }
}

/** Host writes to the inbox (io_in_h) are complete, in order, before anything that makes the device read it.  With the BAR inbox
 *  the block is device memory behind the PCIe BAR, mapped write-combined: stores sit in the core's WC buffers until a fence (or
 *  an uncached write that happens to flush them) — every path that hands the inbox to a kernel goes through here, not only the
 *  `direct` Vanilla one (round-5 advice: the ingest launches relied on the launch path flushing the buffers).  Pinned host
 *  memory needs no fence beyond the release the doorbell write already is; one is issued anyway on non-x86 builds. */
static inline void publishInbox(mppi_handle h)
{
#if defined(__x86_64__)
  if (h->bar_inbox)
    __builtin_ia32_sfence();
#else
  (void)h;
  std::atomic_thread_fence(std::memory_order_seq_cst);
#endif
}

/** inbox -> in_block_d (one tiny kernel on the handle's stream), behind publishInbox() */
static inline void launchIngest(mppi_handle h)
{
  publishInbox(h);
  hipLaunchKernelGGL(kernels::ingestKernel, dim3(1), dim3(256), 0, h->stream, h->io_in_dev, h->in_block_d, (int)h->in_floats);
}

/** spins on a flag the finalize kernel raises in host memory; falls back to a stream synchronisation when the flag does not
 *  show within the limit (a failed launch, a wedged device): the caller then sees the HIP error instead of a hang */
static mppi_status waitHostFlag(mppi_handle h, int idx, unsigned seq)
{
  using clock = std::chrono::steady_clock;
  const clock::time_point t0 = clock::now();
  volatile unsigned* flag = h->io_flags_h + idx;
  unsigned spins = 0;
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
  {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if ((++spins & 0x3ff) == 0 && std::chrono::duration<double>(clock::now() - t0).count() > 2.0)
    {
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      if (h->side_stream)
        HIP_TRY(h, hipStreamSynchronize(h->side_stream));
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
        return fail(h, MPPI_ERR_HIP, "the finalize kernel finished without raising its hand-over flag");
      break;
    }
  }
  return MPPI_OK;
}

/** the same for "the flag has reached seq" (sequence numbers only grow; wrap-around safe): the trajectory phases of a split
 *  hand-over run in order on the side stream, so a later call's flag value covers the earlier ones */
static mppi_status waitHostFlagReached(mppi_handle h, int idx, unsigned seq)
{
  using clock = std::chrono::steady_clock;
  const clock::time_point t0 = clock::now();
  volatile unsigned* flag = h->io_flags_h + idx;
  unsigned spins = 0;
  while ((int)(__atomic_load_n(flag, __ATOMIC_ACQUIRE) - seq) < 0)
  {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if ((++spins & 0x3ff) == 0 && std::chrono::duration<double>(clock::now() - t0).count() > 2.0)
    {
      if (h->side_stream)
        HIP_TRY(h, hipStreamSynchronize(h->side_stream));
      if ((int)(__atomic_load_n(flag, __ATOMIC_ACQUIRE) - seq) < 0)
        return fail(h, MPPI_ERR_HIP, "the trajectory phase of an earlier call finished without raising its flag");
      break;
    }
  }
  return MPPI_OK;
}

/** the state / output trajectories of the last low-latency computeControl: wait for the finalize kernel's second flag */
static mppi_status ensureTrajectories(mppi_handle h)
{
  if (!h->traj_pending)
    return MPPI_OK;
  MPPI_TRY(waitHostFlag(h, 1, h->io_seq));
  const int T = h->cfg.num_timesteps;
  const float* out = h->io_out_h;
  if (h->cfg.controller == MPPI_CONTROLLER_ROBUST)
  {  // system 0: the nominal trajectory (the next call's candidates start from it), system 1: the real one
    MPPI_TRY(waitHostFlag(h, 3, h->io_seq));
    h->traj_pending = false;
    const float* xs = out + (h->state_out_d - h->out_block_d);
    std::copy(xs, xs + (size_t)T * h->S, h->nominal_state_h.begin());
    std::copy(xs + (size_t)T * h->S, xs + (size_t)2 * T * h->S, h->state_h.begin());
    if (!allFinite(h->nominal_state_h))
      return fail(h, MPPI_ERR_NAN, "non-finite value in the nominal state sequence of the last mppi_compute_control");
    return MPPI_OK;
  }
  if (h->cfg.controller == MPPI_CONTROLLER_TUBE)
  {  // system 0: the actual system, system 1: the nominal one (its first state is where the next call starts from)
    MPPI_TRY(waitHostFlag(h, 3, h->io_seq));
    h->traj_pending = false;
    const float* xs = out + (h->state_out_d - h->out_block_d);
    std::copy(xs, xs + (size_t)T * h->S, h->state_h.begin());
    std::copy(xs + (size_t)T * h->S, xs + (size_t)2 * T * h->S, h->nominal_state_h.begin());
    if (!allFinite(h->state_h) || !allFinite(h->nominal_state_h))
      return fail(h, MPPI_ERR_NAN, "non-finite value in the state sequences of the last mppi_compute_control");
    return MPPI_OK;
  }
  h->traj_pending = false;
  std::copy(out + (h->state_out_d - h->out_block_d), out + (h->state_out_d - h->out_block_d) + (size_t)T * h->S,
            h->state_h.begin());
  if (!allFinite(h->state_h))  // base_plant.hpp:515-528 checks the state trajectory as well as the control
    return fail(h, MPPI_ERR_NAN, "non-finite value in the state sequence of the last mppi_compute_control");
  return MPPI_OK;
}

static mppi_status computeControlVanilla(mppi_handle h, const float* x0_true, int stride)
{
  const int T = h->cfg.num_timesteps;
  h->pending_records_d = nullptr;  // this call uploads its own mean: nothing of an earlier (failed) call may be merged over it
  PendingRecordsGuard guard{ h };
  // ColoredMPPI state leash (colored_mppi_controller.cu:150-156): the optimisation starts from the state of the previous
  // solution at index leash_jump, pulled towards the measured state by at most the leash per dimension
  std::vector<float> leashed;
  const float* x0 = x0_true;
  if (h->cfg.controller == MPPI_CONTROLLER_COLORED && h->leash_active)
  {
    MPPI_TRY(ensureTrajectories(h));
    leashed.resize(h->S);
    h->model->hostEnforceLeash(x0_true, &h->state_h[(size_t)h->leash_jump * h->S], h->leash_dist.data(), leashed.data());
    x0 = leashed.data();
  }
  kernels::FinalizeArgs a{};
  a.scratch_d = h->fin_scratch_d;
  a.control_in_d = h->mean_d;
  a.history_d = h->history_d;
  a.history_stride = 0;
  a.x0_d = h->x0_d;
  a.dt = h->cfg.dt;
  a.num_timesteps = T;
  a.smooth_mask = 1;
  a.constrain_mask = 1;
  // ColoredMPPI clamps only control channel 1 after smoothing (colored_mppi_controller.cu:232-237)
  a.constrain_mode = h->cfg.controller == MPPI_CONTROLLER_COLORED ? 1 : 0;
  std::string err;
  if (h->low_latency)
  {
    /* Inputs and results travel through host memory mapped into the device: no copy command, no stream synchronisation.
     * The call returns when the control sequence and the merge statistics are out (flag 0), while the finalize kernel
     * still re-rolls the state trajectory of u* — a T-step serial chain, ~1/3 of the call for Cartpole; the trajectory
     * getters wait for flag 1 (tools/ubench/handover.hip: 3 kernels + spin 15 us against 23 us with copies + synchronise). */
    // a caller that never asked for the previous trajectories: the kernel must be done with io_out and — BAR inbox — with the
    // inputs before the host overwrites them.  Split hand-over: the trajectory phase reads its carry block, writes nothing the
    // control phase of this call writes, and the flag waits of the getters take sequence numbers: nothing to wait for
    if (h->traj_pending && !h->split_finalize)
      MPPI_TRY(waitHostFlag(h, 1, h->io_seq));
    h->traj_pending = false;
    const std::chrono::steady_clock::time_point t_call = std::chrono::steady_clock::now();
    auto stamp = [&](int i) {
      h->host_stamps_us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count();
    };
    float* in = h->io_in_h;
    std::copy(x0, x0 + h->S, in + (h->x0_d - h->in_block_d));
    std::copy(h->control_h.begin(), h->control_h.end(), in + (h->mean_d - h->in_block_d));
    std::copy(h->history_h.begin(), h->history_h.end(), in + (h->history_d - h->in_block_d));
    stamp(0);
    // BAR inbox: no ingest launch — the kernels of this call read the inbox (device memory the stores above went to) themselves
    const bool direct = h->bar_inbox && h->cfg.num_iters >= 1 && h->reduction_mode == MPPI_REDUCTION_FUSED && !tsallisActive(h) &&
                        !exchangeActive(h);
    struct SourceGuard  // the overrides never outlive the call
    {
      mppi_handle h;
      ~SourceGuard()
      {
        h->x0_src_d = h->mean_src_d = nullptr;
      }
    } source_guard{ h };
    if (direct)
    {
      publishInbox(h);  // the write-combined stores are out before the doorbell of the first launch
      h->x0_src_d = h->io_in_dev + (h->x0_d - h->in_block_d);
      h->mean_src_d = h->io_in_dev + (h->mean_d - h->in_block_d);
      a.x0_d = h->x0_src_d;
      a.history_d = h->io_in_dev + (h->history_d - h->in_block_d);
    }
    else
    {
      launchIngest(h);
      HIP_TRY(h, hipGetLastError());
    }
    stamp(1);
    for (int it = 0; it < h->cfg.num_iters; it++)
      MPPI_TRY(iteration(h, it, stride));
    stamp(2);
    // The last iteration's records (streamed merge) are still un-merged.  Split hand-over on a model with the plain one-wave
    // finalize form: the control phase merges them itself (kernels::mergeControlKernel) — one launch on the call's critical
    // path where combineKernel + control phase were two.  Otherwise: merge now; everything below reads mean_d / stats_d.
    const float* fuse_records = nullptr;
    if (h->pending_records_d && h->split_finalize && h->merge_control_enabled && !a.scratch_d && streamMergeApplies(h) &&
        h->model->supportsMergeControl(T))
    {
      fuse_records = h->pending_records_d;
      h->pending_records_d = nullptr;
    }
    else
      MPPI_TRY(flushMerge(h));
    stamp(3);
    a.control_out_d = h->io_out_dev + (h->ctrl_out_d - h->out_block_d);
    a.state_out_d = h->io_out_dev + (h->state_out_d - h->out_block_d);
    a.output_out_d = h->io_out_dev + (h->output_out_d - h->out_block_d);
    a.stats_in_d = h->stats_d;
    a.stats_out_d = h->io_out_dev + (h->stats_d - h->out_block_d);
    a.stats_floats = kernels::STATS_STRIDE;
    a.flags_d = h->io_flags_dev;
    a.seq = ++h->io_seq;
    float* carry = nullptr;
    if (h->split_finalize)
    {
      // this call's carry block was last read by the trajectory phase of the call two hand-overs ago: its flag is up, or we wait
      const unsigned p = a.seq & 1u;
      if (h->carry_seq[p] != 0)
        MPPI_TRY(waitHostFlagReached(h, 1, h->carry_seq[p]));
      carry = h->carry_d + (size_t)p * h->in_floats;
      a.phases = 1;
      a.carry_d = carry;
      a.carry_src_d = direct ? h->io_in_dev : h->in_block_d;
      a.carry_floats = (int)h->in_floats;
      a.carry_mean_off = (int)(h->mean_d - h->in_block_d);
      a.carry_ready_d = reinterpret_cast<unsigned*>(h->carry_d + 2 * h->in_floats) + 2 * p;
    }
    auto ingest_ranges = [&](const float* src) -> mppi_status {
      hipLaunchKernelGGL(kernels::ingestRangesKernel, dim3(1), dim3(256), 0, h->stream, src, h->in_block_d,
                         (int)(h->mean_d - h->in_block_d), (int)(h->history_d - h->in_block_d),
                         (int)(h->in_floats - (size_t)(h->history_d - h->in_block_d)));
      HIP_TRY(h, hipGetLastError());
      return MPPI_OK;
    };
    // Single-launch hand-over with the BAR inbox (MPPI_AMD_SPLIT_FINALIZE=0): the device-resident copy of x0 / history is taken
    // from the INBOX, so it has to be taken before the finalize kernel raises flag 1 — the next call waits for nothing else
    // before it rewrites the inbox (round-5 advice: behind the finalize kernel the copy could read a half-rewritten inbox).  It
    // touches neither what the finalize kernel reads (inbox, mean_d) nor what it writes.
    if (direct && !h->split_finalize)
      MPPI_TRY(ingest_ranges(h->io_in_dev));
    mppi_status st;
    if (fuse_records)
    {
      kernels::MergeControlArgs m{};
      m.records_t_d = recordsTransposed(h, const_cast<float*>(fuse_records));
      m.num_records = h->num_blocks;
      m.lambda = h->cfg.lambda;
      m.num_rollouts_total = h->cfg.num_rollouts;
      m.mean_out_d = h->mean_d;
      m.stats_d = h->stats_d;
      h->n_merge_launches++;  // (the merge of this call: inside the control phase's launch)
      st = h->model->launchMergeControl(a, m, h->stream, err);
    }
    else
      st = h->model->launchFinalize(1, a, h->stream, err);
    if (st != MPPI_OK)
      return fail(h, st, err);
    if (h->split_finalize)
    {  // the trajectory phase, on the side stream: it waits for the control phase's carry block by itself (no event between the
       // streams), and reads nothing else
      kernels::FinalizeArgs b = a;
      b.phases = 2;
      b.carry_d = nullptr;
      b.control_in_d = carry + (h->mean_d - h->in_block_d);
      b.x0_d = carry + (h->x0_d - h->in_block_d);
      b.smooth_mask = 0;
      b.scratch_d = h->fin_scratch2_d;
      const mppi_status st2 = h->model->launchFinalize(1, b, h->side_stream, err);
      if (st2 != MPPI_OK)
        return fail(h, st2, err);
      HIP_TRY(h, hipEventRecord(h->ev_side, h->side_stream));
      h->side_pending = true;
      h->carry_seq[a.seq & 1u] = a.seq;
    }
    if (direct && h->split_finalize)
    {  // behind the control phase, off the caller's path: the device-resident x0 / history later mppi_optimize / operator calls
       // read — from the CARRY block (the host may be rewriting the inbox for its next call by now; the carry block of this parity
       // is not rewritten before the call after next, which first waits for this call's flag 1)
      MPPI_TRY(ingest_ranges(carry));
    }
    h->out_pin_fresh = false;
    h->results_in_io = true;
    h->traj_pending = true;  // set before the wait: a failing wait must not leave io_out unguarded for the next call
    stamp(4);
    MPPI_TRY(waitHostFlag(h, 0, h->io_seq));
    stamp(5);
    const float* out = h->io_out_h;
    std::copy(out, out + (size_t)T * h->C, h->control_h.begin());
    parseStats(h, out + (h->stats_d - h->out_block_d));
    h->stats_h_fresh = true;
    stamp(6);
    if (!allFinite(h->control_h))
      return fail(h, MPPI_ERR_NAN, "mppi_compute_control: non-finite value in the control sequence");
    return MPPI_OK;
  }
  // one hand-over in (x0, nominal control, control history), one back (control, state and output trajectories, stats):
  // two copies through pinned memory and a single synchronisation per call
  float* in = h->in_pin_h;
  std::copy(x0, x0 + h->S, in + (h->x0_d - h->in_block_d));
  std::copy(h->control_h.begin(), h->control_h.end(), in + (h->mean_d - h->in_block_d));
  std::copy(h->history_h.begin(), h->history_h.end(), in + (h->history_d - h->in_block_d));
  HIP_TRY(h, hipMemcpyAsync(h->in_block_d, in, sizeof(float) * h->in_floats, hipMemcpyHostToDevice, h->stream));
  for (int it = 0; it < h->cfg.num_iters; it++)
    MPPI_TRY(iteration(h, it, stride));
  MPPI_TRY(flushMerge(h));  // the last iteration's records (streamed merge): everything below reads mean_d / stats_d
  a.control_out_d = h->ctrl_out_d;
  a.state_out_d = h->state_out_d;
  a.output_out_d = h->output_out_d;
  const mppi_status st = h->model->launchFinalize(1, a, h->stream, err);
  if (st != MPPI_OK)
    return fail(h, st, err);
  HIP_TRY(h, hipMemcpyAsync(h->out_pin_h, h->out_block_d, sizeof(float) * h->out_floats, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const float* out = h->out_pin_h;
  std::copy(out, out + (size_t)T * h->C, h->control_h.begin());
  std::copy(out + (h->state_out_d - h->out_block_d), out + (h->state_out_d - h->out_block_d) + (size_t)T * h->S,
            h->state_h.begin());
  parseStats(h, out + (h->stats_d - h->out_block_d));
  h->out_pin_fresh = true;
  h->results_in_io = false;
  // base_plant.hpp:515-528 checks both the control and the state trajectory
  if (!allFinite(h->control_h) || !allFinite(h->state_h))
    return fail(h, MPPI_ERR_NAN, "mppi_compute_control: non-finite value in the control or state sequence");
  return MPPI_OK;
}

/** reference: Tube-MPPI/tube_mppi_controller.cu:157-299 */
static mppi_status computeControlTube(mppi_handle h, const float* x0, int stride)
{
  const int S = h->S;
  if (!h->nominal_state_init)
  {
    std::copy(x0, x0 + S, h->nominal_state_h.begin());
    std::copy(x0, x0 + S, h->tube_x_h.begin());
    h->nominal_state_init = true;
  }
  if (h->low_latency)
  {
    /* Inputs and results through host memory mapped into the device, flags instead of copies + synchronisations (see
     * computeControlVanilla).  Every optimisation pass needs both trajectories on the host (the nominal system is replaced by
     * the actual one when that is the better of the two, :264-277), so the loop waits for all four flags; the final smoothing
     * pass returns with the control sequences and leaves its two trajectories to ensureTrajectories(). */
    // (split hand-over: nothing of this call touches what the last call's trajectory phase reads or writes — see
    // computeControlVanilla — and the nominal system's state is tube_x_h, not row 0 of a trajectory still on its way)
    if (!h->split_finalize)
      MPPI_TRY(ensureTrajectories(h));
    const int T = h->cfg.num_timesteps;
    auto stage_inputs = [&]() -> mppi_status {
      float* in = h->io_in_h;
      std::copy(x0, x0 + S, in + (h->x0_d - h->in_block_d));
      std::copy(h->tube_x_h.begin(), h->tube_x_h.end(), in + (h->x0_d - h->in_block_d) + S);
      float* mean = in + (h->mean_d - h->in_block_d);
      std::copy(h->control_h.begin(), h->control_h.end(), mean);
      std::copy(h->nominal_control_h.begin(), h->nominal_control_h.end(), mean + h->TC);
      std::copy(h->history_h.begin(), h->history_h.end(), in + (h->history_d - h->in_block_d));
      launchIngest(h);
      HIP_TRY(h, hipGetLastError());
      return MPPI_OK;
    };
    auto finalize_flagged = [&](const int smooth_mask) -> mppi_status {
      kernels::FinalizeArgs a{};
  a.scratch_d = h->fin_scratch_d;
      a.control_in_d = h->mean_d;
      a.history_d = h->history_d;
      a.history_stride = 0;
      a.x0_d = h->x0_d;
      a.dt = h->cfg.dt;
      a.num_timesteps = T;
      a.smooth_mask = smooth_mask;
      a.constrain_mask = 0;
      a.constrain_mode = 0;
      a.control_out_d = h->io_out_dev + (h->ctrl_out_d - h->out_block_d);
      a.state_out_d = h->io_out_dev + (h->state_out_d - h->out_block_d);
      a.output_out_d = h->io_out_dev + (h->output_out_d - h->out_block_d);
      a.stats_in_d = h->stats_d;
      a.stats_out_d = h->io_out_dev + (h->stats_d - h->out_block_d);
      a.stats_floats = 2 * kernels::STATS_STRIDE;
      a.flags_d = h->io_flags_dev;
      a.seq = ++h->io_seq;
      std::string err;
      float* carry = nullptr;
      if (h->split_finalize)
      {  // as computeControlVanilla: control phase here, both systems' re-rollouts on the side stream from the carry block
        const unsigned p = a.seq & 1u;
        if (h->carry_seq[p] != 0)
        {
          MPPI_TRY(waitHostFlagReached(h, 1, h->carry_seq[p]));
          MPPI_TRY(waitHostFlagReached(h, 3, h->carry_seq[p]));
        }
        carry = h->carry_d + (size_t)p * h->in_floats;
        a.phases = 1;
        a.carry_d = carry;
        a.carry_src_d = h->in_block_d;  // (ingested; tubeSelectKernel has put the chosen nominal state and control there)
        a.carry_floats = (int)h->in_floats;
        a.carry_mean_off = (int)(h->mean_d - h->in_block_d);
        a.carry_ready_d = reinterpret_cast<unsigned*>(h->carry_d + 2 * h->in_floats) + 2 * p;
      }
      const mppi_status st = h->model->launchFinalize(2, a, h->stream, err);
      if (st != MPPI_OK)
        return fail(h, st, err);
      if (h->split_finalize)
      {
        kernels::FinalizeArgs b = a;
        b.phases = 2;
        b.carry_d = nullptr;
        b.control_in_d = carry + (h->mean_d - h->in_block_d);
        b.x0_d = carry + (h->x0_d - h->in_block_d);
        b.smooth_mask = 0;
        b.scratch_d = h->fin_scratch2_d;
        const mppi_status st2 = h->model->launchFinalize(2, b, h->side_stream, err);
        if (st2 != MPPI_OK)
          return fail(h, st2, err);
        HIP_TRY(h, hipEventRecord(h->ev_side, h->side_stream));
        h->side_pending = true;
        h->carry_seq[a.seq & 1u] = a.seq;
      }
      h->out_pin_fresh = false;
      h->results_in_io = true;
      h->traj_pending = true;  // set before the waits: a failing wait must not leave io_out unguarded for the next call
      MPPI_TRY(waitHostFlag(h, 0, h->io_seq));
      MPPI_TRY(waitHostFlag(h, 2, h->io_seq));
      const float* out = h->io_out_h;
      std::copy(out, out + (size_t)T * h->C, h->control_h.begin());
      std::copy(out + (size_t)T * h->C, out + (size_t)2 * T * h->C, h->nominal_control_h.begin());
      parseStats(h, out + (h->stats_d - h->out_block_d));
      h->stats_h_fresh = true;
      return MPPI_OK;
    };
    /* Round 5: ONE hand-over per call.  Between two optimisation passes the reference decides on the host whether the nominal
     * system restarts from the actual one (:264-277) — after computing both state trajectories, of which the decision needs
     * nothing and the next pass only row 0, the initial state.  Rounds 2-4 mirrored that: finalize pass, wait for both
     * trajectories, decide, stage, second finalize pass (control on the host after 86 us at config 3).  The decision is two
     * baselines the merge has just written: tubeSelectKernel takes it on the device (nominal mean and initial state
     * overwritten where the actual system wins), the passes chain without the host, and a single finalize pass — smoothing the
     * nominal control, re-rolling both trajectories — hands everything over.  Same values in every host-visible field. */
    MPPI_TRY(stage_inputs());
    for (int it = 0; it < h->cfg.num_iters; it++)
    {
      MPPI_TRY(iteration(h, it, stride));
      hipLaunchKernelGGL(kernels::tubeSelectKernel, dim3(1), dim3(256), 0, h->stream, h->stats_d, h->mean_d, h->x0_d, h->TC, S,
                         h->nominal_threshold);
      HIP_TRY(h, hipGetLastError());
    }
    // smoothControlTrajectory() smooths the nominal control (:281, :325-329), then computeStateTrajectory(state)
    MPPI_TRY(finalize_flagged(/*smooth nominal*/ 2));
    if (h->cfg.num_iters > 0)
    {
      const float* st1 = h->io_out_h + (h->stats_d - h->out_block_d) + kernels::STATS_STRIDE;
      // tubeSelectKernel: bit 0 = the LAST pass kept the nominal system (nominalStateUsed), bit 1 = the nominal system's initial
      // state on the device is the actual one — after a take-over in ANY pass of this call (the reference's
      // nominal_state_trajectory_ persists across the passes, tube_mppi_controller.cu:268-277), not only in the last
      const int sel = (int)st1[7];
      h->stats_h.nominal_state_used = sel & 1;
      if (sel & 2)
        std::copy(x0, x0 + S, h->tube_x_h.begin());
    }
    if (!allFinite(h->control_h) || !allFinite(h->nominal_control_h))
      return fail(h, MPPI_ERR_NAN, "mppi_compute_control: non-finite value in the control sequence");
    return MPPI_OK;
  }
  std::vector<float>* co[2] = { &h->control_h, &h->nominal_control_h };
  std::vector<float>* so[2] = { &h->state_h, &h->nominal_state_h };
  for (int it = 0; it < h->cfg.num_iters; it++)
  {
    MPPI_TRY(uploadTube(h, x0));
    MPPI_TRY(iteration(h, it, stride));
    // new means -> host control_ / nominal_control_trajectory_, then both state trajectories (:255-263)
    MPPI_TRY(finalize(h, h->mean_d, 0, 0, co, so));
    MPPI_TRY(fetchStats(h));
    if (h->stats_h.real_sys.baseline < h->stats_h.nominal_sys.baseline + h->nominal_threshold)
    {
      h->stats_h.nominal_state_used = 0;
      h->nominal_state_h = h->state_h;
      std::copy(x0, x0 + S, h->tube_x_h.begin());
      h->nominal_control_h = h->control_h;
    }
    else
    {
      h->stats_h.nominal_state_used = 1;
    }
  }
  // smoothControlTrajectory() smooths the nominal control (:281, :325-329), then computeStateTrajectory(state)
  HIP_TRY(h, hipMemcpyAsync(h->ctrl_in_d, h->control_h.data(), sizeof(float) * h->TC, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->ctrl_in_d + h->TC, h->nominal_control_h.data(), sizeof(float) * h->TC,
                            hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->x0_d + S, h->tube_x_h.data(), sizeof(float) * S, hipMemcpyHostToDevice, h->stream));
  MPPI_TRY(finalize(h, h->ctrl_in_d, /*smooth nominal*/ 2, 0, co, so));
  if (!allFinite(h->control_h) || !allFinite(h->nominal_control_h) || !allFinite(h->state_h) ||
      !allFinite(h->nominal_state_h))
    return fail(h, MPPI_ERR_NAN, "mppi_compute_control: non-finite value in the control or state sequence");
  return MPPI_OK;
}


/* ---------------------------------------------------------------- Robust MPPI host logic ------------------------- */
/** reference: robust_mppi_controller.cu:480-500 (computeLineSearchWeights) — [3][NC] row-major */
static void rmLineSearchWeights(int nc, std::vector<float>& w)
{
  w.assign((size_t)3 * nc, 0.0f);
  const int half = nc / 2;
  for (int i = 0; i < half + 1; i++)
  {
    w[0 * nc + i] = 1 - i / float(half);
    w[1 * nc + i] = i / float(half);
    w[2 * nc + i] = 0.0f;
  }
  for (int i = 1; i < half + 1; i++)
  {
    w[0 * nc + half + i] = 0.0f;
    w[1 * nc + half + i] = 1 - i / float(half);
    w[2 * nc + half + i] = i / float(half);
  }
}
/** reference: robust_mppi_controller.cu:502-512 — round((0, stride, stride) . weights) */
static void rmImportanceSamplerStrides(int stride, int nc, const std::vector<float>& w, std::vector<int>& out)
{
  out.resize(nc);
  for (int i = 0; i < nc; i++)
  {
    float acc = 0.0f * w[0 * nc + i];
    acc += (float)stride * w[1 * nc + i];
    acc += (float)stride * w[2 * nc + i];
    out[i] = (int)roundf(acc);
  }
}
/** reference: robust_mppi_controller.cu:514-545 (computeCandidateBaseline, computeBestIndex); expf / logf -> det */
static void rmBestIndex(mppi_handle h)
{
  const int nc = h->num_candidates, ns = h->samples_per_candidate;
  const float lambda = h->cfg.lambda;
  float baseline = h->rm_cand_costs[0];
  for (int i = 1; i < nc * ns; i++)
    if (h->rm_cand_costs[i] < baseline)
      baseline = h->rm_cand_costs[i];
  h->rm_cand_free_energy.assign(nc, 0.0f);
  for (int i = 0; i < nc; i++)
  {
    float fe = 0.0f;
    for (int j = 0; j < ns; j++)
      fe += mppi::det::exp((float)(-1.0 / (double)lambda * (double)(h->rm_cand_costs[(size_t)i * ns + j] - baseline)));
    fe = (float)((double)fe / (1.0 * ns));
    fe = -lambda * mppi::det::log(fe) + baseline;
    h->rm_cand_free_energy[i] = fe;
    if (fe < h->value_function_threshold)
      h->best_index = i;
  }
}

static mppi_status rmEnsureCandidateBuffers(mppi_handle h)
{
  const int n = h->num_candidates * h->samples_per_candidate;
  if (n <= h->cand_capacity && h->num_candidates <= h->cand_capacity_nc)
    return MPPI_OK;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->cand_states_d)
    (void)hipFree(h->cand_states_d);
  if (h->cand_costs_d)
    (void)hipFree(h->cand_costs_d);
  if (h->cand_strides_d)
    (void)hipFree(h->cand_strides_d);
  if (h->cand_io_h)
    (void)hipHostFree(h->cand_io_h);
  h->cand_states_d = h->cand_costs_d = nullptr;
  h->cand_strides_d = nullptr;
  h->cand_io_h = h->cand_io_dev = nullptr;
  HIP_TRY(h, hipHostMalloc((void**)&h->cand_io_h, sizeof(float) * ((size_t)h->num_candidates * (h->S + 1) + n),
                           hipHostMallocMapped | hipHostMallocCoherent));
  HIP_TRY(h, hipHostGetDevicePointer((void**)&h->cand_io_dev, h->cand_io_h, 0));
  HIP_TRY(h, hipMalloc((void**)&h->cand_states_d, sizeof(float) * h->num_candidates * h->S));
  HIP_TRY(h, hipMalloc((void**)&h->cand_costs_d, sizeof(float) * n));
  HIP_TRY(h, hipMalloc((void**)&h->cand_strides_d, sizeof(int) * h->num_candidates));
  h->cand_capacity = n;
  h->cand_capacity_nc = h->num_candidates;
  return MPPI_OK;
}

/** the nominal state trajectory from rm_nominal_state under nominal_control_h (computeStateTrajectoryHelper) */
static mppi_status rmNominalStateTrajectory(mppi_handle h)
{
  if (h->low_latency)
  {
    // inputs with the input block, the trajectory back through the device-mapped output block + flag (system 0 only)
    const int T = h->cfg.num_timesteps;
    float* in = h->io_in_h;
    std::copy(h->rm_nominal_state.begin(), h->rm_nominal_state.begin() + h->S, in + (h->x0_d - h->in_block_d));
    std::copy(h->nominal_control_h.begin(), h->nominal_control_h.end(), in + (h->mean_d - h->in_block_d));
    launchIngest(h);
    HIP_TRY(h, hipGetLastError());
    kernels::FinalizeArgs a{};
    a.scratch_d = h->fin_scratch_d;
    a.control_in_d = h->mean_d;
    a.history_d = h->history_d;
    a.history_stride = 0;
    a.x0_d = h->x0_d;
    a.dt = h->cfg.dt;
    a.num_timesteps = T;
    a.smooth_mask = 0;
    a.constrain_mask = 0;
    a.constrain_mode = 0;
    a.control_out_d = h->io_out_dev + (h->ctrl_out_d - h->out_block_d);
    a.state_out_d = h->io_out_dev + (h->state_out_d - h->out_block_d);
    a.output_out_d = h->io_out_dev + (h->output_out_d - h->out_block_d);
    a.flags_d = h->io_flags_dev;
    a.seq = ++h->io_seq;
    std::string err;
    const mppi_status st = h->model->launchFinalize(1, a, h->stream, err);
    if (st != MPPI_OK)
      return fail(h, st, err);
    h->out_pin_fresh = false;
    h->results_in_io = true;
    MPPI_TRY(waitHostFlag(h, 1, h->io_seq));
    const float* xs = h->io_out_h + (h->state_out_d - h->out_block_d);
    std::copy(xs, xs + (size_t)T * h->S, h->nominal_state_h.begin());
    return MPPI_OK;
  }
  HIP_TRY(h, hipMemcpyAsync(h->x0_d, h->rm_nominal_state.data(), sizeof(float) * h->S, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->ctrl_in_d, h->nominal_control_h.data(), sizeof(float) * h->TC, hipMemcpyHostToDevice,
                            h->stream));
  std::vector<float>* co[2] = { nullptr, nullptr };
  std::vector<float>* so[2] = { &h->nominal_state_h, nullptr };
  return finalize(h, h->ctrl_in_d, 0, 0, co, so, 1);
}

/** reference: robust_mppi_controller.cu:571-626 (computeNominalStateAndStride) */
static mppi_status rmNominalStateAndStride(mppi_handle h, const float* state, int stride)
{
  const int S = h->S, nc = h->num_candidates, ns = h->samples_per_candidate;
  if (!h->rm_nominal_init)
  {
    std::copy(state, state + S, h->rm_nominal_state.begin());
    h->rm_nominal_init = true;
    h->nominal_stride = 0;
    return MPPI_OK;
  }
  // (injected noise on a K-sharded handle: the slab this call consumes must hold the GLOBAL rollouts' rows 0 .. ns-1 on
  //  every rank — the evaluation samples are the same rows for every candidate and every rank, robust_mppi_controller.cu:596)
  if (ns > h->K_local && h->noise_source == MPPI_NOISE_INJECTED)
    return fail(h, MPPI_ERR_INVALID_ARG, "samples_per_candidate exceeds the injected noise rows");
  // candidates = [nominal_x_k, nominal_x_k+1, real_x_k+1] * line search weights (:350-362)
  rmLineSearchWeights(nc, h->rm_line_weights);
  h->rm_cand_states.assign((size_t)nc * S, 0.0f);
  for (int c = 0; c < nc; c++)
    for (int i = 0; i < S; i++)
    {
      float acc = h->nominal_state_h[0 * S + i] * h->rm_line_weights[0 * nc + c];
      acc += h->nominal_state_h[1 * S + i] * h->rm_line_weights[1 * nc + c];
      acc += state[i] * h->rm_line_weights[2 * nc + c];
      h->rm_cand_states[(size_t)c * S + i] = acc;
    }
  rmImportanceSamplerStrides(stride, nc, h->rm_line_weights, h->rm_cand_strides);
  MPPI_TRY(rmEnsureCandidateBuffers(h));
  float* cand_costs_dev = h->cand_costs_d;
  if (h->low_latency)
  {
    // candidate states and strides stay in host memory mapped into the device (the kernel reads them once, in place), the
    // nominal control goes up with the input block: no copy command
    std::copy(h->rm_cand_states.begin(), h->rm_cand_states.end(), h->cand_io_h);
    std::memcpy(h->cand_io_h + (size_t)nc * S, h->rm_cand_strides.data(), sizeof(int) * nc);
    std::copy(h->nominal_control_h.begin(), h->nominal_control_h.end(), h->io_in_h + (h->mean_d - h->in_block_d));
    launchIngest(h);
    HIP_TRY(h, hipGetLastError());
    cand_costs_dev = h->cand_io_dev + (size_t)nc * (S + 1);
  }
  else
  {
    HIP_TRY(h, hipMemcpyAsync(h->cand_states_d, h->rm_cand_states.data(), sizeof(float) * nc * S, hipMemcpyHostToDevice,
                              h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->cand_strides_d, h->rm_cand_strides.data(), sizeof(int) * nc, hipMemcpyHostToDevice,
                              h->stream));
    // copyNominalControlToDevice: distribution 0 <- nominal control (:409-412)
    HIP_TRY(h, hipMemcpyAsync(h->mean_d, h->nominal_control_h.data(), sizeof(float) * h->TC, hipMemcpyHostToDevice,
                              h->stream));
  }
  /* K-sharded handles connected over the P2P mailbox evaluate the candidates SHARDED BY CANDIDATE ("RMPPI
   * init-eval shards over candidates x samples the same way"): rank r takes candidates [r * ceil(nc / world), ...), writes
   * their costs at their position of the array in every peer's aux mailbox (postAuxKernel) and every rank assembles all
   * nc x ns costs from its own (gatherAuxKernel) — same kernel, same bits as the replicated evaluation, 1 / world of the work.
   * Other exchanges (RCCL, caller-driven) keep evaluating all candidates on every rank: the costs are needed on the HOST of
   * every rank, and a block's T-step chain takes as long for one candidate as for nine. */
  const int world = h->cfg.world_size;
  const bool shard_eval = world > 1 && h->p2p_ready && nc * ns <= kernels::MAILBOX_AUX_FLOATS;
  int c_lo = 0, c_hi = nc;
  if (shard_eval)
  {
    const int chunk = (nc + world - 1) / world;
    c_lo = std::min(nc, h->cfg.rank * chunk);
    c_hi = std::min(nc, c_lo + chunk);
  }
  kernels::InitEvalArgs a{};
  a.dt = h->cfg.dt;
  a.num_timesteps = h->cfg.num_timesteps;
  a.num_eval_rollouts = (c_hi - c_lo) * ns;
  a.samples_per_candidate = ns;
  a.lambda = h->cfg.lambda;
  a.alpha = h->cfg.alpha;
  a.strides_d = (h->low_latency ? reinterpret_cast<const int*>(h->cand_io_dev + (size_t)nc * S) : h->cand_strides_d) + c_lo;
  a.states_d = (h->low_latency ? h->cand_io_dev : h->cand_states_d) + (size_t)c_lo * S;
  // sharded: the slice goes to the device buffer (posted from there), the assembled array to where the host reads it
  float* slice_dev = (shard_eval ? h->cand_costs_d : cand_costs_dev) + (size_t)c_lo * ns;
  a.trajectory_costs_d = slice_dev;
  SamplerLaunchState s{};
  s.num_rollouts_local = h->K_local;
  s.num_rollouts_global = h->cfg.num_rollouts;
  s.rollout_offset = 0;  // eval samples are the GLOBAL rollouts 0 .. samples_per_candidate-1 on every rank
  s.num_timesteps = h->cfg.num_timesteps;
  s.num_distributions = h->D;
  s.control_means_d = h->mean_d;
  s.eps_d = nullptr;
  if (h->noise_source == MPPI_NOISE_INJECTED)
  {
    if (!h->eps_d || h->n_eps_iters <= 0)
      return fail(h, MPPI_ERR_STATE, "noise source is MPPI_NOISE_INJECTED but no noise has been injected");
    s.eps_d = h->eps_d + (size_t)(h->generation % (uint32_t)h->n_eps_iters) * epsFloatsPerIteration(h);
  }
  else if (h->noise_source == MPPI_NOISE_ROCRAND_HOST)
    return fail(h, MPPI_ERR_UNSUPPORTED, "this call draws through the sampler's random-access path: use the Philox or the "
                                         "injected noise source (MPPI_NOISE_ROCRAND_HOST fills the rollout kernel's eps buffer only)");
  s.control_samples_d = nullptr;
  s.seed = h->cfg.seed;
  s.generation = h->generation;
  s.iteration = 0;  // generateSamples(stride, 0, gen) (:596)
  s.optimization_stride = stride;
  s.independent_noise = h->independent_noise ? 1 : 0;
  std::string err;
  if (c_hi > c_lo)
  {
    mppi_status st = h->model->launchInitEval(h->rm_pipeline, a, s, h->stream, err);
    if (st != MPPI_OK)
      return fail(h, st, err);
  }
  h->generation++;
  if (shard_eval)
  {
    const unsigned seq = ++h->aseq;
    const unsigned parity = seq & 1u;
    kernels::AuxTargets t{};
    t.world = world;
    t.seq = seq;
    for (int p = 0; p < world; p++)
    {
      float* aux = h->peer_mbox[p] + h->mbox_aux_off;
      t.peer_aux[p] = aux + (size_t)parity * kernels::MAILBOX_AUX_FLOATS;
      t.peer_flag[p] = reinterpret_cast<unsigned*>(aux + 2 * (size_t)kernels::MAILBOX_AUX_FLOATS) + parity * world + h->cfg.rank;
    }
    hipLaunchKernelGGL(kernels::postAuxKernel, dim3(1), dim3(256), 0, h->stream, slice_dev, c_lo * ns, (c_hi - c_lo) * ns, t);
    const float* my_aux = h->mbox_d + h->mbox_aux_off;
    hipLaunchKernelGGL(kernels::gatherAuxKernel, dim3(1), dim3(256), 0, h->stream,
                       my_aux + (size_t)parity * kernels::MAILBOX_AUX_FLOATS,
                       reinterpret_cast<const unsigned*>(my_aux + 2 * (size_t)kernels::MAILBOX_AUX_FLOATS) + parity * world, world,
                       seq, 200000000ull, nc * ns, cand_costs_dev);
    HIP_TRY(h, hipGetLastError());
  }
  h->rm_cand_costs.resize((size_t)nc * ns);
  if (h->low_latency)
  {
    const unsigned seq = ++h->cand_seq;
    hipLaunchKernelGGL(kernels::raiseFlagKernel, dim3(1), dim3(64), 0, h->stream, h->io_flags_dev + 9, seq);
    HIP_TRY(h, hipGetLastError());
    MPPI_TRY(waitHostFlag(h, 9, seq));
    const float* costs = h->cand_io_h + (size_t)nc * (S + 1);
    std::copy(costs, costs + (size_t)nc * ns, h->rm_cand_costs.begin());
  }
  else
  {
    HIP_TRY(h, hipMemcpyAsync(h->rm_cand_costs.data(), h->cand_costs_d, sizeof(float) * nc * ns, hipMemcpyDeviceToHost,
                              h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  if (shard_eval)
    for (float c : h->rm_cand_costs)
      if (c != c)  // gatherAuxKernel's mark: a peer never delivered its slice (a trajectory cost itself is clamped, never NaN)
        return fail(h, MPPI_ERR_COMM, "Robust MPPI candidate evaluation: a peer's slice of the candidate costs did not arrive");
  rmBestIndex(h);
  h->stats_h.nominal_state_used = h->best_index;
  h->nominal_stride = h->rm_cand_strides[h->best_index];
  std::copy(h->rm_cand_states.begin() + (size_t)h->best_index * S, h->rm_cand_states.begin() + (size_t)(h->best_index + 1) * S,
            h->rm_nominal_state.begin());
  return MPPI_OK;
}

/** reference: robust_mppi_controller.cu:635-755 */
static mppi_status computeControlRobust(mppi_handle h, const float* x0_real, int stride)
{
  const int S = h->S;
  if (!h->gains_set)
    return fail(h, MPPI_ERR_STATE, "Robust MPPI: set the DDP feedback gains first (mppi_set_feedback_gains)");
  if (!h->low_latency)
  {
    HIP_TRY(h, hipMemcpyAsync(h->x0_d, h->rm_nominal_state.data(), sizeof(float) * S, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->x0_d + S, x0_real, sizeof(float) * S, hipMemcpyHostToDevice, h->stream));
    // both importance samplers start from the nominal control (:655-656); later iterations continue from the NEW nominal
    HIP_TRY(h, hipMemcpyAsync(h->mean_d, h->nominal_control_h.data(), sizeof(float) * h->TC, hipMemcpyHostToDevice,
                              h->stream));
  }
  if (h->low_latency)
  {
    /* As computeControlVanilla: inputs and results through host memory mapped into the device, no copy command and no stream
     * synchronisation; the call returns when both control sequences and the statistics are out, while the finalize kernel
     * still re-rolls the two state trajectories (the nominal one is what the NEXT call's candidate states are built from:
     * rmNominalStateAndStride and the trajectory getters wait for it).  AutoRally-NN, T = 150: 179 us of a 622 us call. */
    if (h->traj_pending)
      MPPI_TRY(ensureTrajectories(h));
    float* in = h->io_in_h;
    std::copy(h->rm_nominal_state.begin(), h->rm_nominal_state.begin() + S, in + (h->x0_d - h->in_block_d));
    std::copy(x0_real, x0_real + S, in + (h->x0_d - h->in_block_d) + S);
    float* mean = in + (h->mean_d - h->in_block_d);
    std::copy(h->nominal_control_h.begin(), h->nominal_control_h.end(), mean);
    std::copy(h->nominal_control_h.begin(), h->nominal_control_h.end(), mean + h->TC);
    float* hist = in + (h->history_d - h->in_block_d);
    std::copy(h->nominal_history_h.begin(), h->nominal_history_h.end(), hist);
    std::copy(h->history_h.begin(), h->history_h.end(), hist + 2 * h->C);
    launchIngest(h);
    HIP_TRY(h, hipGetLastError());
    for (int it = 0; it < h->cfg.num_iters; it++)
    {
      if (it > 0)
        HIP_TRY(h, hipMemcpyAsync(h->mean_d + h->TC, h->mean_d, sizeof(float) * h->TC, hipMemcpyDeviceToDevice, h->stream));
      MPPI_TRY(iteration(h, it, stride));
    }
    const int T = h->cfg.num_timesteps;
    kernels::FinalizeArgs a{};
  a.scratch_d = h->fin_scratch_d;
    a.control_in_d = h->mean_d;
    a.history_d = h->history_d;
    a.history_stride = 2 * h->C;
    a.x0_d = h->x0_d;
    a.dt = h->cfg.dt;
    a.num_timesteps = T;
    a.smooth_mask = 3;
    a.constrain_mask = 0;
    a.constrain_mode = 0;
    a.control_out_d = h->io_out_dev + (h->ctrl_out_d - h->out_block_d);
    a.state_out_d = h->io_out_dev + (h->state_out_d - h->out_block_d);
    a.output_out_d = h->io_out_dev + (h->output_out_d - h->out_block_d);
    a.stats_in_d = h->stats_d;
    a.stats_out_d = h->io_out_dev + (h->stats_d - h->out_block_d);
    a.stats_floats = 2 * kernels::STATS_STRIDE;
    a.flags_d = h->io_flags_dev;
    a.seq = ++h->io_seq;
    std::string err;
    const mppi_status st = h->model->launchFinalize(2, a, h->stream, err);
    if (st != MPPI_OK)
      return fail(h, st, err);
    h->out_pin_fresh = false;
    h->results_in_io = true;
    h->traj_pending = true;  // set before the waits: a failing wait must not leave io_out unguarded for the next call
    MPPI_TRY(waitHostFlag(h, 0, h->io_seq));
    MPPI_TRY(waitHostFlag(h, 2, h->io_seq));
    const float* out = h->io_out_h;
    std::copy(out, out + (size_t)T * h->C, h->nominal_control_h.begin());
    std::copy(out + (size_t)T * h->C, out + (size_t)2 * T * h->C, h->control_h.begin());
    parseStats(h, out + (h->stats_d - h->out_block_d));
    h->stats_h_fresh = true;
    if (!allFinite(h->control_h) || !allFinite(h->nominal_control_h))
      return fail(h, MPPI_ERR_NAN, "mppi_compute_control: non-finite value in the control sequence");
    return MPPI_OK;
  }
  for (int it = 0; it < h->cfg.num_iters; it++)
  {
    HIP_TRY(h, hipMemcpyAsync(h->mean_d + h->TC, h->mean_d, sizeof(float) * h->TC, hipMemcpyDeviceToDevice, h->stream));
    MPPI_TRY(iteration(h, it, stride));
  }
  // smooth both with their own history, then the nominal state trajectory from the smoothed nominal control (:732-737)
  std::vector<float>* co[2] = { &h->nominal_control_h, &h->control_h };
  std::vector<float>* so[2] = { &h->nominal_state_h, &h->state_h };
  MPPI_TRY(finalize(h, h->mean_d, /*smooth both*/ 3, /*constrain*/ 0, co, so));
  MPPI_TRY(fetchStats(h));
  if (!allFinite(h->control_h) || !allFinite(h->nominal_control_h) || !allFinite(h->nominal_state_h))
    return fail(h, MPPI_ERR_NAN, "mppi_compute_control: non-finite value in the control or state sequence");
  return MPPI_OK;
}

mppi_status mppi_compute_control(mppi_handle h, const float* x0, int stride)
{
  CHECK_HANDLE_HOST(h);
  if (!x0 || stride < 0)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_compute_control: null state or negative stride");
  for (int i = 0; i < h->S; i++)  // base_plant.hpp:466-470 skips the iteration on a non-finite state; here the call says so
    if (!std::isfinite(x0[i]))
      return fail(h, MPPI_ERR_NAN, "mppi_compute_control: non-finite initial state");
  RoctxRange range("mppi:compute_control");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  h->last_stride = stride;
  if (h->cfg.controller == MPPI_CONTROLLER_TUBE)
    return computeControlTube(h, x0, stride);
  if (h->cfg.controller == MPPI_CONTROLLER_ROBUST)
    return computeControlRobust(h, x0, stride);
  return computeControlVanilla(h, x0, stride);
}

mppi_status mppi_get_control_seq(mppi_handle h, float* u)
{
  CHECK_HANDLE_HOST(h);
  if (!u)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  std::copy(h->control_h.begin(), h->control_h.end(), u);
  return MPPI_OK;
}
mppi_status mppi_get_state_seq(mppi_handle h, float* x)
{
  CHECK_HANDLE_HOST(h);
  if (!x)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  MPPI_TRY(ensureTrajectories(h));
  // RobustMPPI::getTargetStateSeq returns the nominal state trajectory (robust_mppi_controller.cuh:131-134)
  const std::vector<float>& src = h->cfg.controller == MPPI_CONTROLLER_ROBUST ? h->nominal_state_h : h->state_h;
  std::copy(src.begin(), src.end(), x);
  return MPPI_OK;
}
mppi_status mppi_get_output_seq(mppi_handle h, float* y)
{
  CHECK_HANDLE_HOST(h);
  if (!y)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_get_output_seq: null");
  // system 0 of the last finalize pass is the trajectory mppi_get_state_seq reports (real system for Vanilla / Tube, the
  // nominal one for Robust MPPI)
  if (h->results_in_io)
  {  // the last finalize pass wrote its outputs to the device-mapped host block
    MPPI_TRY(ensureTrajectories(h));
    const float* src = h->io_out_h + (h->output_out_d - h->out_block_d);
    std::copy(src, src + (size_t)h->cfg.num_timesteps * h->O, y);
    return MPPI_OK;
  }
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  joinSideStream(h);
  HIP_TRY(h, hipMemcpyAsync(y, h->output_out_d, sizeof(float) * h->cfg.num_timesteps * h->O, hipMemcpyDeviceToHost,
                            h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

mppi_status mppi_get_nominal_control_seq(mppi_handle h, float* u)
{
  CHECK_HANDLE_HOST(h);
  if (!u)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  if (h->D != 2)
    return fail(h, MPPI_ERR_STATE, "no nominal system in this controller");
  std::copy(h->nominal_control_h.begin(), h->nominal_control_h.end(), u);
  return MPPI_OK;
}
mppi_status mppi_get_nominal_state_seq(mppi_handle h, float* x)
{
  CHECK_HANDLE_HOST(h);
  if (!x)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  if (h->D != 2)
    return fail(h, MPPI_ERR_STATE, "no nominal system in this controller");
  MPPI_TRY(ensureTrajectories(h));
  std::copy(h->nominal_state_h.begin(), h->nominal_state_h.end(), x);
  return MPPI_OK;
}

static void saveControlHistory(int steps, const std::vector<float>& u, std::vector<float>& hist, int C);
static void slideSequence(std::vector<float>& u, int T, int C, int steps, const float* zero, const float* scale);
static mppi_status rmNominalStateAndStride(mppi_handle h, const float* state, int stride);
static mppi_status rmNominalStateTrajectory(mppi_handle h);

/** reference: controllers/controller.cuh:602-615 */
static void saveControlHistory(int steps, const std::vector<float>& u, std::vector<float>& hist, int C)
{
  if (steps == 1)
  {
    for (int c = 0; c < C; c++)
    {
      hist[c] = hist[C + c];
      hist[C + c] = u[c];
    }
  }
  else if (steps >= 2)
  {
    for (int c = 0; c < C; c++)
    {
      hist[c] = u[(size_t)(steps - 2) * C + c];
      hist[C + c] = u[(size_t)(steps - 1) * C + c];
    }
  }
}
/** reference: controllers/controller.cuh:588-600 */
static void slideSequence(std::vector<float>& u, int T, int C, int steps, const float* zero, const float* scale)
{
  for (int i = 0; i < T; i++)
  {
    const int ind = std::min(i + steps, T - 1);
    for (int c = 0; c < C; c++)
    {
      u[(size_t)i * C + c] = u[(size_t)ind * C + c];
      if (i + steps > T - 1)
        u[(size_t)i * C + c] = (u[(size_t)ind * C + c] - zero[c]) * scale[c] + zero[c];
    }
  }
}

/** x <- one model step under u (u <- the clamped control when `enforce`), in host memory mapped into the device: no copy
 *  command; the host spins on a flag raised behind the kernel (MPPI_AMD_NO_SPIN=1: a stream synchronisation instead) */
mppi_status modelStepInPlace(mppi_handle h, float* x, float* u, float dt, int enforce)
{
  std::copy(x, x + h->S, h->step_pin_h);
  std::copy(u, u + h->C, h->step_pin_h + h->S);
  std::string err;
  mppi_status st = h->model->launchModelStep(h->step_pin_dev, h->step_pin_dev + h->S, dt, enforce, h->stream, err);
  if (st != MPPI_OK)
    return fail(h, st, err);
  if (h->low_latency)
  {
    const unsigned seq = ++h->step_seq;
    hipLaunchKernelGGL(kernels::raiseFlagKernel, dim3(1), dim3(64), 0, h->stream, h->io_flags_dev + 8, seq);
    HIP_TRY(h, hipGetLastError());
    MPPI_TRY(waitHostFlag(h, 8, seq));
  }
  else
  {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  std::copy(h->step_pin_h, h->step_pin_h + h->S, x);
  std::copy(h->step_pin_h + h->S, h->step_pin_h + h->S + h->C, u);
  return MPPI_OK;
}

mppi_status mppi_slide(mppi_handle h, int steps)
{
  CHECK_HANDLE_HOST(h);
  const int T = h->cfg.num_timesteps, C = h->C;
  if (steps < 0 || steps > T)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_slide: steps out of range");
  if (h->cfg.controller == MPPI_CONTROLLER_ROBUST)
    return MPPI_OK;  // slideControlSequence is empty there (robust_mppi_controller.cuh:178): the slide is part of
                     // updateImportanceSamplingControl
  std::vector<float> zero(C);
  h->model->getZeroControl(zero.data());
  if (h->cfg.controller == MPPI_CONTROLLER_TUBE)
  {
    // tube_mppi_controller.cu:312-323: updateNominalState(nominal_control.col(0)) — one in-place model step, no clamp
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    // single launch: the step queues behind the last finalize kernel anyway, and a later trajectory read must not find io_out
    // rewritten — take the trajectories now.  Split hand-over: the step runs beside the trajectory phase
    if (!h->split_finalize)
      MPPI_TRY(ensureTrajectories(h));
    std::vector<float> u0(h->nominal_control_h.begin(), h->nominal_control_h.begin() + C);
    MPPI_TRY(modelStepInPlace(h, h->tube_x_h.data(), u0.data(), h->cfg.dt, 0));
    saveControlHistory(steps, h->nominal_control_h, h->history_h, C);
    slideSequence(h->nominal_control_h, T, C, steps, zero.data(), h->slide_scale_h.data());
    slideSequence(h->control_h, T, C, steps, zero.data(), h->slide_scale_h.data());
    return MPPI_OK;
  }
  saveControlHistory(steps, h->control_h, h->history_h, C);
  slideSequence(h->control_h, T, C, steps, zero.data(), h->slide_scale_h.data());
  return MPPI_OK;
}


/* ---------------------------------------------------------------- Robust MPPI API -------------------------------- */
mppi_status mppi_set_rmppi_params(mppi_handle h, float value_function_threshold, int num_candidates,
                                  int samples_per_candidate)
{
  CHECK_HANDLE(h);
  if (h->cfg.controller != MPPI_CONTROLLER_ROBUST)
    return fail(h, MPPI_ERR_STATE, "mppi_set_rmppi_params: the handle is not a Robust MPPI controller");
  // updateNumCandidates (robust_mppi_controller.cu:414-448): odd, >= 3, candidates * samples <= NUM_ROLLOUTS
  if (num_candidates < 3)
    return fail(h, MPPI_ERR_INVALID_ARG, "ERROR: number of candidates must be greater or equal to 3");
  if (num_candidates % 2 == 0)
    return fail(h, MPPI_ERR_INVALID_ARG, "ERROR: number of candidates must be odd");
  if (samples_per_candidate <= 0 || (long long)num_candidates * samples_per_candidate > h->cfg.num_rollouts)
    return fail(h, MPPI_ERR_INVALID_ARG, "ERROR: (number of candidates) * (SAMPLES_PER_CANDIDATE) cannot exceed NUM_ROLLOUTS");
  h->value_function_threshold = value_function_threshold;
  h->num_candidates = num_candidates;
  h->samples_per_candidate = samples_per_candidate;
  return MPPI_OK;
}

mppi_status mppi_set_feedback_gains(mppi_handle h, const float* gains, int accumulate_all_states)
{
  CHECK_HANDLE(h);
  if (!gains)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_feedback_gains: null");
  if (h->cfg.controller != MPPI_CONTROLLER_ROBUST)
    return fail(h, MPPI_ERR_STATE, "mppi_set_feedback_gains: the handle is not a Robust MPPI controller");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  std::string err;
  mppi_status st = h->model->setFeedbackGains(gains, h->cfg.num_timesteps, accumulate_all_states != 0, h->stream, err);
  if (st != MPPI_OK)
    return fail(h, st, err);
  h->gains_set = true;
  h->fb_accumulate_all = accumulate_all_states != 0;
  return MPPI_OK;
}

/** reference: robust_mppi_controller.cu:548-568 (updateImportanceSamplingControl); the DDP gain computation at its end
 *  (computeNominalFeedbackGains) is the caller's: mppi_set_feedback_gains */
mppi_status mppi_update_importance_sampling_control(mppi_handle h, const float* state, int stride)
{
  CHECK_HANDLE(h);
  if (!state || stride < 0)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_update_importance_sampling_control: null state or negative stride");
  if (h->cfg.controller != MPPI_CONTROLLER_ROBUST)
    return fail(h, MPPI_ERR_STATE, "mppi_update_importance_sampling_control: the handle is not a Robust MPPI controller");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  MPPI_TRY(ensureTrajectories(h));  // the nominal state trajectory of the last mppi_compute_control (low-latency hand-over)
  const int T = h->cfg.num_timesteps, C = h->C;
  h->real_stride = stride;
  MPPI_TRY(rmNominalStateAndStride(h, state, stride));
  saveControlHistory(h->nominal_stride, h->nominal_control_h, h->nominal_history_h, C);
  saveControlHistory(h->real_stride, h->control_h, h->history_h, C);
  std::vector<float> zero(C);
  h->model->getZeroControl(zero.data());
  slideSequence(h->nominal_control_h, T, C, h->nominal_stride, zero.data(), h->slide_scale_h.data());
  return rmNominalStateTrajectory(h);
}

mppi_status mppi_get_rmppi_state(mppi_handle h, float* nominal_state, int* best_index, int* nominal_stride,
                                 float* candidate_free_energy)
{
  CHECK_HANDLE(h);
  if (h->cfg.controller != MPPI_CONTROLLER_ROBUST)
    return fail(h, MPPI_ERR_STATE, "mppi_get_rmppi_state: the handle is not a Robust MPPI controller");
  if (nominal_state)
    std::copy(h->rm_nominal_state.begin(), h->rm_nominal_state.end(), nominal_state);
  if (best_index)
    *best_index = h->best_index;
  if (nominal_stride)
    *nominal_stride = h->nominal_stride;
  if (candidate_free_energy)
    for (int i = 0; i < h->num_candidates; i++)
      candidate_free_energy[i] = i < (int)h->rm_cand_free_energy.size() ? h->rm_cand_free_energy[i] : 0.0f;
  return MPPI_OK;
}

mppi_status mppi_get_costs(mppi_handle h, float* costs)
{
  CHECK_HANDLE(h);
  if (!costs)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(costs, h->costs_d, sizeof(float) * h->D * h->K_local, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}
mppi_status mppi_get_stats(mppi_handle h, mppi_stats* out)
{
  CHECK_HANDLE_HOST(h);
  if (!out)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  if (!h->stats_h_fresh)
  {  // a read from the device: behind everything, the side stream included
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    joinSideStream(h);
  }
  const int used = h->stats_h.nominal_state_used;
  MPPI_TRY(fetchStats(h));
  h->stats_h.nominal_state_used = used;
  *out = h->stats_h;
  if (h->exchange_failed)
    return fail(h, MPPI_ERR_COMM, "P2P exchange: a peer's record did not arrive within 2 s; the merge was abandoned");
  return MPPI_OK;
}
mppi_status mppi_get_sampled_controls(mppi_handle h, float* v)
{
  CHECK_HANDLE(h);
  if (!v)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  if (!h->samples_d)
    return fail(h, MPPI_ERR_STATE, "mppi_get_sampled_controls: handle was created without save_samples");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(v, h->samples_d, sizeof(float) * h->D * h->K_local * h->TC, hipMemcpyDeviceToHost,
                            h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

mppi_status mppi_sample_noise(mppi_handle h, int optimization_stride, float* eps_out)
{
  CHECK_HANDLE(h);
  if (!eps_out || optimization_stride < 0)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_sample_noise: bad arguments");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  SamplerLaunchState s{};
  s.num_rollouts_local = h->K_local;
  s.num_rollouts_global = h->cfg.num_rollouts;
  s.rollout_offset = h->K_offset;
  s.num_timesteps = h->cfg.num_timesteps;
  s.num_distributions = 1;
  s.control_means_d = h->mean_d;
  s.eps_d = nullptr;
  if (h->noise_source == MPPI_NOISE_INJECTED)
  {
    if (!h->eps_d || h->n_eps_iters <= 0)
      return fail(h, MPPI_ERR_STATE, "noise source is MPPI_NOISE_INJECTED but no noise has been injected");
    s.eps_d = h->eps_d + (size_t)(h->generation % (uint32_t)h->n_eps_iters) * epsFloatsPerIteration(h);
  }
  else if (h->noise_source == MPPI_NOISE_ROCRAND_HOST)
    return fail(h, MPPI_ERR_UNSUPPORTED, "this call draws through the sampler's random-access path: use the Philox or the "
                                         "injected noise source (MPPI_NOISE_ROCRAND_HOST fills the rollout kernel's eps buffer only)");
  s.control_samples_d = nullptr;
  s.seed = h->cfg.seed;
  s.generation = h->generation;
  s.iteration = 0;
  s.optimization_stride = optimization_stride;
  s.independent_noise = h->independent_noise ? 1 : 0;
  const size_t n = (size_t)h->K_local * h->TC;
  float* out_d = nullptr;
  HIP_TRY(h, hipMalloc((void**)&out_d, n * sizeof(float)));
  std::string err;
  mppi_status st = h->model->launchNoiseDump(s, out_d, h->stream, err);
  hipError_t e = hipSuccess;
  if (st == MPPI_OK)
  {
    e = hipMemcpyAsync(eps_out, out_d, n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess)
      e = hipStreamSynchronize(h->stream);
  }
  (void)hipFree(out_d);
  if (st != MPPI_OK)
    return fail(h, st, err);
  if (e != hipSuccess)
    return fail(h, MPPI_ERR_HIP, std::string("mppi_sample_noise: ") + hipGetErrorString(e));
  return MPPI_OK;
}
