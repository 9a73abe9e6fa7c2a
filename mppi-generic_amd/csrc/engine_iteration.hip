/**
 * engine_iteration.hip — one optimisation iteration: rollout launch, merges, in-iteration exchange, post-processing pass.
 * Part of the implementation of include/mppi_amd.h; see engine_internal.hpp for how the engine is divided and
 * engine_core.hip for the references its logic follows.
 */
#include "engine_internal.hpp"

/* ---------------------------------------------------------------- internals -------------------------------------- */
static kernels::CombineArgs combineArgs(mppi_handle h, const float* records, int num_records, int finalize, float* record_out,
                                        int k_total, bool world_major = false, const unsigned* wait_flags = nullptr,
                                        unsigned wait_seq = 0, const kernels::PostTargets* post = nullptr)
{
  kernels::CombineArgs a{};
  if (post)
    a.post = *post;
  a.records_d = records;
  a.num_records = num_records;
  // block records of the rollout kernel: [D][num_blocks][PS]; records gathered from the ranks: [world][D][PS]
  a.z_stride = world_major ? h->PS : num_records * h->PS;
  a.rec_stride = world_major ? h->D * h->PS : h->PS;
  a.TC = h->TC;
  a.PS = h->PS;
  a.lambda = h->cfg.lambda;
  a.num_rollouts_total = k_total;
  a.finalize = finalize;
  a.mean_out_d = h->mean_d;
  a.record_out_d = record_out;
  a.stats_out_d = h->stats_d;
  a.wait_flags_d = wait_flags;
  a.wait_seq = wait_seq;
  a.wait_limit_ticks = 200000000ull;  // 2 s of the 100 MHz wall clock
  return a;
}

mppi_status launchCombine(mppi_handle h, const float* records, int num_records, int finalize, float* record_out,
                                 int k_total, bool world_major, const unsigned* wait_flags,
                                 unsigned wait_seq, const kernels::PostTargets* post)
{
  RoctxRange range(finalize ? "mppi:merge" : "mppi:merge_local");
  h->n_merge_launches++;
  const kernels::CombineArgs a =
      combineArgs(h, records, num_records, finalize, record_out, k_total, world_major, wait_flags, wait_seq, post);
  hipLaunchKernelGGL(kernels::combineKernel, dim3(h->D, kernels::combineGridY(h->TC)), dim3(kernels::MERGE_THREADS), 0,
                     h->stream, a);
  HIP_TRY(h, hipGetLastError());
  return MPPI_OK;
}

/** may the NEXT rollout launch merge the previous launch's records itself (rolloutPipelineKernel STREAM_MERGE)? */
bool streamMergeApplies(const mppi_handle_s* h)
{
  return h->stream_merge_enabled && h->pipeline && h->bz == 1 && h->D == 1 && h->by == 1 && h->bx == 64 && !h->rows_in_hbm &&
         !exchangeActive(h) && h->noise_source == MPPI_NOISE_PHILOX_FUSED && h->reduction_mode == MPPI_REDUCTION_FUSED &&
         !tsallisActive(h) && h->cfg.controller != MPPI_CONTROLLER_ROBUST && (h->TC & 3) == 0 && h->num_blocks <= 256 &&
         h->model->supportsStreamedMerge();
}
/** the transposed copy that belongs to a record buffer (partials_d / partials_alt_d): behind its [D][num_blocks][PS] records */
float* recordsTransposed(const mppi_handle_s* h, float* records)
{
  return records + (size_t)h->D * h->num_blocks * h->PS;
}
/** the records of the last rollout launch are still un-merged: merge them now (combineKernel -> mean_d, stats_d) */
mppi_status flushMerge(mppi_handle h)
{
  if (!h->pending_records_d)
    return MPPI_OK;
  const float* rec = h->pending_records_d;
  h->pending_records_d = nullptr;
  return launchCombine(h, rec, h->num_blocks, 1, nullptr, h->cfg.num_rollouts, false, nullptr, 0, nullptr);
}

mppi_status launchRollout(mppi_handle h, int iteration, int stride)
{
  RoctxRange range("mppi:rollout");
  h->n_rollout_launches++;
  kernels::RolloutArgs a{};
  a.dt = h->cfg.dt;
  a.num_timesteps = h->cfg.num_timesteps;
  a.num_rollouts = h->K_local;
  a.lambda = h->cfg.lambda;
  a.alpha = h->cfg.alpha;
  a.init_x_d = h->x0_src_d ? h->x0_src_d : h->x0_d;
  a.trajectory_costs_d = h->costs_d;
  a.partials_d = h->partials_d;
  a.save_samples = h->samples_d ? 1 : 0;
  a.prev_records_d = nullptr;
  a.prev_num_records = 0;
  a.prev_records_t_d = nullptr;
  // the launches whose records the next launch may merge itself also write the transposed copy its sampler waves read
  a.records_t_d = streamMergeApplies(h) ? recordsTransposed(h, h->partials_d) : nullptr;
  if (h->pending_records_d)
  {
    if (!streamMergeApplies(h))
      MPPI_TRY(flushMerge(h));  // (a setting changed between two launches: merge the pending records the ordinary way)
    else
    {
      a.prev_records_d = h->pending_records_d;
      a.prev_records_t_d = recordsTransposed(h, const_cast<float*>(h->pending_records_d));
      a.prev_num_records = h->num_blocks;
      h->pending_records_d = nullptr;
    }
  }
  SamplerLaunchState s{};
  s.num_rollouts_local = h->K_local;
  s.num_rollouts_global = h->cfg.num_rollouts;
  s.rollout_offset = h->K_offset;
  s.num_timesteps = h->cfg.num_timesteps;
  s.num_distributions = h->D;
  s.control_means_d = h->mean_src_d ? const_cast<float*>(h->mean_src_d) : h->mean_d;  // (the kernels only read it)
  h->mean_src_d = nullptr;  // one launch only: later iterations read what the merge wrote to mean_d
  s.eps_d = nullptr;
  if (h->noise_source == MPPI_NOISE_INJECTED)
  {
    if (!h->eps_d || h->n_eps_iters <= 0)
      return fail(h, MPPI_ERR_STATE, "noise source is MPPI_NOISE_INJECTED but no noise has been injected");
    s.eps_d = h->eps_d + (size_t)(h->generation % (uint32_t)h->n_eps_iters) * epsFloatsPerIteration(h);
  }
  else if (h->noise_source == MPPI_NOISE_ROCRAND_HOST)
  {
    MPPI_TRY(rocrandFill(h));
    s.eps_d = h->rocrand_eps_d;
  }
  s.control_samples_d = h->samples_d;
  s.seed = h->cfg.seed;
  s.generation = h->generation;
  s.iteration = iteration;
  s.optimization_stride = stride;
  s.independent_noise = h->independent_noise ? 1 : 0;
  std::string err;
  mppi_status st;
  if (h->cfg.controller == MPPI_CONTROLLER_ROBUST)
  {
    kernels::RMPPIArgs ra{};
    ra.base = a;
    ra.value_function_threshold = h->value_function_threshold;
    st = h->model->launchRMPPI(h->bx, h->rm_pipeline, ra, s, h->stream, err);
  }
  else
  {
    st = h->model->launchRollout(h->bx, h->by, h->bz, h->pipeline, a, s, h->stream, err);
  }
  if (st != MPPI_OK)
    return fail(h, st, err);
  h->generation++;
  h->out_pin_fresh = false;
  h->stats_h_fresh = false;
  return MPPI_OK;
}

nccl_allgather_fn g_ncclAllGather = nullptr;

/** where this rank's merged record of exchange `seq` goes in every peer's mailbox */
static kernels::PostTargets p2pTargets(mppi_handle h, unsigned seq)
{
  kernels::PostTargets t{};
  const int world = h->cfg.world_size;
  const size_t dps = (size_t)h->D * h->PS;
  const unsigned parity = seq & 1u;
  for (int p = 0; p < world; p++)
  {
    float* base = h->peer_mbox[p];
    t.peer_slot[p] = base + ((size_t)parity * world + h->cfg.rank) * dps;
    t.peer_flag[p] = reinterpret_cast<unsigned*>(base + (size_t)2 * world * dps) + parity * world + h->cfg.rank;
  }
  t.world = world;
  t.seq = seq;
  // the ticket counter sits behind the flags of this rank's own mailbox
  t.ticket_d = reinterpret_cast<unsigned*>(h->mbox_d + (size_t)2 * world * dps) + 2 * world;
  return t;
}


/** the reference's own last stage, operation for operation (exact_reduce_kernels.hpp): global rho -> weights -> eta in
 *  double, index order -> per-rollout weight / eta, cells of sum_strides rollouts, cells in order */
static mppi_status launchExactReduction(mppi_handle h)
{
  RoctxRange range("mppi:reduce_reference_order");
  h->n_merge_launches++;
  kernels::ExactWeightsArgs a{};
  a.num_rollouts = h->K_local;
  a.costs_d = h->costs_d;
  a.weights_d = h->exact_weights_d;
  a.stats_out_d = h->stats_d;
  a.lambda = h->cfg.lambda;
  a.lambda_inv = (float)(1.0 / (double)h->cfg.lambda);
  a.tsallis_gamma = tsallisActive(h) ? h->tsallis_gamma : 0.0f;
  a.tsallis_r = tsallisActive(h) ? h->tsallis_r : 0.0f;
  hipLaunchKernelGGL(kernels::exactWeightsKernel, dim3(h->D), dim3(kernels::COMBINE_THREADS),
                     kernels::EXACT_WEIGHTS_LDS_BYTES, h->stream, a);
  const int cells = (h->K_local - 1) / h->sum_strides + 1;
  const dim3 grid((h->TC + 63) / 64, (cells + kernels::COMBINE_THREADS / 64 - 1) / (kernels::COMBINE_THREADS / 64), h->D);
  if (h->reduction_mode == MPPI_REDUCTION_REFERENCE_ORDER_FMA)
    hipLaunchKernelGGL(kernels::exactReductionCellsKernel<1>, grid, dim3(kernels::COMBINE_THREADS), 0, h->stream,
                       h->exact_weights_d, h->samples_d, h->stats_d, h->TC, h->K_local, h->sum_strides, cells,
                       h->exact_inter_d);
  else
    hipLaunchKernelGGL(kernels::exactReductionCellsKernel<0>, grid, dim3(kernels::COMBINE_THREADS), 0, h->stream,
                       h->exact_weights_d, h->samples_d, h->stats_d, h->TC, h->K_local, h->sum_strides, cells,
                       h->exact_inter_d);
  hipLaunchKernelGGL(kernels::exactReductionFinalKernel, dim3((h->TC + 63) / 64, h->D), dim3(64), 0, h->stream,
                     h->exact_inter_d, h->TC, cells, h->mean_d);
  HIP_TRY(h, hipGetLastError());
  return MPPI_OK;
}

mppi_status iterationLocal(mppi_handle h, int iteration, int stride)
{
  MPPI_TRY(launchRollout(h, iteration, stride));
  if (h->reduction_mode != MPPI_REDUCTION_FUSED)
    return launchExactReduction(h);
  if (tsallisActive(h) && exchangeActive(h))
    return fail(h, MPPI_ERR_UNSUPPORTED, "Tsallis weights need two exchanges per iteration: on a K-sharded handle use "
                                         "mppi_optimize / mppi_compute_control over the P2P mailbox or RCCL, not the caller-driven "
                                         "mppi_iteration_local / mppi_iteration_merge pair");
  if (tsallisActive(h))
  {  // global baseline -> Tsallis weights -> weighted mean of the dumped samples (reduce_kernels.hpp)
    hipLaunchKernelGGL(kernels::tsallisWeightsKernel, dim3(1), dim3(kernels::COMBINE_THREADS), 0, h->stream, h->K_local,
                       h->costs_d, h->tsallis_gamma, h->tsallis_r, h->cfg.lambda, h->tsallis_weights_d, h->stats_d);
    hipLaunchKernelGGL(kernels::tsallisMeanKernel, dim3((h->TC + kernels::COMBINE_COLS - 1) / kernels::COMBINE_COLS),
                       dim3(kernels::COMBINE_THREADS), 0, h->stream, h->tsallis_weights_d, h->samples_d, h->stats_d, h->TC,
                       h->K_local, h->mean_d);
    HIP_TRY(h, hipGetLastError());
    return MPPI_OK;
  }
  if (!exchangeActive(h))
  {
    if (streamMergeApplies(h))
    {  // leave the records to the next rollout launch (or to flushMerge) and write the next ones into the other buffer
      h->pending_records_d = h->partials_d;
      std::swap(h->partials_d, h->partials_alt_d);
      return MPPI_OK;
    }
    return launchCombine(h, h->partials_d, h->num_blocks, 1, nullptr, h->cfg.num_rollouts);
  }
  return launchCombine(h, h->partials_d, h->num_blocks, 0, h->send_d, h->K_local);
}

mppi_status iterationMerge(mppi_handle h)
{
  if (!exchangeActive(h))
    return MPPI_OK;
  // recv_d is [world][D][PS]: combineKernel walks it with world-major strides (no regroup launch)
  return launchCombine(h, h->recv_d, h->cfg.world_size, 1, nullptr, h->cfg.num_rollouts, true);
}

/** this rank's mailbox half of exchange `seq`: the world's records and the flags the peers raise */
static inline void p2pInbox(mppi_handle h, unsigned seq, const float** records, const unsigned** flags)
{
  const int world = h->cfg.world_size;
  const size_t dps = (size_t)h->D * h->PS;
  const unsigned parity = seq & 1u;
  *records = h->mbox_d + (size_t)parity * world * dps;
  *flags = reinterpret_cast<const unsigned*>(h->mbox_d + (size_t)2 * world * dps) + parity * world;
}

/** P2P exchange: the local merge has posted this rank's record (iterationLocal); the global merge waits for the peers' flags */
static mppi_status iterationMergeP2P(mppi_handle h)
{
  const float* records;
  const unsigned* flags;
  p2pInbox(h, h->xseq, &records, &flags);
  return launchCombine(h, records, h->cfg.world_size, 1, nullptr, h->cfg.num_rollouts, true, flags, h->xseq);
}

/** rollout kernel done: local merge + post + wait + global merge in ONE launch (combineShardedKernel) */
static mppi_status launchCombineSharded(mppi_handle h)
{
  RoctxRange range("mppi:merge_sharded");
  const unsigned seq = ++h->xseq;
  const kernels::PostTargets t = p2pTargets(h, seq);
  // The fused form's waves wait for the launch's own last ticket: a grid that cannot be resident at once takes two launches.
  // The bound is THIS device's: its CU count x the occupancy the runtime reports for the kernel, halved — rollout kernels of
  // other handles / streams may hold slots — and never above the constant the kernel was reviewed for (round-5 advice: the
  // constant alone assumed 256 free CUs).
  if (h->combine_sharded_max_blocks < 0)
  {
    int per_cu = 0, cus = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernels::combineShardedKernel, kernels::MERGE_THREADS, 0) == hipSuccess &&
        hipGetDeviceProperties(&prop, h->cfg.device) == hipSuccess)
      cus = prop.multiProcessorCount;
    (void)hipGetLastError();
    const long bound = (long)per_cu * cus / 2;
    h->combine_sharded_max_blocks = (int)std::min<long>(kernels::COMBINE_SHARDED_MAX_BLOCKS, bound > 0 ? bound : 0);
  }
  if (h->D * kernels::combineGridY(h->TC) > h->combine_sharded_max_blocks)
  {
    MPPI_TRY(launchCombine(h, h->partials_d, h->num_blocks, 0, h->send_d, h->K_local, false, nullptr, 0, &t));
    return iterationMergeP2P(h);
  }
  h->n_merge_launches++;
  const float* records;
  const unsigned* flags;
  p2pInbox(h, seq, &records, &flags);
  const kernels::CombineArgs loc = combineArgs(h, h->partials_d, h->num_blocks, 0, h->send_d, h->K_local, false, nullptr, 0, &t);
  const kernels::CombineArgs glob = combineArgs(h, records, h->cfg.world_size, 1, nullptr, h->cfg.num_rollouts, true, flags, seq);
  hipLaunchKernelGGL(kernels::combineShardedKernel, dim3(h->D, kernels::combineGridY(h->TC)), dim3(kernels::MERGE_THREADS), 0,
                     h->stream, loc, glob);
  HIP_TRY(h, hipGetLastError());
  return MPPI_OK;
}

static mppi_status exchangeAllGather(mppi_handle h)
{
  if (!h->comm || !g_ncclAllGather)
    return fail(h, MPPI_ERR_STATE,
                "world_size > 1: call mppi_p2p_connect / mppi_comm_init_rccl first, or drive the exchange yourself with "
                "mppi_iteration_local / mppi_get_exchange_buffers / mppi_iteration_merge");
  const int rc = g_ncclAllGather(h->send_d, h->recv_d, (size_t)h->D * h->PS, /*ncclFloat32*/ 7, h->comm, h->stream);
  if (rc != 0)
    return fail(h, MPPI_ERR_COMM, "ncclAllGather failed with code " + std::to_string(rc));
  return MPPI_OK;
}

/**
 * ColoredMPPI's Tsallis weights on a K-sharded handle (reference: core/mppi_common.cu:968-985 on all K rollouts).  The weights
 * are not shift-invariant, so the GLOBAL baseline has to exist before any of them: two exchanges per iteration —
 *   1. the ranks' minima (the local merge's record; only its tail is used),
 *   2. {sum w v | rho, sum w, sum w^2} of every rank under that common baseline; the merge then rescales by exp(0) = 1.
 * Over the P2P mailbox (two sequence numbers per iteration) or RCCL; the caller-driven exchange has one hop per iteration and
 * does not offer it.
 */
static mppi_status iterationShardedTsallis(mppi_handle h, int iteration, int stride)
{
  const int world = h->cfg.world_size;
  MPPI_TRY(launchRollout(h, iteration, stride));
  const float* peer_records = nullptr;
  const unsigned* flags = nullptr;
  unsigned seq1 = 0;
  if (h->p2p_ready)
  {
    seq1 = ++h->xseq;
    const kernels::PostTargets t = p2pTargets(h, seq1);
    MPPI_TRY(launchCombine(h, h->partials_d, h->num_blocks, 0, h->send_d, h->K_local, false, nullptr, 0, &t));
    p2pInbox(h, seq1, &peer_records, &flags);
  }
  else
  {
    MPPI_TRY(launchCombine(h, h->partials_d, h->num_blocks, 0, h->send_d, h->K_local));
    MPPI_TRY(exchangeAllGather(h));
    peer_records = h->recv_d;
  }
  float* rec = h->tsallis_record_d;
  hipLaunchKernelGGL(kernels::tsallisWeightsKernel, dim3(1), dim3(kernels::COMBINE_THREADS), 0, h->stream, h->K_local,
                     h->costs_d, h->tsallis_gamma, h->tsallis_r, h->cfg.lambda, h->tsallis_weights_d, h->stats_d, peer_records,
                     world, h->D * h->PS, h->TC, flags, seq1, 200000000ull, rec + h->TC);
  hipLaunchKernelGGL(kernels::tsallisMeanKernel, dim3((h->TC + kernels::COMBINE_COLS - 1) / kernels::COMBINE_COLS),
                     dim3(kernels::COMBINE_THREADS), 0, h->stream, h->tsallis_weights_d, h->samples_d, h->stats_d, h->TC,
                     h->K_local, rec, 0);
  HIP_TRY(h, hipGetLastError());
  if (h->p2p_ready)
  {
    const unsigned seq2 = ++h->xseq;
    const kernels::PostTargets t = p2pTargets(h, seq2);
    MPPI_TRY(launchCombine(h, rec, 1, 0, h->send_d, h->K_local, false, nullptr, 0, &t));
    return iterationMergeP2P(h);
  }
  MPPI_TRY(launchCombine(h, rec, 1, 0, h->send_d, h->K_local));
  MPPI_TRY(exchangeAllGather(h));
  return iterationMerge(h);
}

mppi_status iteration(mppi_handle h, int it, int stride)
{
  if (exchangeActive(h) && tsallisActive(h))
  {
    if (!h->p2p_ready && !h->comm)
      return fail(h, MPPI_ERR_STATE, "Tsallis weights on a K-sharded handle need the P2P mailbox or the RCCL communicator "
                                     "(two exchanges per iteration): mppi_p2p_connect / mppi_comm_init_rccl");
    return iterationShardedTsallis(h, it, stride);
  }
  if (exchangeActive(h) && h->p2p_ready)
  {  // two launches: rollout, then merge + post + wait + merge in one kernel
    MPPI_TRY(launchRollout(h, it, stride));
    return launchCombineSharded(h);
  }
  MPPI_TRY(iterationLocal(h, it, stride));
  if (exchangeActive(h))
  {
    MPPI_TRY(exchangeAllGather(h));
    MPPI_TRY(iterationMerge(h));
  }
  return MPPI_OK;
}

mppi_status fetchStats(mppi_handle h)
{
  if (h->stats_h_fresh)  // parsed at the hand-over of the last mppi_compute_control, no launch since
    return MPPI_OK;
  if (h->out_pin_fresh)
  {  // the last finalize pass brought the statistics along
    parseStats(h, h->out_pin_h + (h->stats_d - h->out_block_d));
    return MPPI_OK;
  }
  float st[2 * kernels::STATS_STRIDE] = { 0 };
  HIP_TRY(h, hipMemcpyAsync(st, h->stats_d, sizeof(float) * h->D * kernels::STATS_STRIDE, hipMemcpyDeviceToHost,
                            h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  parseStats(h, st);
  return MPPI_OK;
}

bool allFinite(const std::vector<float>& v)
{
  for (float f : v)
    if (!std::isfinite(f))
      return false;
  return true;
}

/** smoothing / state trajectories / constraints for the D systems in ctrl_in_d, results to the host vectors */
mppi_status finalize(mppi_handle h, const float* ctrl_in_d, int smooth_mask, int constrain_mask,
                            std::vector<float>* ctrl_out[2], std::vector<float>* state_out[2], int num_systems)
{
  RoctxRange range("mppi:finalize");
  const int T = h->cfg.num_timesteps;
  kernels::FinalizeArgs a{};
  a.scratch_d = h->fin_scratch_d;
  // the control history goes up through its slice of the pinned input block (one small asynchronous copy)
  float* hist_pin = h->in_pin_h + (h->history_d - h->in_block_d);
  if (h->cfg.controller == MPPI_CONTROLLER_ROBUST)
  {  // system 0 (nominal) smooths with nominal_control_history_, system 1 (real) with control_history_
    std::copy(h->nominal_history_h.begin(), h->nominal_history_h.end(), hist_pin);
    std::copy(h->history_h.begin(), h->history_h.end(), hist_pin + 2 * h->C);
    a.history_stride = 2 * h->C;
  }
  else
  {
    std::copy(h->history_h.begin(), h->history_h.end(), hist_pin);
    a.history_stride = 0;
  }
  HIP_TRY(h, hipMemcpyAsync(h->history_d, hist_pin, sizeof(float) * 4 * h->C, hipMemcpyHostToDevice, h->stream));
  a.control_in_d = ctrl_in_d;
  a.history_d = h->history_d;
  a.x0_d = h->x0_d;
  a.control_out_d = h->ctrl_out_d;
  a.state_out_d = h->state_out_d;
  a.output_out_d = h->output_out_d;
  a.dt = h->cfg.dt;
  a.num_timesteps = T;
  a.smooth_mask = smooth_mask;
  a.constrain_mask = constrain_mask;
  // ColoredMPPI clamps only control channel 1 after smoothing (colored_mppi_controller.cu:232-237)
  a.constrain_mode = h->cfg.controller == MPPI_CONTROLLER_COLORED ? 1 : 0;
  std::string err;
  const int nsys = num_systems > 0 ? num_systems : h->D;
  mppi_status st = h->model->launchFinalize(nsys, a, h->stream, err);
  if (st != MPPI_OK)
    return fail(h, st, err);
  h->results_in_io = false;
  // controls, states, outputs and the merge statistics come back with ONE copy into pinned memory and one synchronisation
  HIP_TRY(h, hipMemcpyAsync(h->out_pin_h, h->out_block_d, sizeof(float) * h->out_floats, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int z = 0; z < nsys; z++)
  {
    const float* c = h->out_pin_h + (h->ctrl_out_d - h->out_block_d) + (size_t)z * T * h->C;
    const float* x = h->out_pin_h + (h->state_out_d - h->out_block_d) + (size_t)z * T * h->S;
    if (ctrl_out[z])
      std::copy(c, c + (size_t)T * h->C, ctrl_out[z]->begin());
    if (state_out[z])
      std::copy(x, x + (size_t)T * h->S, state_out[z]->begin());
  }
  h->out_pin_fresh = true;
  return MPPI_OK;
}

/* (here, not with the other debug entry points: the counters are a `static __device__` array of reduce_kernels.hpp, i.e. one per
 * translation unit — this is the unit that launches the merge kernels which fill it) */
#if defined(MPPI_COMBINE_TIMING)
extern "C" int mppi_debug_read_combine_timing(unsigned long long* out, int capacity)
{
  if (!out || capacity < 32)
    return -32;
  if (hipDeviceSynchronize() != hipSuccess)
    return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(kernels::g_combine_timing), sizeof(unsigned long long) * 32) != hipSuccess)
    return -2;
  return 32;
}
#endif
