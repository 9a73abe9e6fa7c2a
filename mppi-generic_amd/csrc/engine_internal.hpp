/**
 * engine_internal.hpp — what the translation units of the engine share: the handle (mppi_handle_s), the model table, the error /
 * locking macros and the declarations of the internal functions that cross a file boundary.  Not installed, not part of the
 * C ABI (include/mppi_amd.h is); the kernels' argument structs live in reduce_kernels.hpp / exact_reduce_kernels.hpp and in
 * include/mppi_amd/engine/.
 *
 * Round 6 split the single 4 300-line engine.hip by concern (same C ABI, same tests):
 *   engine_core.hip         version / model table / lifecycle (mppi_create, mppi_destroy) / parameters / blobs, .npz, rocRAND
 *   engine_iteration.hip    one optimisation iteration: rollout launch, merges (fused, streamed, reference order, Tsallis),
 *                           the exchange inside an iteration, the post-processing pass
 *   engine_controllers.hip  mppi_compute_control for the Vanilla / Colored, Tube and Robust controllers, hand-over (inbox,
 *                           flags, split finalize), getters, slide
 *   engine_loop.hip         the device-resident loop (mppi_optimize), timing and kernel choice entry points
 *   engine_exchange.hip     multi-GPU set-up: exchange buffers, P2P mailbox sessions, RCCL communicator
 *   engine_operators.hip    kernel-level operators and probes of the C ABI (normExp, weighted reduction, Philox, textures, ...)
 * Functions defined in one of them and called from another are declared at the bottom of this header, hidden from the
 * library's export table (MPPI_ENGINE_INTERNAL); everything else keeps internal linkage in its file.
 */
#ifndef MPPI_AMD_CSRC_ENGINE_INTERNAL_HPP_
#define MPPI_AMD_CSRC_ENGINE_INTERNAL_HPP_

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "mppi_amd.h"
#include "npz_reader.hpp"
#include "mppi_amd/engine/model_instance.hpp"
#include "reduce_kernels.hpp"
#include "exact_reduce_kernels.hpp"
static_assert(mppi::kernels::STATS_STRIDE == mppi::kernels::MERGE_CONTROL_STATS, "mergeControlKernel writes the statistics block");
#include "mppi_amd/utils/texture_helpers/two_d_texture_helper.hpp"
#include "mppi_amd/utils/nn_helpers/lstm_lstm_helper.hpp"

using namespace mppi;
using namespace mppi::engine;

/* ------------------------------------------------------------------------------------------------------------------ */
inline thread_local std::string g_create_error;

/* the model table (include/mppi_amd/engine/model_registry.hpp): filled by static initialisers of the model translation
 * units (csrc/models/[*].hip) and of out-of-tree plugins, hence a function-local static */
struct ModelRegistry
{
  std::mutex mu;
  std::map<std::pair<std::string, int>, mppi_model_factory> factories;
  std::string listing;
  int refused = 0;  // registrations turned down so far (mppi_load_plugin reports the ones of the library it loaded)
  std::string last_refusal;
};
inline ModelRegistry& registry()
{
  static ModelRegistry* r = new ModelRegistry();  // never destroyed: plugins may unregister nothing at exit
  return *r;
}
inline ModelBase* makeModel(const std::string& name, bool colored)
{
  ModelRegistry& r = registry();
  mppi_model_factory f = nullptr;
  {
    std::lock_guard<std::mutex> lock(r.mu);
    auto it = r.factories.find({ name, colored ? MPPI_SAMPLER_COLORED : MPPI_SAMPLER_GAUSSIAN });
    if (it != r.factories.end())
      f = it->second;
  }
  return f ? static_cast<ModelBase*>(f()) : nullptr;
}

struct mppi_handle_s
{
  /* Entry points of one handle are serialised: the reference's controllers are single-caller, but its BasePlant calls them
   * from two threads (state callback + control loop, core/base_plant.hpp:398-428) behind its own mutex — here the handle
   * carries it.  mppi_enforce_constraints' host path deliberately does NOT take it (a control publication must never wait
   * for a computeControl in flight); it reads the control ranges under params_mu only. */
  std::recursive_mutex mu;
  std::mutex params_mu;
  mppi_config cfg{};
  std::string model_name;
  std::unique_ptr<ModelBase> model;
  int D = 1, S = 0, C = 0, O = 0;
  int K_local = 0, K_offset = 0;
  int bx = 64, by = 1, bz = 1;
  bool pipeline = false;
  int num_blocks = 0;
  int TC = 0, PS = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string last_error;

  /* device buffers */
  float* x0_d = nullptr;           // [D][S]
  float* mean_d = nullptr;         // [D][T][C]
  float* costs_d = nullptr;        // [D][K_local]
  float* partials_d = nullptr;     // [D][num_blocks][PS]: the records the NEXT rollout launch writes (+ their transposed copy)
  /* Streamed merge (rolloutPipelineKernel, STREAM_MERGE): the records of the last rollout launch stay un-merged in
   * pending_records_d until the next rollout launch merges them in its sampler waves — or flushMerge() runs combineKernel on
   * them, which everything that reads mean_d / stats_d does first.  Two record buffers alternate. */
  float* partials_alt_d = nullptr;
  const float* pending_records_d = nullptr;
  unsigned long long n_rollout_launches = 0, n_merge_launches = 0;  // mppi_get_launch_counts
  bool stream_merge_enabled = true;  // MPPI_AMD_NO_STREAM_MERGE=1 switches it off (A/B)
  bool merge_control_enabled = true;  // MPPI_AMD_NO_MERGE_CONTROL=1: combineKernel + control phase as two launches (A/B, tests)
  float* send_d = nullptr;         // [D][PS]
  float* recv_d = nullptr;         // [world][D][PS]
  float* gather_tmp_d = nullptr;   // [D][world][PS] (records regrouped per system)
  float* stats_d = nullptr;        // [D][STATS_STRIDE]
  float* eps_d = nullptr;          // [n_eps_iters][K_local][T][C]
  float* samples_d = nullptr;      // [D][K_local][T][C]
  float* rows_d = nullptr;         // [num_blocks][bx * bz][rowStride]: the sampler's rows when they do not fit the LDS
  float* fin_scratch_d = nullptr;  // [D][(2 T + 4) C]: smoothing buffer + sequence of the finalize kernels at long horizons
  bool rm_pipeline = false;        // Robust MPPI: ask the model for its role-pipelined rollout kernel (rows in HBM, bx = 64)
  bool rows_in_hbm = false;
  /* ColoredMPPI options (controllers/ColoredMPPI/colored_mppi_controller.cuh:18-22, 159-193): Tsallis weights and state leash */
  float tsallis_gamma = 0.0f, tsallis_r = 0.0f;
  float* tsallis_weights_d = nullptr;  // [K_local]
  float* tsallis_record_d = nullptr;   // [PS] K-sharded Tsallis: {sum w v | rho, sum w, sum w^2, 0} of this rank
  /* reference-order reduction (mppi_set_reduction_mode, exact_reduce_kernels.hpp) */
  int reduction_mode = MPPI_REDUCTION_FUSED;
  int sum_strides = 32;                // GaussianParams::sum_strides (sampling_distributions/gaussian/gaussian.cuh:30)
  float* exact_weights_d = nullptr;    // [D][K_local]
  float* exact_inter_d = nullptr;      // [D][ceil(K_local / sum_strides)][T*C]
  int exact_inter_cells = 0;
  float* std_dev_time_d = nullptr;     // [D][T][C] time_specific_std_dev table
  bool leash_active = false;
  int leash_jump = 1;
  std::vector<float> leash_dist;       // [S]
  float* history_d = nullptr;      // [2][C]
  float* ctrl_in_d = nullptr;      // [D][T][C]
  float* ctrl_out_d = nullptr;     // [D][T][C]
  float* state_out_d = nullptr;    // [D][T][S]
  float* output_out_d = nullptr;   // [D][T][O]
  /* x0_d | mean_d | history_d are slices of ONE device block, ctrl_out_d | state_out_d | output_out_d | stats_d of
   * another, each mirrored in pinned host memory: mppi_compute_control hands its inputs over with one copy and takes its
   * results back with one copy and one synchronisation (single-system controllers; the others copy slice by slice) */
  float* in_block_d = nullptr;
  float* out_block_d = nullptr;
  float* in_pin_h = nullptr;
  float* out_pin_h = nullptr;
  /* low-latency hand-over of the single-system controllers (computeControlVanilla): host memory mapped into the device —
   * the first kernel reads the inputs from io_in, the finalize kernel writes the results to io_out and raises io_flags the
   * host spins on (flag 0: control sequence + statistics out; flag 1: state / output trajectories out) */
  float* io_in_h = nullptr;
  float* io_in_dev = nullptr;
  float* io_out_h = nullptr;
  float* io_out_dev = nullptr;
  unsigned* io_flags_h = nullptr;
  unsigned* io_flags_dev = nullptr;
  unsigned io_seq = 0;
  bool results_in_io = false;      // the last finalize pass wrote to io_out_h (low-latency path), not to out_block_d
  bool traj_pending = false;       // state_h / output of the last call are still being written by the finalize kernel
  bool low_latency = true;         // MPPI_AMD_NO_SPIN=1 in the environment: copy + hipStreamSynchronize hand-over instead
  /* Round 5: the input block of the low-latency hand-over is DEVICE memory the host writes through the PCIe BAR
   * (hipExtMallocWithFlags(hipDeviceMallocFinegrained) on a large-BAR device: the allocation accepts CPU stores,
   * tools/ubench/bar_write.hip — {write 2 KB, launch, flag back} 8.3 us against 17.4 us with mapped host memory).  io_in_h and
   * io_in_dev then are the same pointer; the host only ever WRITES it (write-combined, fenced before the launch).  With it
   * the Vanilla / Colored computeControl needs no ingest launch: the first rollout launch reads its mean, every rollout launch
   * and the finalize kernel their initial state and history, from the inbox (HBM, not PCIe).  MPPI_AMD_BAR_INBOX=0: mapped
   * host memory + ingest kernel as before. */
  bool bar_inbox = false;
  int combine_sharded_max_blocks = -1;  // co-residency bound of combineShardedKernel on this device (-1: not asked yet)
  /* Split hand-over (round 5; Vanilla / Colored and Tube MPPI, low-latency path): the finalize pass as two launches — the control phase on
   * the handle's stream, the re-rollout of the state trajectory on side_stream, which waits for it on a device flag — so the re-rollout of call N
   * (a lone wave, T dependent steps: 22 of a Cartpole call's 61 us period) runs beside the rollouts of call N + 1.  The
   * trajectory phase reads nothing but a carry block the control phase wrote (finalize_kernel.hpp: FinalizeArgs::phases) and
   * writes nothing but the trajectory part of io_out and its flag; two carry blocks alternate, and the control phase of call
   * N + 2 is not enqueued before call N's trajectory flag is up (carry_seq).  Every OTHER entry point that touches the device
   * first orders the handle's stream behind the side stream (CHECK_HANDLE -> joinSideStream).  MPPI_AMD_SPLIT_FINALIZE=0: one
   * launch as before. */
  bool split_finalize = false;
  bool side_pending = false;        // a trajectory phase is (possibly) in flight that h->stream has not been ordered behind
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_side = nullptr;      // recorded behind every trajectory phase: what joinSideStream orders h->stream behind
  float* carry_d = nullptr;         // [2][in_floats] + 2 x 2 words: the blocks' ready flags (FinalizeArgs::carry_ready_d)
  float* fin_scratch2_d = nullptr;  // the trajectory phase's own smoothing-buffer block at long horizons (fin_scratch_d's twin)
  unsigned carry_seq[2] = { 0, 0 };  // hand-over sequence number of the call whose trajectory phase reads carry block i (0: none)
  const float* x0_src_d = nullptr;    // where rollout launches read the initial state from (nullptr: x0_d)
  const float* mean_src_d = nullptr;  // where the NEXT rollout launch reads its nominal control from (nullptr: mean_d; one-shot)
  /* host-side stamps of the last low-latency Vanilla mppi_compute_control, microseconds since the call's first statement
   * (mppi_debug_host_stamps; tools/compute_control_host_timing.py): [0] inputs written, [1] ingest enqueued, [2] iterations
   * enqueued, [3] merge flushed, [4] finalize enqueued, [5] flag 0 seen, [6] results copied out */
  double host_stamps_us[8] = { 0 };
  float* step_pin_h = nullptr;     // [S + C] host memory mapped into the device: [x | u] of a single model step
  float* step_pin_dev = nullptr;   // its device address
  unsigned step_seq = 0;           // hand-over counter of the model-step flag (io_flags[8])
  size_t in_floats = 0, out_floats = 0;
  bool out_pin_fresh = false;      // out_pin_h holds the results (incl. stats) of the last finalize pass; reset by launches
  bool stats_h_fresh = false;      // stats_h IS the statistics of the last merge (parsed at a low-latency hand-over); reset by launches
  float* step_x_d = nullptr;       // [S]
  float* step_u_d = nullptr;       // [C]
  int n_eps_iters = 0;
  size_t noise_floats = 0;         // injected-noise floats per rollout (T*C, or C*(2T+2) spectrum entries when colored)
  hipEvent_t ev_a = nullptr, ev_b = nullptr;

  /* host state (the reference's control_, control_history_, state_, nominal_* members) */
  std::vector<float> control_h, history_h, state_h, nominal_control_h, nominal_state_h, slide_scale_h;
  bool nominal_state_init = false;
  std::vector<float> tube_x_h;     // Tube MPPI: the nominal system's current state (the reference's nominal_state_; [S])
  float nominal_threshold = 20.0f;  // Tube-MPPI/tube_mppi_controller.cuh:20
  mppi_stats stats_h{};
  uint32_t generation = 0;
  int last_stride = 1;
  bool independent_noise = false;  // use_same_noise_for_all_distributions == false (sampling_distribution.cuh:20)
  int external_iteration = 0;  // opt_iter of a caller-driven loop (mppi_iteration_local), reset by mppi_upload_state
  int noise_source = MPPI_NOISE_PHILOX_FUSED;

  /* Robust MPPI (controllers/R-MPPI/robust_mppi_controller.cuh:46-53, 270-310) */
  float value_function_threshold = 1000.0f;
  int num_candidates = 9;
  int samples_per_candidate = 32;  // eval_dyn_kernel_dim_.x default (robust_mppi_controller.cu:326-330)
  bool fb_accumulate_all = false;
  bool gains_set = false;
  bool rm_nominal_init = false;
  int best_index = 0, nominal_stride = 0, real_stride = 0;
  std::vector<float> rm_nominal_state, rm_line_weights, rm_cand_states, rm_cand_costs, rm_cand_free_energy,
      nominal_history_h;
  std::vector<int> rm_cand_strides;
  float* cand_states_d = nullptr;
  float* cand_costs_d = nullptr;
  int* cand_strides_d = nullptr;
  int cand_capacity = 0;     // candidates * samples the cost buffers hold
  int cand_capacity_nc = 0;  // candidates the state / stride buffers hold
  /* the same three in host memory mapped into the device (low-latency hand-over: the candidate kernel reads its inputs and
   * writes its costs in place, the host waits on io_flags[9]): [states (nc * S) | strides (nc ints) | costs (nc * ns)] */
  float* cand_io_h = nullptr;
  float* cand_io_dev = nullptr;
  unsigned cand_seq = 0;

  /* rocRAND host API (MPPI_NOISE_ROCRAND_HOST; librocrand.so loaded lazily): the reference's structure — a library
   * generator fills an eps buffer in HBM (curandGenerateNormal, sampling_distributions/gaussian/gaussian.cu:380-394) */
  void* rocrand_lib = nullptr;
  void* rocrand_gen = nullptr;
  float* rocrand_eps_d = nullptr;  // [K_local][noise floats per rollout], refilled before every rollout launch
  /* RCCL (loaded lazily) */
  void* rccl_lib = nullptr;
  void* comm = nullptr;

  /* P2P mailbox exchange over xGMI (mppi_p2p_*): this rank's mailbox — records [2 parities][world][D * PS] followed by
   * flags [2][world] — lives in this GPU's memory and is written by the peers' postRecordsKernel */
  float* mbox_d = nullptr;
  size_t mbox_bytes = 0;
  bool mbox_uncached = false;
  float* peer_mbox[16] = { nullptr };
  bool peer_opened[16] = { false };  // hipIpcOpenMemHandle'd (to be closed)
  bool p2p_ready = false;
  bool exchange_failed = false;  // a merge kernel gave up waiting for a peer (stats[6] mark), sticky until mppi_p2p_connect
  unsigned xseq = 0;  // exchange sequence number: flags carry it, its parity selects the mailbox half
  size_t mbox_aux_off = 0;  // aux channel of the mailbox (Robust MPPI candidate costs), in 4-byte words from mbox_d
  unsigned aseq = 0;        // its own sequence number
};

struct RocrandApi
{
  int (*create)(void**, int) = nullptr;
  int (*destroy)(void*) = nullptr;
  int (*set_seed)(void*, unsigned long long) = nullptr;
  int (*set_offset)(void*, unsigned long long) = nullptr;
  int (*set_stream)(void*, hipStream_t) = nullptr;
  int (*normal)(void*, float*, size_t, float, float) = nullptr;
};
inline RocrandApi g_rocrand;

/** floats of injected / library-generated noise one rollout launch consumes: [K_local][noise floats], times D slabs when every
 *  distribution draws its own noise */
static inline size_t epsFloatsPerIteration(const mppi_handle_s* h)
{
  return (size_t)h->K_local * h->noise_floats * (h->independent_noise ? (size_t)h->D : 1);
}

/** the multi-rank path (local merge -> all-gather -> global merge) runs for world_size > 1, and for a world of ONE when
 *  the caller asks for it (cfg.force_exchange): that exercises the RCCL plumbing on a single GPU */
static inline bool exchangeActive(const mppi_handle_s* h)
{
  return h->cfg.world_size > 1 || h->cfg.force_exchange != 0;
}

inline mppi_status fail(mppi_handle h, mppi_status s, const std::string& msg)
{
  if (h)
    h->last_error = msg;
  else
    g_create_error = msg;
  return s;
}

#define HIP_TRY(h, expr)                                                                                             \
  do                                                                                                                 \
  {                                                                                                                  \
    hipError_t e__ = (expr);                                                                                         \
    if (e__ != hipSuccess)                                                                                           \
      return fail((h), MPPI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));                          \
  } while (0)

#define MPPI_TRY(expr)               \
  do                                 \
  {                                  \
    mppi_status s__ = (expr);        \
    if (s__ != MPPI_OK)              \
      return s__;                    \
  } while (0)

/** split hand-over: order the handle's stream behind the trajectory phase that may still run on the side stream — whatever an
 *  entry point enqueues or synchronises on h->stream then sees the state a single in-order stream would have given it */
static inline void joinSideStream(mppi_handle h)
{
  if (!h->side_pending)
    return;
  (void)hipStreamWaitEvent(h->stream, h->ev_side, 0);
  h->side_pending = false;
}
/** entry points: lock the handle, join the side stream.  CHECK_HANDLE_HOST: the few that a control loop calls every cycle and
 *  that either never touch the device or are written for the split (mppi_compute_control, the result getters, mppi_slide,
 *  mppi_model_step): no join, so the next call's rollouts are not ordered behind the last call's re-rollout */
#define CHECK_HANDLE_HOST(h)          \
  if (!(h))                           \
    return MPPI_ERR_INVALID_ARG;      \
  std::lock_guard<std::recursive_mutex> handle_lock__((h)->mu)
#define CHECK_HANDLE(h)  \
  CHECK_HANDLE_HOST(h);  \
  joinSideStream(h)

/** buffers of the reference-order reduction: samples in HBM (what cfg.save_samples allocates), weights, cell partials */
inline mppi_status ensureExactBuffers(mppi_handle h)
{
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (!h->samples_d)
    HIP_TRY(h, hipMalloc((void**)&h->samples_d, sizeof(float) * (size_t)h->D * h->K_local * h->TC));
  if (!h->exact_weights_d)
    HIP_TRY(h, hipMalloc((void**)&h->exact_weights_d, sizeof(float) * (size_t)h->D * h->K_local));
  const int cells = (h->K_local - 1) / h->sum_strides + 1;
  if (!h->exact_inter_d || cells > h->exact_inter_cells)
  {
    if (h->exact_inter_d)
      (void)hipFree(h->exact_inter_d);
    h->exact_inter_d = nullptr;
    HIP_TRY(h, hipMalloc((void**)&h->exact_inter_d, sizeof(float) * (size_t)h->D * cells * h->TC));
    h->exact_inter_cells = cells;
  }
  // the attribute belongs to (function, device): set per call — it is cheap — rather than once per process
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kernels::exactWeightsKernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kernels::EXACT_WEIGHTS_LDS_BYTES));
  return MPPI_OK;
}

/* ---------------------------------------------------------------- roctx ranges --------------------------------------- */
/**
 * Marker ranges around the enqueue of the rollout, merge and post-processing kernels (the reference has no
 * profiler ranges; `rocprofv3 --marker-trace --kernel-trace` then attributes the kernels of an iteration).  Opt-in:
 * MPPI_AMD_ROCTX=1 — libroctx64 is dlopen'ed on first use; when the variable is unset a range is one predictable branch.
 */
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)();
struct Roctx
{
  roctx_push_fn push = nullptr;
  roctx_pop_fn pop = nullptr;
  Roctx()
  {
    const char* on = getenv("MPPI_AMD_ROCTX");
    if (!on || on[0] == '0' || on[0] == '\0')
      return;
    for (const char* name : { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" })
    {
      void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!lib)
        continue;
      push = (roctx_push_fn)dlsym(lib, "roctxRangePushA");
      pop = (roctx_pop_fn)dlsym(lib, "roctxRangePop");
      if (push && pop)
        return;
      push = nullptr;
      pop = nullptr;
    }
  }
};
inline const Roctx& roctx()
{
  static const Roctx r;
  return r;
}
struct RoctxRange
{
  bool active;
  explicit RoctxRange(const char* name) : active(roctx().push != nullptr)
  {
    if (active)
      roctx().push(name);
  }
  ~RoctxRange()
  {
    if (active)
      roctx().pop();
  }
};


static inline bool tsallisActive(const mppi_handle_s* h)
{  // colored_mppi_controller.cu:198: the exponential weights unless BOTH parameters are set
  return h->cfg.controller == MPPI_CONTROLLER_COLORED && h->tsallis_gamma != 0.0f && h->tsallis_r != 0.0f;
}

/** An iteration loop that leaves through an error must not leave pending_records_d behind: the next call would upload a fresh
 *  mean and its first launch would merge the stale records over it.  (After a successful flushMerge the pointer is null.) */
struct PendingRecordsGuard
{
  mppi_handle h;
  ~PendingRecordsGuard()
  {
    h->pending_records_d = nullptr;
  }
};


/* ---------------------------------------------------------------- internal functions that cross a file boundary ------------ */
#define MPPI_ENGINE_INTERNAL __attribute__((visibility("hidden")))
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
MPPI_ENGINE_INTERNAL extern nccl_allgather_fn g_ncclAllGather;  ///< engine_exchange.hip (mppi_comm_init_rccl resolves it)

MPPI_ENGINE_INTERNAL mppi_status rocrandFill(mppi_handle h);
MPPI_ENGINE_INTERNAL mppi_status launchCombine(mppi_handle h, const float* records, int num_records, int finalize, float* record_out,
                                 int k_total, bool world_major = false, const unsigned* wait_flags = nullptr,
                                 unsigned wait_seq = 0, const kernels::PostTargets* post = nullptr);
MPPI_ENGINE_INTERNAL bool streamMergeApplies(const mppi_handle_s* h);
MPPI_ENGINE_INTERNAL float* recordsTransposed(const mppi_handle_s* h, float* records);
MPPI_ENGINE_INTERNAL mppi_status flushMerge(mppi_handle h);
MPPI_ENGINE_INTERNAL mppi_status launchRollout(mppi_handle h, int iteration, int stride);
MPPI_ENGINE_INTERNAL mppi_status iterationLocal(mppi_handle h, int iteration, int stride);
MPPI_ENGINE_INTERNAL mppi_status iterationMerge(mppi_handle h);
MPPI_ENGINE_INTERNAL mppi_status iteration(mppi_handle h, int it, int stride);
MPPI_ENGINE_INTERNAL mppi_status fetchStats(mppi_handle h);
MPPI_ENGINE_INTERNAL bool allFinite(const std::vector<float>& v);
MPPI_ENGINE_INTERNAL mppi_status finalize(mppi_handle h, const float* ctrl_in_d, int smooth_mask, int constrain_mask,
                            std::vector<float>* ctrl_out[2], std::vector<float>* state_out[2], int num_systems = 0);
MPPI_ENGINE_INTERNAL void parseStats(mppi_handle h, const float* st);
MPPI_ENGINE_INTERNAL mppi_status modelStepInPlace(mppi_handle h, float* x, float* u, float dt, int enforce);

#endif
