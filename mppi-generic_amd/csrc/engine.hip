/**
 * engine.hip — implementation of the C ABI in include/mppi_amd.h: handle, device buffers, controller loops.
 *
 * Host-side logic restated from the reference's controllers (paths relative to the reference's include/mppi/):
 *   Vanilla loop            controllers/MPPI/mppi_controller.cu:151-241
 *   Tube loop               controllers/Tube-MPPI/tube_mppi_controller.cu:157-341
 *   slide / history         controllers/controller.cuh:351-356, 588-615
 * What differs by design: per iteration there is no D2H copy and no host scan — baseline, normaliser and the weighted
 * reduction stay on the device (rollout_kernel.hpp, reduce_kernels.hpp); smoothing and the nominal state trajectory
 * run in finalize_kernel.hpp.  The host owns the control sequence between calls exactly as the reference's control_.
 */
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
#include "mppi_amd/utils/texture_helpers/two_d_texture_helper.hpp"
#include "mppi_amd/utils/nn_helpers/lstm_lstm_helper.hpp"

using namespace mppi;
using namespace mppi::engine;

/* ------------------------------------------------------------------------------------------------------------------ */
static thread_local std::string g_create_error;

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
static ModelRegistry& registry()
{
  static ModelRegistry* r = new ModelRegistry();  // never destroyed: plugins may unregister nothing at exit
  return *r;
}
static ModelBase* makeModel(const std::string& name, bool colored)
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
  float* partials_d = nullptr;     // [D][num_blocks][PS]: the records the NEXT rollout launch writes
  /* Streamed merge (rolloutPipelineKernel, STREAM_MERGE): the records of the last rollout launch stay un-merged in
   * pending_records_d until the next rollout launch merges them in its sampler waves — or flushMerge() runs combineKernel on
   * them, which everything that reads mean_d / stats_d does first.  Two record buffers alternate. */
  float* partials_alt_d = nullptr;
  const float* pending_records_d = nullptr;
  unsigned long long n_rollout_launches = 0, n_merge_launches = 0;  // mppi_get_launch_counts
  bool stream_merge_enabled = true;  // MPPI_AMD_NO_STREAM_MERGE=1 switches it off (A/B)
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

namespace
{
struct RocrandApi
{
  int (*create)(void**, int) = nullptr;
  int (*destroy)(void*) = nullptr;
  int (*set_seed)(void*, unsigned long long) = nullptr;
  int (*set_offset)(void*, unsigned long long) = nullptr;
  int (*set_stream)(void*, hipStream_t) = nullptr;
  int (*normal)(void*, float*, size_t, float, float) = nullptr;
};
RocrandApi g_rocrand;
}  // namespace

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

static mppi_status fail(mppi_handle h, mppi_status s, const std::string& msg)
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
static mppi_status ensureExactBuffers(mppi_handle h)
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

/* ------------------------------------------------------------------------------------------------------------------ */
extern "C" {

const char* mppi_version(void)
{
  return "mppi-generic_amd 0.2 (gfx950)";
}

extern "C" const char* mppi_source_hash_impl(void);  // generated by buildlib.py (build/source_hash.cpp)
const char* mppi_source_hash(void)
{
  return mppi_source_hash_impl();
}

const char* mppi_status_string(mppi_status s)
{
  switch (s)
  {
    case MPPI_OK: return "ok";
    case MPPI_ERR_INVALID_ARG: return "invalid argument";
    case MPPI_ERR_UNKNOWN_MODEL: return "unknown model";
    case MPPI_ERR_NO_DEVICE: return "no usable HIP device";
    case MPPI_ERR_HIP: return "HIP runtime error";
    case MPPI_ERR_LAUNCH_SHAPE: return "unsupported launch shape";
    case MPPI_ERR_LDS_OVERFLOW: return "LDS request exceeds 160 KiB";
    case MPPI_ERR_STATE: return "invalid state for this call";
    case MPPI_ERR_NAN: return "non-finite control";
    case MPPI_ERR_COMM: return "RCCL error";
    case MPPI_ERR_UNSUPPORTED: return "unsupported";
  }
  return "unknown status";
}

int mppi_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

const char* mppi_list_models(void)
{
  ModelRegistry& r = registry();
  std::lock_guard<std::mutex> lock(r.mu);
  r.listing.clear();
  for (const auto& kv : r.factories)
    if (kv.first.second == MPPI_SAMPLER_GAUSSIAN || !r.factories.count({ kv.first.first, MPPI_SAMPLER_GAUSSIAN }))
      r.listing += (r.listing.empty() ? "" : "\n") + kv.first.first;
  return r.listing.c_str();
}

mppi_status mppi_register_model(const char* name, int sampler_kind, mppi_model_factory factory, int model_base_size)
{
  if (!name || !*name || !factory || (sampler_kind != MPPI_SAMPLER_GAUSSIAN && sampler_kind != MPPI_SAMPLER_COLORED))
    return fail(nullptr, MPPI_ERR_INVALID_ARG, "mppi_register_model: null name / factory or unknown sampler kind");
  if (model_base_size != engineAbiFingerprint())
  {
    ModelRegistry& rr = registry();
    std::lock_guard<std::mutex> lock(rr.mu);
    rr.refused++;
    rr.last_refusal = std::string("model '") + name + "' was built against other mppi_amd/engine headers than this library — "
                      "rebuild the plugin";
  }
  if (model_base_size != engineAbiFingerprint())
    return fail(nullptr, MPPI_ERR_INVALID_ARG,
                std::string("mppi_register_model('") + name + "'): built against a different mppi_amd/engine/model_instance.hpp "
                "than this library (ABI fingerprint " + std::to_string(model_base_size) + " vs " +
                std::to_string(engineAbiFingerprint()) + "): rebuild the plugin");
  ModelRegistry& r = registry();
  std::lock_guard<std::mutex> lock(r.mu);
  // the same factory again (a plugin loaded twice, a second controller of the same template instantiation) is fine; ANOTHER
  // factory under a name that is already taken would silently switch the kernels of handles created later: refused
  auto it = r.factories.find({ name, sampler_kind });
  if (it != r.factories.end() && it->second != factory)
    return fail(nullptr, MPPI_ERR_STATE, std::string("mppi_register_model('") + name + "'): the name is already registered "
                                         "with a different factory");
  r.factories[{ name, sampler_kind }] = factory;
  return MPPI_OK;
}

mppi_status mppi_register_model_checked(const char* name, int sampler_kind, mppi_model_factory factory, int model_base_size,
                                        unsigned flags)
{
  if ((flags & MPPI_MODEL_ROLE_SEPARATED) && !(flags & MPPI_MODEL_BARRIER_FREE_DECLARED))
  {
    const std::string what =
        std::string("model '") + (name ? name : "?") +
        "' asks for the role-separated rollout kernels (PIPELINE = true or a replicated-lane dynamics form) but its plugin "
        "classes do not all declare MPPI_BARRIER_FREE_STEP (mppi_amd/plugin/parallel_utils.hpp): a block barrier inside a "
        "per-step method would hang those kernels — declare it, or register with PIPELINE = false";
    {
      ModelRegistry& rr = registry();
      std::lock_guard<std::mutex> lock(rr.mu);
      rr.refused++;
      rr.last_refusal = what;
    }
    return fail(nullptr, MPPI_ERR_INVALID_ARG, "mppi_register_model: " + what);
  }
  return mppi_register_model(name, sampler_kind, factory, model_base_size);
}

mppi_status mppi_load_plugin(const char* path)
{
  if (!path)
    return fail(nullptr, MPPI_ERR_INVALID_ARG, "mppi_load_plugin: null path");
  int refused_before = 0;
  {
    std::lock_guard<std::mutex> lock(registry().mu);
    refused_before = registry().refused;
  }
  void* lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!lib)
  {
    const char* e = dlerror();
    return fail(nullptr, MPPI_ERR_INVALID_ARG, std::string("mppi_load_plugin: ") + (e ? e : "dlopen failed"));
  }
  {
    std::lock_guard<std::mutex> lock(registry().mu);
    if (registry().refused != refused_before)
      return fail(nullptr, MPPI_ERR_INVALID_ARG, "mppi_load_plugin: " + registry().last_refusal);
  }
  return MPPI_OK;  // its static initialisers have registered the models; the library stays loaded
}

const char* mppi_last_error(mppi_handle h)
{
  return h ? h->last_error.c_str() : g_create_error.c_str();
}

/* ---------------------------------------------------------------- lifecycle -------------------------------------- */
static void freeAll(mppi_handle h)
{
  // x0_d, mean_d, history_d and ctrl_out_d, state_out_d, output_out_d, stats_d are slices of in_block_d / out_block_d
  h->x0_d = h->mean_d = h->history_d = h->ctrl_out_d = h->state_out_d = h->output_out_d = h->stats_d = nullptr;
  if (h->rocrand_gen && g_rocrand.destroy)
    (void)g_rocrand.destroy(h->rocrand_gen);
  h->rocrand_gen = nullptr;
  for (int p = 0; p < 16; p++)
  {
    if (h->peer_opened[p] && h->peer_mbox[p])
      (void)hipIpcCloseMemHandle(h->peer_mbox[p]);
    h->peer_opened[p] = false;
    h->peer_mbox[p] = nullptr;
  }
  if (h->mbox_d)
    (void)hipFree(h->mbox_d);
  h->mbox_d = nullptr;
  h->p2p_ready = false;
  if (h->side_stream)
  {  // a trajectory phase may still be writing io_out
    (void)hipStreamSynchronize(h->side_stream);
    (void)hipStreamDestroy(h->side_stream);
  }
  if (h->ev_side)
    (void)hipEventDestroy(h->ev_side);
  h->side_stream = nullptr;
  h->ev_side = nullptr;
  h->split_finalize = h->side_pending = false;
  if (h->io_in_h)
    (void)(h->bar_inbox ? hipFree(h->io_in_h) : hipHostFree(h->io_in_h));
  if (h->io_out_h)
    (void)hipHostFree(h->io_out_h);
  if (h->io_flags_h)
    (void)hipHostFree(h->io_flags_h);
  h->io_in_h = h->io_out_h = nullptr;
  h->io_flags_h = nullptr;
  if (h->in_pin_h)
    (void)hipHostFree(h->in_pin_h);
  if (h->out_pin_h)
    (void)hipHostFree(h->out_pin_h);
  if (h->step_pin_h)
    (void)hipHostFree(h->step_pin_h);
  h->in_pin_h = h->out_pin_h = h->step_pin_h = nullptr;
  h->step_u_d = nullptr;  // slice of the step_x_d block
  float** bufs[] = { &h->in_block_d, &h->out_block_d, &h->costs_d,   &h->partials_d,  &h->partials_alt_d, &h->send_d,     &h->recv_d,
                     &h->eps_d,     &h->samples_d, &h->ctrl_in_d,  &h->step_x_d, &h->gather_tmp_d, &h->rows_d, &h->fin_scratch_d,
                     &h->fin_scratch2_d, &h->carry_d, &h->tsallis_weights_d, &h->tsallis_record_d, &h->rocrand_eps_d, &h->std_dev_time_d, &h->exact_weights_d, &h->exact_inter_d };
  for (float** b : bufs)
  {
    if (*b)
      (void)hipFree(*b);
    *b = nullptr;
  }
  if (h->cand_states_d)
    (void)hipFree(h->cand_states_d);
  if (h->cand_costs_d)
    (void)hipFree(h->cand_costs_d);
  if (h->cand_strides_d)
    (void)hipFree(h->cand_strides_d);
  if (h->cand_io_h)
    (void)hipHostFree(h->cand_io_h);
  if (h->ev_a)
    (void)hipEventDestroy(h->ev_a);
  if (h->ev_b)
    (void)hipEventDestroy(h->ev_b);
  if (h->own_stream && h->stream)
    (void)hipStreamDestroy(h->stream);
}

mppi_status mppi_create(const mppi_config* cfg, mppi_handle* out)
{
  if (!cfg || !out || !cfg->model)
    return fail(nullptr, MPPI_ERR_INVALID_ARG, "mppi_create: null argument");
  *out = nullptr;
  if (cfg->num_rollouts <= 0 || cfg->num_timesteps <= 0 || !(cfg->dt > 0.0f) || !(cfg->lambda > 0.0f) ||
      cfg->num_iters <= 0)
    return fail(nullptr, MPPI_ERR_INVALID_ARG, "mppi_create: num_rollouts, num_timesteps, dt, lambda, num_iters must be > 0");
  if (cfg->kernel_variant < 0 || cfg->kernel_variant > MPPI_KERNEL_PIPELINE)
    return fail(nullptr, MPPI_ERR_INVALID_ARG, "mppi_create: unknown kernel_variant");
  const int world = cfg->world_size > 0 ? cfg->world_size : 1;
  if (cfg->rank < 0 || cfg->rank >= world || cfg->num_rollouts % world != 0)
    return fail(nullptr, MPPI_ERR_INVALID_ARG, "mppi_create: rank/world_size invalid or num_rollouts not divisible by world_size");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MPPI_ERR_NO_DEVICE, "mppi_create: no HIP device visible (this library has no CPU path)");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(nullptr, MPPI_ERR_INVALID_ARG, "mppi_create: device ordinal out of range");

  std::unique_ptr<mppi_handle_s> h(new mppi_handle_s());
  h->cfg = *cfg;
  h->cfg.world_size = world;
  h->model_name = cfg->model;
  h->cfg.model = h->model_name.c_str();
  const bool colored = cfg->controller == MPPI_CONTROLLER_COLORED;
  h->model.reset(makeModel(h->model_name, colored));
  if (!h->model && colored && std::unique_ptr<ModelBase>(makeModel(h->model_name, false)))
    return fail(nullptr, MPPI_ERR_UNSUPPORTED, "mppi_create: model '" + h->model_name + "' has no colored-noise instantiation");
  if (!h->model)
    return fail(nullptr, MPPI_ERR_UNKNOWN_MODEL, "mppi_create: model '" + h->model_name + "' is not registered; have:\n" + mppi_list_models());
  {  // whichever way the model got into the table: role-separated kernels only for plugins that declare barrier-free steps
    const std::string undeclared = h->model->undeclaredBarrierFreePlugins();
    if (!undeclared.empty())
      return fail(nullptr, MPPI_ERR_INVALID_ARG,
                  "mppi_create: model '" + h->model_name + "' carries role-separated rollout kernels but these plugin classes do "
                  "not declare MPPI_BARRIER_FREE_STEP (mppi_amd/plugin/parallel_utils.hpp): " + undeclared +
                  "— a block barrier inside a per-step method would hang the GPU there; declare it or register with PIPELINE = false");
  }
  mppi_handle hp = h.get();
  switch (cfg->controller)
  {
    case MPPI_CONTROLLER_VANILLA: h->D = 1; break;
    case MPPI_CONTROLLER_COLORED: h->D = 1; break;
    case MPPI_CONTROLLER_ROBUST:
      h->D = 2;
      if (!h->model->supportsRMPPI())
        return fail(nullptr, MPPI_ERR_UNSUPPORTED, "mppi_create: model '" + h->model_name + "' is not instantiated for Robust MPPI");
      break;
    case MPPI_CONTROLLER_TUBE: h->D = 2; break;
    default:
      return fail(nullptr, MPPI_ERR_UNSUPPORTED, "mppi_create: controller kind not available in this build");
  }
  h->S = h->model->S;
  h->C = h->model->C;
  h->O = h->model->O;
  h->K_local = cfg->num_rollouts / world;
  h->K_offset = cfg->rank * h->K_local;
  h->bx = cfg->block_x > 0 ? cfg->block_x : h->model->default_bx;
  h->by = cfg->block_y > 0 ? cfg->block_y : h->model->default_by;
  h->bz = h->D;
  if (cfg->controller == MPPI_CONTROLLER_ROBUST)
  {  // rolloutRMPPIKernel: (64 rollouts, 1 lane, 2 systems); (32, 1, 2) when the sample rows of 64 x 2 overflow the LDS
    if ((cfg->block_x != 0 && cfg->block_x != 64 && cfg->block_x != 32) || (cfg->block_y != 0 && cfg->block_y != 1))
      return fail(nullptr, MPPI_ERR_LAUNCH_SHAPE, "mppi_create: Robust MPPI runs with block shape (64, 1, 2) or (32, 1, 2)");
    h->bx = cfg->block_x != 0 ? cfg->block_x : 64;
    h->by = 1;
    if (cfg->block_x == 0 && h->model->rmppiSharedBytes(64, cfg->num_timesteps) > MAX_LDS_BYTES)
      h->bx = 32;
  }
  if (cfg->controller == MPPI_CONTROLLER_ROBUST && cfg->kernel_variant != MPPI_KERNEL_FUSED &&
      (cfg->block_x == 0 || cfg->block_x == 64) && h->model->globalRowsFloats(1, 1, cfg->num_timesteps) > 0)
  {
    // replicated-lane (MFMA / four-lane) dynamics: the role-pipelined kernel (rmppi_pipeline_kernel.hpp), 64 rollouts x 2
    // systems per block with the sample rows in HBM — the rings take their place in the LDS
    h->model->setGlobalRows(reinterpret_cast<float*>(16));  // placeholder until the buffer exists: sizes the LDS request
    const size_t need = h->model->rmppiPipelineSharedBytes(cfg->num_timesteps);
    if (need > 0 && need <= MAX_LDS_BYTES)
    {
      h->rm_pipeline = true;
      h->rows_in_hbm = true;
      h->bx = 64;
    }
    else
      h->model->setGlobalRows(nullptr);
  }
  if (cfg->controller == MPPI_CONTROLLER_ROBUST && cfg->kernel_variant == MPPI_KERNEL_PIPELINE && !h->rm_pipeline)
    return fail(nullptr, MPPI_ERR_LAUNCH_SHAPE,
                "mppi_create: the pipelined Robust MPPI kernel needs a model registered for it (replicated-lane MFMA / four-lane "
                "dynamics, or a one-lane model registered with PIPELINE), the Gaussian sampler and block_x 0 or 64");
  // Tube with a pipeline-capable model and no explicit shape: fold the two systems into the lanes of a wave (32, 1, 2)
  if (h->D == 2 && cfg->controller == MPPI_CONTROLLER_TUBE && cfg->block_x == 0 && cfg->block_y == 0 &&
      cfg->kernel_variant != MPPI_KERNEL_FUSED && h->model->supportsPipelineFold(32, 1, 2) &&
      h->model->rolloutSharedBytes(32, 1, 2, cfg->num_timesteps, 2, true) <= MAX_LDS_BYTES)
  {
    h->bx = 32;
    h->by = 1;
  }
  // no shape requested and the model's default (bx, by) has no instantiation for this many systems (a Tube controller on a
  // model whose default shape exists for bz = 1 only): the registered shape for bz with the same lanes per rollout if there
  // is one, otherwise the first one listed (replicated-lane shapes come first)
  if (cfg->controller != MPPI_CONTROLLER_ROBUST && cfg->block_x == 0 && cfg->block_y == 0 &&
      !h->model->supportsShape(h->bx, h->by, h->bz) && !h->model->supportsPipelineFold(h->bx, h->by, h->bz))
  {
    std::vector<int> shapes;
    h->model->listShapes(shapes);
    int pick = -1;
    for (size_t i = 0; i + 2 < shapes.size(); i += 3)
    {
      if (shapes[i + 2] != h->bz)
        continue;
      if (pick < 0 || (shapes[i + 1] == h->by && shapes[pick + 1] != h->by))
        pick = (int)i;
    }
    if (pick >= 0)
    {
      h->bx = shapes[pick];
      h->by = shapes[pick + 1];
    }
  }
  // (Robust MPPI's two kernels are instantiated per model — checked above — not per block shape)
  if (cfg->controller != MPPI_CONTROLLER_ROBUST && !h->model->supportsShape(h->bx, h->by, h->bz) && !h->model->supportsPipelineFold(h->bx, h->by, h->bz))
    return fail(nullptr, MPPI_ERR_LAUNCH_SHAPE,
                "mppi_create: block shape (" + std::to_string(h->bx) + "," + std::to_string(h->by) + "," +
                    std::to_string(h->bz) + ") is not instantiated for model '" + h->model_name + "'");
  const bool pipe_ok = cfg->controller != MPPI_CONTROLLER_ROBUST &&
                       ((h->model->supportsPipeline() && h->bx == 64 && h->by == 1) ||
                        h->model->supportsPipelineFold(h->bx, h->by, h->bz) ||
                        h->model->supportsPipelineRep(h->bx, h->by, h->bz));
  if (cfg->kernel_variant == MPPI_KERNEL_PIPELINE && !pipe_ok && !h->rm_pipeline)
    return fail(nullptr, MPPI_ERR_LAUNCH_SHAPE,
                "mppi_create: the pipeline variant needs a model registered for it and block shape (64, 1), or (64, REP, 1) "
                "for replicated-lane (MFMA) dynamics");
  h->pipeline = pipe_ok && cfg->kernel_variant != MPPI_KERNEL_FUSED;
  size_t lds = cfg->controller == MPPI_CONTROLLER_ROBUST ?
                   h->model->rmppiSharedBytes(h->bx, cfg->num_timesteps) :
                   h->model->rolloutSharedBytes(h->bx, h->by, h->bz, cfg->num_timesteps, h->D, h->pipeline);
  const char* force_hbm_rows = getenv("MPPI_AMD_ROWS_IN_HBM");  // test hook: the HBM-row variant at any horizon
  const bool hbm_forced = force_hbm_rows && force_hbm_rows[0] == '1';
  if (h->pipeline && (lds > MAX_LDS_BYTES || hbm_forced) && h->model->globalRowsFloats(1, 1, cfg->num_timesteps) > 0)
  {
    // the role-pipelined kernels at horizons whose rows do not fit the LDS next to the output ring: rows in HBM (round 3:
    // the dynamics waves fetch a trip ahead, the clamped control reaches the cost waves through the ring)
    h->model->setGlobalRows(reinterpret_cast<float*>(16));  // placeholder until the buffer exists: sizes the LDS request
    const size_t need = h->model->rolloutSharedBytes(h->bx, h->by, h->bz, cfg->num_timesteps, h->D, true);
    if (need <= MAX_LDS_BYTES)
    {
      h->rows_in_hbm = true;
      lds = need;
    }
    else
      h->model->setGlobalRows(nullptr);
  }
  if (lds > MAX_LDS_BYTES && h->pipeline && cfg->kernel_variant == MPPI_KERNEL_AUTO)
  {  // the output ring does not fit next to the sample rows: fall back to the fused variant
    h->pipeline = false;
    lds = h->model->rolloutSharedBytes(h->bx, h->by, h->bz, cfg->num_timesteps, h->D, false);
  }
  const bool want_hbm_rows = !h->rm_pipeline && !h->rows_in_hbm && (lds > MAX_LDS_BYTES || hbm_forced);
  if (want_hbm_rows && cfg->controller == MPPI_CONTROLLER_ROBUST && h->model->globalRowsFloats(1, 1, cfg->num_timesteps) > 0)
  {
    // Robust MPPI at horizons whose rows of even 32 rollouts x 2 systems overflow the LDS (or on request): the (64, 1, 2)
    // block with the rows in HBM — rolloutRMPPIKernel reaches them through the same pointer as the fused rollout kernel
    if (lds > MAX_LDS_BYTES)
      h->bx = cfg->block_x != 0 ? cfg->block_x : 64;
    h->model->setGlobalRows(reinterpret_cast<float*>(16));  // placeholder until the buffer exists: sizes the LDS request
    lds = h->model->rmppiSharedBytes(h->bx, cfg->num_timesteps);
    if (lds <= MAX_LDS_BYTES)
      h->rows_in_hbm = true;
    else
      h->model->setGlobalRows(nullptr);
  }
  if (want_hbm_rows && cfg->controller != MPPI_CONTROLLER_ROBUST && cfg->kernel_variant != MPPI_KERNEL_PIPELINE &&
      h->model->globalRowsFloats(1, 1, cfg->num_timesteps) > 0)
  {
    // Horizons whose sample rows do not fit the LDS next to the default block: the rows move to HBM (the reference's layout — it has no
    // horizon limit, sampling_distribution.cu:169-205), the fused kernel with the default block shape runs on them.
    // (Cartpole K=16384, T=1500: 1.2 ms per iteration against 3.7 ms with 16-rollout blocks whose rows fit the LDS.)
    if (lds > MAX_LDS_BYTES)
    {
      h->bx = cfg->block_x > 0 ? cfg->block_x : h->model->default_bx;
      h->by = cfg->block_y > 0 ? cfg->block_y : h->model->default_by;
    }
    if (h->model->supportsShape(h->bx, h->by, h->bz))
    {
      h->pipeline = false;
      h->model->setGlobalRows(reinterpret_cast<float*>(16));  // placeholder until the buffer exists: sizes the LDS request
      lds = h->model->rolloutSharedBytes(h->bx, h->by, h->bz, cfg->num_timesteps, h->D, false);
      if (lds <= MAX_LDS_BYTES)
        h->rows_in_hbm = true;
      else
        h->model->setGlobalRows(nullptr);
    }
  }
  if (lds > MAX_LDS_BYTES && cfg->block_x == 0 && cfg->block_y == 0 && cfg->controller != MPPI_CONTROLLER_ROBUST &&
      cfg->kernel_variant != MPPI_KERNEL_PIPELINE)
  {
    // Long horizons with a sampler / controller that has no rows-in-HBM form: the sample rows of the default block (T * C
    // floats per rollout and system) do not fit the 160 KiB of LDS.  No shape was requested, so take the registered shape
    // with the most rollouts per block that does fit (fused variant).
    std::vector<int> shapes;
    h->model->listShapes(shapes);
    int best = -1;
    for (size_t i = 0; i + 2 < shapes.size(); i += 3)
    {
      if (shapes[i + 2] != h->bz)
        continue;
      const size_t need = h->model->rolloutSharedBytes(shapes[i], shapes[i + 1], shapes[i + 2], cfg->num_timesteps, h->D, false);
      if (need <= MAX_LDS_BYTES && (best < 0 || shapes[i] > shapes[best]))
        best = (int)i;
    }
    if (best >= 0)
    {
      h->bx = shapes[best];
      h->by = shapes[best + 1];
      h->pipeline = false;
      lds = h->model->rolloutSharedBytes(h->bx, h->by, h->bz, cfg->num_timesteps, h->D, false);
    }
  }
  if (lds > MAX_LDS_BYTES)
    return fail(nullptr, MPPI_ERR_LDS_OVERFLOW,
                "mppi_create: rollout kernel needs " + std::to_string(lds) + " B of LDS per block (max 163840) — the sample "
                "rows of a block live in LDS and this sampler / controller / kernel variant has no rows-in-HBM form; shorten "
                "the horizon or register a block shape with fewer rollouts");
  h->num_blocks = (h->K_local + h->bx - 1) / h->bx;
  h->TC = cfg->num_timesteps * h->C;
  h->PS = kernels::partialStride(cfg->num_timesteps, h->C);
  h->noise_source = cfg->noise_source;
  h->noise_floats = h->model->noiseFloatsPerRollout(cfg->num_timesteps);
  if (h->noise_source < 0 || h->noise_source > MPPI_NOISE_ROCRAND_HOST)
    return fail(nullptr, MPPI_ERR_INVALID_ARG, "mppi_create: unknown noise_source");

  HIP_TRY(nullptr, hipSetDevice(cfg->device));
  if (cfg->stream)
  {
    h->stream = (hipStream_t)cfg->stream;
  }
  else
  {
    HIP_TRY(nullptr, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  const int T = cfg->num_timesteps, D = h->D, S = h->S, C = h->C, K = h->K_local;
  auto alloc = [&](float** p, size_t n) -> hipError_t {
    hipError_t e = hipMalloc((void**)p, n * sizeof(float));
    if (e == hipSuccess)
      e = hipMemsetAsync(*p, 0, n * sizeof(float), h->stream);
    return e;
  };
#define ALLOC_OR_FAIL(ptr, n)                                                                       \
  do                                                                                                \
  {                                                                                                 \
    hipError_t e__ = alloc(&(ptr), (n));                                                            \
    if (e__ != hipSuccess)                                                                          \
    {                                                                                               \
      freeAll(hp);                                                                                  \
      return fail(nullptr, MPPI_ERR_HIP, std::string("hipMalloc " #ptr ": ") + hipGetErrorString(e__)); \
    }                                                                                               \
  } while (0)
  {
    auto pad4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    const size_t o_mean = pad4((size_t)D * S), o_hist = o_mean + pad4((size_t)D * T * C);
    h->in_floats = o_hist + pad4((size_t)4 * C);  // history: [2 systems][2][C] (RMPPI smooths both with their own)
    ALLOC_OR_FAIL(h->in_block_d, h->in_floats);
    h->x0_d = h->in_block_d;
    h->mean_d = h->in_block_d + o_mean;
    h->history_d = h->in_block_d + o_hist;
    const size_t o_state = pad4((size_t)D * T * C), o_out = o_state + pad4((size_t)D * T * S),
                 o_stats = o_out + pad4((size_t)D * T * h->O);
    h->out_floats = o_stats + pad4((size_t)D * kernels::STATS_STRIDE);
    ALLOC_OR_FAIL(h->out_block_d, h->out_floats);
    h->ctrl_out_d = h->out_block_d;
    h->state_out_d = h->out_block_d + o_state;
    h->output_out_d = h->out_block_d + o_out;
    h->stats_d = h->out_block_d + o_stats;
    if (hipHostMalloc((void**)&h->in_pin_h, h->in_floats * sizeof(float), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->out_pin_h, h->out_floats * sizeof(float), hipHostMallocDefault) != hipSuccess)
    {
      freeAll(hp);
      return fail(nullptr, MPPI_ERR_HIP, "hipHostMalloc of the pinned hand-over buffers failed");
    }
    memset(h->in_pin_h, 0, h->in_floats * sizeof(float));
    memset(h->out_pin_h, 0, h->out_floats * sizeof(float));
    const char* no_spin = getenv("MPPI_AMD_NO_SPIN");
    h->low_latency = !(no_spin && no_spin[0] == '1');
    const unsigned map_flags = hipHostMallocMapped | hipHostMallocCoherent;
    {
      int large_bar = 0;
      const char* env = getenv("MPPI_AMD_BAR_INBOX");
      if (h->low_latency && !(env && env[0] == '0') &&
          hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, cfg->device) == hipSuccess && large_bar &&
          hipExtMallocWithFlags((void**)&h->io_in_h, h->in_floats * sizeof(float), hipDeviceMallocFinegrained) == hipSuccess)
      {
        h->bar_inbox = true;
        h->io_in_dev = h->io_in_h;
      }
      else
        (void)hipGetLastError();
    }
    if ((!h->bar_inbox && (hipHostMalloc((void**)&h->io_in_h, h->in_floats * sizeof(float), map_flags) != hipSuccess ||
                           hipHostGetDevicePointer((void**)&h->io_in_dev, h->io_in_h, 0) != hipSuccess)) ||
        hipHostMalloc((void**)&h->io_out_h, h->out_floats * sizeof(float), map_flags) != hipSuccess ||
        hipHostMalloc((void**)&h->io_flags_h, 64, map_flags) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->io_out_dev, h->io_out_h, 0) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->io_flags_dev, h->io_flags_h, 0) != hipSuccess)
    {
      freeAll(hp);
      return fail(nullptr, MPPI_ERR_HIP, "hipHostMalloc of the device-mapped hand-over buffers failed");
    }
    memset(h->io_in_h, 0, h->in_floats * sizeof(float));
    memset(h->io_out_h, 0, h->out_floats * sizeof(float));
    memset(h->io_flags_h, 0, 64);
  }
  ALLOC_OR_FAIL(h->costs_d, (size_t)D * K);
  ALLOC_OR_FAIL(h->partials_d, (size_t)D * h->num_blocks * h->PS);
  ALLOC_OR_FAIL(h->partials_alt_d, (size_t)D * h->num_blocks * h->PS);
  {
    const char* no_stream = getenv("MPPI_AMD_NO_STREAM_MERGE");
    h->stream_merge_enabled = !(no_stream && no_stream[0] == '1');
  }
  ALLOC_OR_FAIL(h->send_d, (size_t)D * h->PS);
  ALLOC_OR_FAIL(h->recv_d, (size_t)world * D * h->PS);
  ALLOC_OR_FAIL(h->gather_tmp_d, (size_t)world * D * h->PS);
  ALLOC_OR_FAIL(h->ctrl_in_d, (size_t)D * T * C);
  ALLOC_OR_FAIL(h->step_x_d, (size_t)S + C);  // [x | u] of mppi_model_step
  h->step_u_d = h->step_x_d + S;
  // [x | u] of a single model step, mapped into the device: the kernel reads and writes it in place, the host waits on a flag
  if (hipHostMalloc((void**)&h->step_pin_h, ((size_t)S + C) * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent) !=
          hipSuccess ||
      hipHostGetDevicePointer((void**)&h->step_pin_dev, h->step_pin_h, 0) != hipSuccess)
  {
    freeAll(hp);
    return fail(nullptr, MPPI_ERR_HIP, "hipHostMalloc of the device-mapped model-step buffer failed");
  }
  if (cfg->save_samples)
    ALLOC_OR_FAIL(h->samples_d, (size_t)D * K * T * C);
  if (h->rows_in_hbm)
  {
    ALLOC_OR_FAIL(h->rows_d, h->model->globalRowsFloats(h->num_blocks, h->bx * h->bz, T));
    h->model->setGlobalRows(h->rows_d);
  }
  {
    // the finalize kernels keep the control sequence twice in LDS (smoothing buffer + result) while that is a minor share of
    // it; beyond 64 KiB per system (T * C > ~8 000) — or on request (test hook) — both live in HBM
    const char* force = getenv("MPPI_AMD_FINALIZE_SCRATCH");
    const size_t per_sys = kernels::finalizeScratchFloats(T, C);
    if (per_sys * sizeof(float) > 64 * 1024 || (force && force[0] == '1'))
      ALLOC_OR_FAIL(h->fin_scratch_d, (size_t)D * per_sys);
    // split hand-over (mppi_handle_s::split_finalize): Vanilla / Colored / Tube handles on the low-latency path
    const char* split = getenv("MPPI_AMD_SPLIT_FINALIZE");
    // (not on a caller's stream: there a stream synchronisation is the caller's way to wait for everything the library launched)
    if (h->low_latency && h->own_stream && world == 1 && !cfg->force_exchange && !(split && split[0] == '0') &&
        (((cfg->controller == MPPI_CONTROLLER_VANILLA || cfg->controller == MPPI_CONTROLLER_COLORED) && D == 1) ||
         (cfg->controller == MPPI_CONTROLLER_TUBE && D == 2)))
    {
      ALLOC_OR_FAIL(h->carry_d, 2 * h->in_floats + 4);  // two carry blocks + their ready words (one per system)
      if (hipMemsetAsync(h->carry_d, 0, sizeof(float) * (2 * h->in_floats + 4), h->stream) != hipSuccess)
      {
        freeAll(hp);
        return fail(nullptr, MPPI_ERR_HIP, "mppi_create: clearing the carry blocks");
      }
      if (h->fin_scratch_d)
        ALLOC_OR_FAIL(h->fin_scratch2_d, (size_t)D * per_sys);
      if (hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming) != hipSuccess)
      {
        freeAll(hp);
        return fail(nullptr, MPPI_ERR_HIP, "mppi_create: side stream of the split hand-over");
      }
      h->split_finalize = true;
    }
  }
#undef ALLOC_OR_FAIL
  {
    hipError_t e = hipEventCreate(&h->ev_a);
    if (e == hipSuccess)
      e = hipEventCreate(&h->ev_b);
    if (e == hipSuccess)
      e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess)
    {
      freeAll(hp);
      return fail(nullptr, MPPI_ERR_HIP, std::string("mppi_create: event / stream setup: ") + hipGetErrorString(e));
    }
  }

  h->control_h.assign((size_t)T * C, 0.0f);
  h->history_h.assign((size_t)2 * C, 0.0f);
  h->state_h.assign((size_t)T * S, 0.0f);
  h->nominal_control_h.assign((size_t)T * C, 0.0f);
  h->nominal_state_h.assign((size_t)T * S, 0.0f);
  h->tube_x_h.assign((size_t)S, 0.0f);
  h->slide_scale_h.assign(C, 0.0f);  // controller.cuh:67 slide_control_scale_ = Zero()
  h->nominal_history_h.assign((size_t)2 * C, 0.0f);
  h->rm_nominal_state.assign(S, 0.0f);
  {
    // MPPI_AMD_REDUCTION=reference | reference_fma: every handle of the process starts in the reference-order reduction
    const char* red = getenv("MPPI_AMD_REDUCTION");
    if (red && red[0] && !exchangeActive(hp))
    {
      const int mode = !strcmp(red, "reference") ? MPPI_REDUCTION_REFERENCE_ORDER :
                       !strcmp(red, "reference_fma") ? MPPI_REDUCTION_REFERENCE_ORDER_FMA : MPPI_REDUCTION_FUSED;
      if (mode != MPPI_REDUCTION_FUSED)
      {
        hp->reduction_mode = mode;
        if (ensureExactBuffers(hp) != MPPI_OK)
        {
          g_create_error = hp->last_error;
          freeAll(hp);
          return MPPI_ERR_HIP;
        }
      }
    }
  }
  *out = h.release();
  return MPPI_OK;
}

void mppi_destroy(mppi_handle h)
{
  if (!h)
    return;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream)
    (void)hipStreamSynchronize(h->stream);
  if (h->comm && h->rccl_lib)
  {
    typedef int (*destroy_fn)(void*);
    destroy_fn d = (destroy_fn)dlsym(h->rccl_lib, "ncclCommDestroy");
    if (d)
      (void)d(h->comm);
    h->comm = nullptr;
  }
  freeAll(h);
  delete h;
}

mppi_status mppi_get_dims(mppi_handle h, int* s, int* c, int* o, int* d)
{
  CHECK_HANDLE_HOST(h);
  if (s)
    *s = h->S;
  if (c)
    *c = h->C;
  if (o)
    *o = h->O;
  if (d)
    *d = h->D;
  return MPPI_OK;
}

mppi_status mppi_get_local_rollouts(mppi_handle h, int* k_local, int* k_offset)
{
  CHECK_HANDLE_HOST(h);
  if (k_local)
    *k_local = h->K_local;
  if (k_offset)
    *k_offset = h->K_offset;
  return MPPI_OK;
}

mppi_status mppi_get_launch_counts(mppi_handle h, unsigned long long* rollout_launches, unsigned long long* merge_launches)
{
  CHECK_HANDLE_HOST(h);
  if (rollout_launches)
    *rollout_launches = h->n_rollout_launches;
  if (merge_launches)
    *merge_launches = h->n_merge_launches;
  return MPPI_OK;
}

/* ---------------------------------------------------------------- parameters ------------------------------------- */
mppi_status mppi_set_dynamics_params(mppi_handle h, const void* pod, size_t nbytes)
{
  CHECK_HANDLE(h);
  if (!pod)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_dynamics_params: null");
  mppi_status s = h->model->setDynamicsParams(pod, nbytes);
  return s == MPPI_OK ? s : fail(h, s, "mppi_set_dynamics_params: size does not match the model's parameter struct");
}
mppi_status mppi_set_cost_params(mppi_handle h, const void* pod, size_t nbytes)
{
  CHECK_HANDLE(h);
  if (!pod)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_cost_params: null");
  mppi_status s = h->model->setCostParams(pod, nbytes);
  return s == MPPI_OK ? s : fail(h, s, "mppi_set_cost_params: size does not match the model's parameter struct");
}
mppi_status mppi_set_sampler_params(mppi_handle h, const mppi_gaussian_params* p)
{
  CHECK_HANDLE(h);
  if (!p || !p->std_dev || !p->control_cost_coeff)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_sampler_params: null");
  for (int i = 0; i < h->C * h->D; i++)
    if (!(p->std_dev[i] > 0.0f))
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_sampler_params: std_dev must be > 0");
  h->model->setSamplerParams(p, h->D);
  if (p->sum_strides > 0 && p->sum_strides != h->sum_strides)
  {
    h->sum_strides = p->sum_strides;
    if (h->reduction_mode != MPPI_REDUCTION_FUSED)
      MPPI_TRY(ensureExactBuffers(h));
  }
  return MPPI_OK;
}

mppi_status mppi_set_reduction_mode(mppi_handle h, int mode)
{
  CHECK_HANDLE(h);
  if (mode != MPPI_REDUCTION_FUSED && mode != MPPI_REDUCTION_REFERENCE_ORDER && mode != MPPI_REDUCTION_REFERENCE_ORDER_FMA)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_reduction_mode: unknown mode");
  if (mode != MPPI_REDUCTION_FUSED && exchangeActive(h))
    return fail(h, MPPI_ERR_UNSUPPORTED,
                "the reference-order reduction sums over ALL rollouts in index order: not available on a K-sharded handle");
  h->reduction_mode = mode;
  if (mode != MPPI_REDUCTION_FUSED)
    MPPI_TRY(ensureExactBuffers(h));
  return MPPI_OK;
}
mppi_status mppi_set_independent_noise(mppi_handle h, int independent)
{
  CHECK_HANDLE(h);
  if (independent && h->cfg.controller == MPPI_CONTROLLER_COLORED)
    return fail(h, MPPI_ERR_UNSUPPORTED, "independent noise per distribution: the colored-noise sampler has one distribution");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if ((independent != 0) != h->independent_noise)
  {  // buffers sized for the other layout are dropped; injected noise has to be injected again
    if (h->eps_d)
      HIP_TRY(h, hipFree(h->eps_d));
    h->eps_d = nullptr;
    h->n_eps_iters = 0;
    if (h->rocrand_eps_d)
      HIP_TRY(h, hipFree(h->rocrand_eps_d));
    h->rocrand_eps_d = nullptr;
    if (h->noise_source == MPPI_NOISE_INJECTED)
      h->noise_source = h->cfg.noise_source == MPPI_NOISE_INJECTED ? MPPI_NOISE_PHILOX_FUSED : h->cfg.noise_source;
  }
  h->independent_noise = independent != 0;
  return MPPI_OK;
}

mppi_status mppi_set_time_specific_std_dev(mppi_handle h, const float* std_dev)
{
  CHECK_HANDLE(h);
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // earlier launches may still read the table
  if (!std_dev)
  {
    h->model->setTimeSpecificStdDev(nullptr);
    return MPPI_OK;
  }
  const size_t n = (size_t)h->D * h->TC;
  for (size_t i = 0; i < n; i++)
    if (!(std_dev[i] > 0.0f))
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_time_specific_std_dev: every sigma[d][t][c] must be > 0");
  if (!h->std_dev_time_d)
    HIP_TRY(h, hipMalloc((void**)&h->std_dev_time_d, n * sizeof(float)));
  HIP_TRY(h, hipMemcpyAsync(h->std_dev_time_d, std_dev, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->model->setTimeSpecificStdDev(h->std_dev_time_d);
  return MPPI_OK;
}

mppi_status mppi_set_colored_noise_params(mppi_handle h, const float* exponents, float offset_decay_rate, float fmin)
{
  CHECK_HANDLE(h);
  if (!exponents)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_colored_noise_params: null");
  for (int i = 0; i < h->C; i++)
    if (!(exponents[i] >= 0.0f))
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_colored_noise_params: exponents must be >= 0");
  if (!(fmin >= 0.0f) || !(offset_decay_rate >= 0.0f))
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_colored_noise_params: fmin and offset_decay_rate must be >= 0");
  mppi_status s = h->model->setColoredNoiseParams(exponents, offset_decay_rate, fmin);
  return s == MPPI_OK ? s : fail(h, s, "mppi_set_colored_noise_params: the handle's sampler is Gaussian (create it with MPPI_CONTROLLER_COLORED)");
}
mppi_status mppi_set_colored_mppi_params(mppi_handle h, float gamma, float r_exp, const float* state_leash_dist,
                                         int leash_active, int leash_jump)
{
  CHECK_HANDLE(h);
  if (h->cfg.controller != MPPI_CONTROLLER_COLORED)
    return fail(h, MPPI_ERR_STATE, "mppi_set_colored_mppi_params: not a ColoredMPPI handle");
  if (leash_jump < 0 || leash_jump >= h->cfg.num_timesteps)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_colored_mppi_params: leash_jump must index the state trajectory");
  if (gamma != 0.0f && r_exp != 0.0f)
  {
    if (r_exp == 1.0f || !(gamma > 0.0f))
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_colored_mppi_params: Tsallis weights need gamma > 0 and r != 1");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!h->samples_d)  // the weighted mean is formed from the samples in HBM (control_samples_d_ of the reference)
      HIP_TRY(h, hipMalloc((void**)&h->samples_d, sizeof(float) * (size_t)h->D * h->K_local * h->TC));
    if (!h->tsallis_weights_d)
      HIP_TRY(h, hipMalloc((void**)&h->tsallis_weights_d, sizeof(float) * (size_t)h->K_local));
    if (exchangeActive(h) && !h->tsallis_record_d)  // K-sharded: this rank's record of the second exchange (iterationShardedTsallis)
      HIP_TRY(h, hipMalloc((void**)&h->tsallis_record_d, sizeof(float) * (size_t)h->D * h->PS));
  }
  h->tsallis_gamma = gamma;
  h->tsallis_r = r_exp;
  h->leash_active = leash_active != 0;
  h->leash_jump = leash_jump;
  if (state_leash_dist)
    h->leash_dist.assign(state_leash_dist, state_leash_dist + h->S);
  else if (h->leash_dist.empty())
    h->leash_dist.assign(h->S, 0.0f);
  return MPPI_OK;
}

mppi_status mppi_set_control_ranges(mppi_handle h, const float* lo_hi)
{
  CHECK_HANDLE(h);
  std::lock_guard<std::mutex> params_lock(h->params_mu);
  if (!lo_hi)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_control_ranges: null");
  h->model->setControlRanges(lo_hi);
  return MPPI_OK;
}
mppi_status mppi_set_control_deadband(mppi_handle h, const float* db)
{
  CHECK_HANDLE(h);
  std::lock_guard<std::mutex> params_lock(h->params_mu);
  if (!db)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_control_deadband: null");
  h->model->setControlDeadband(db);
  return MPPI_OK;
}
mppi_status mppi_set_lambda_alpha(mppi_handle h, float lambda, float alpha)
{
  CHECK_HANDLE(h);
  if (!(lambda > 0.0f))
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_lambda_alpha: lambda must be > 0");
  h->cfg.lambda = lambda;
  h->cfg.alpha = alpha;
  return MPPI_OK;
}
mppi_status mppi_set_num_iters(mppi_handle h, int n)
{
  CHECK_HANDLE(h);
  if (n <= 0)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_num_iters: must be > 0");
  h->cfg.num_iters = n;
  return MPPI_OK;
}
mppi_status mppi_set_slide_control_scale(mppi_handle h, const float* scale)
{
  CHECK_HANDLE(h);
  if (!scale)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_slide_control_scale: null");
  for (int i = 0; i < h->C; i++)
    h->slide_scale_h[i] = scale[i];
  return MPPI_OK;
}
mppi_status mppi_set_nominal_threshold(mppi_handle h, float t)
{
  CHECK_HANDLE(h);
  h->nominal_threshold = t;
  return MPPI_OK;
}
mppi_status mppi_set_model_blob(mppi_handle h, const char* name, const float* data, size_t count, const int* dims,
                                int ndims)
{
  CHECK_HANDLE(h);
  if (!name || !data || count == 0)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_model_blob: null or empty");
  // Network parameters and recurrent states must be finite.  det::tanh / det::sigmoid clamp their argument with min / max,
  // which return the non-NaN bound: tanh(NaN) = -1 (det_math.h) — a NaN can NOT travel through an activation the way it
  // does through tanhf().  With finite parameters a network's output is bounded by its last layer's |W| and |b| whatever its
  // input, so the states it drives stay finite from a finite initial state (mppi_compute_control refuses a non-finite one);
  // the one way a NaN could enter a network and be masked is a corrupt parameter file, which is refused here.
  if (strstr(name, "weights") || strstr(name, "lstm_state"))
    for (size_t i = 0; i < count; i++)
      if (!std::isfinite(data[i]))
        return fail(h, MPPI_ERR_NAN, "mppi_set_model_blob: '" + std::string(name) + "' holds a non-finite value at index " +
                                         std::to_string(i));
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::string err;
  mppi_status st = h->model->setBlob(name, data, count, dims, ndims, h->stream, err);
  if (st != MPPI_OK)
    return fail(h, st, err);
  // Networks of another shape than the replicated-lane (four lanes per rollout) form of the model is compiled for: no block
  // shape was asked for, so move to the registered one-lane shape (same or more rollouts per block: the per-block buffers
  // stay large enough) instead of refusing the next launch
  if (h->cfg.block_x == 0 && h->cfg.block_y == 0 && h->cfg.controller != MPPI_CONTROLLER_ROBUST &&
      h->model->fastShapeRefused(h->bx, h->by, h->bz))
  {
    std::vector<int> shapes;
    h->model->listShapes(shapes);
    int pick = -1;
    for (size_t i = 0; i + 2 < shapes.size(); i += 3)
    {
      if (shapes[i + 2] != h->bz || shapes[i] < h->bx || h->model->fastShapeRefused(shapes[i], shapes[i + 1], shapes[i + 2]))
        continue;
      if (pick < 0 || shapes[i] < shapes[pick])
        pick = (int)i;
    }
    if (pick >= 0)
    {
      h->bx = shapes[pick];
      h->by = shapes[pick + 1];
      h->pipeline = false;
      h->num_blocks = (h->K_local + h->bx - 1) / h->bx;
      if (h->rows_in_hbm)
      {  // ceil(K / bx) * bx can GROW with bx (K = 70: 3 x 32 = 96 rows, 2 x 64 = 128): the row buffer follows the shape
        float* rows = nullptr;
        const size_t n = h->model->globalRowsFloats(h->num_blocks, h->bx * h->bz, h->cfg.num_timesteps);
        HIP_TRY(h, hipMalloc((void**)&rows, n * sizeof(float)));
        HIP_TRY(h, hipMemsetAsync(rows, 0, n * sizeof(float), h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->rows_d)
          HIP_TRY(h, hipFree(h->rows_d));
        h->rows_d = rows;
        h->model->setGlobalRows(h->rows_d);
      }
    }
  }
  // the LDS request may depend on the blob (network size): re-check it
  const size_t lds = h->cfg.controller == MPPI_CONTROLLER_ROBUST ?
                         h->model->rmppiSharedBytes(h->bx, h->cfg.num_timesteps) :
                         h->model->rolloutSharedBytes(h->bx, h->by, h->bz, h->cfg.num_timesteps, h->D, h->pipeline);
  if (lds > MAX_LDS_BYTES)
    return fail(h, MPPI_ERR_LDS_OVERFLOW, "rollout kernel LDS request exceeds 160 KiB after loading '" + std::string(name) + "'");
  return MPPI_OK;
}


mppi_status mppi_set_lstm_initial_state(mppi_handle h, const float* hidden, const float* cell)
{
  CHECK_HANDLE(h);
  if (!hidden || !cell)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_set_lstm_initial_state: null");

  HIP_TRY(h, hipSetDevice(h->cfg.device));
  std::string err;
  mppi_status st = h->model->setLSTMInitialState(hidden, cell, h->stream, err);
  return st == MPPI_OK ? MPPI_OK : fail(h, st, err);
}

mppi_status mppi_lstm_lstm_initialize(int init_input_dim, int init_hidden_dim, const int* init_output_layers,
                                      int num_init_output_layers, const float* init_lstm_blob, const float* init_output_blob,
                                      int hidden_dim, int init_len, const float* buffer, int cols, float* hidden_cell_out)
{
  if (!init_output_layers || !init_lstm_blob || !init_output_blob || !buffer || !hidden_cell_out || num_init_output_layers < 2)
    return MPPI_ERR_INVALID_ARG;
  try
  {
    const std::vector<int> init_layers(init_output_layers, init_output_layers + num_init_output_layers);
    // the prediction network's own shape plays no part in the initialisation (only its hidden size does)
    mppi::LSTMLSTMHelper helper(init_input_dim, init_hidden_dim, init_layers, 1, hidden_dim, { hidden_dim + 1, 1 }, init_len);
    helper.getInitModel()->setAllValues(init_lstm_blob, init_output_blob);
    helper.initializeLSTM(buffer, cols, hidden_cell_out, hidden_cell_out + hidden_dim);
  }
  catch (const std::exception&)
  {
    return MPPI_ERR_INVALID_ARG;
  }
  return MPPI_OK;
}

/* ---------------------------------------------------------------- .npz model data ---------------------------------- */
static mppi_status setBlobD(mppi_handle h, const char* name, const std::vector<double>& v, const std::vector<int>& dims)
{
  std::vector<float> f(v.begin(), v.end());
  return mppi_set_model_blob(h, name, f.data(), f.size(), dims.data(), (int)dims.size());
}

/** FNN blob [W1 | b1 | W2 | b2 | ...] from keys {prefix}dynamics_W{i}, {prefix}dynamics_b{i} (fnn_helper.cu:96-174) */
static bool fnnBlobFromNpz(const std::map<std::string, npz::Array>& d, const std::string& prefix, std::vector<double>& blob,
                           std::string& err)
{
  blob.clear();
  for (int i = 1;; i++)
  {
    auto w = d.find(prefix + "dynamics_W" + std::to_string(i));
    auto b = d.find(prefix + "dynamics_b" + std::to_string(i));
    if (w == d.end() || b == d.end())
    {
      if (i == 1)
      {
        err = "no key '" + prefix + "dynamics_W1' / '" + prefix + "dynamics_b1' in the archive";
        return false;
      }
      return true;
    }
    blob.insert(blob.end(), w->second.data.begin(), w->second.data.end());
    blob.insert(blob.end(), b->second.data.begin(), b->second.data.end());
  }
}

mppi_status mppi_load_npz(mppi_handle h, const char* kind, const char* path, const char* prefix_c)
{
  CHECK_HANDLE(h);
  if (!kind || !path)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_load_npz: null");
  std::map<std::string, npz::Array> d;
  std::string err;
  if (!npz::load(path, d, err))
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_load_npz: " + err);
  std::string prefix = prefix_c ? prefix_c : "";
  const std::string k(kind);
  if (k == "dynamics")
  {
    std::vector<double> blob;
    if (!fnnBlobFromNpz(d, prefix, blob, err))
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_load_npz: " + err);
    return setBlobD(h, "dynamics_weights", blob, { (int)blob.size() });
  }
  if (k == "lstm" || k == "mean_lstm" || k == "unc_lstm")
  {
    // which network of the model: the steering / only LSTM, or the mean / uncertainty LSTM of RacerDubinsElevationLSTMUncertainty
    // (its constructor reads all three from one archive: "steering/model/", "terra/mean_network/", "terra/uncertainty_network/",
    // racer_dubins_elevation_lstm_unc.cu:30-33)
    const std::string blob_stem = k;
    // LSTMHelper::loadParams (lstm_helper.cu:514-585): optional trailing '/', optional "model/" in front, PyTorch gate
    // order i, f, g, o re-ordered to i, f, o, c, the two bias vectors summed; output network under {prefix}output/
    if (!prefix.empty() && prefix.back() != '/')
      prefix += "/";
    if (d.count("model/" + prefix + "lstm/weight_hh_l0"))
      prefix = "model/" + prefix;
    const char* keys[4] = { "lstm/weight_hh_l0", "lstm/weight_ih_l0", "lstm/bias_hh_l0", "lstm/bias_ih_l0" };
    for (const char* key : keys)
      if (!d.count(prefix + key))
        return fail(h, MPPI_ERR_INVALID_ARG, std::string("mppi_load_npz: no key '") + prefix + key + "' in the archive");
    const npz::Array &whh = d[prefix + keys[0]], &wih = d[prefix + keys[1]], &bhh = d[prefix + keys[2]],
                     &bih = d[prefix + keys[3]];
    const int H = (int)bhh.size() / 4;
    if (H <= 0 || (int)whh.size() != 4 * H * H || wih.size() % (4 * H) != 0 || bih.size() != bhh.size())
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_load_npz: inconsistent LSTM array shapes");
    const int I = (int)wih.size() / (4 * H);
    const int order[4] = { 0, 1, 3, 2 };  // blob gate g <- torch gate order[g]
    std::vector<double> blob;
    for (int g = 0; g < 4; g++)
      blob.insert(blob.end(), whh.data.begin() + (size_t)order[g] * H * H, whh.data.begin() + (size_t)(order[g] + 1) * H * H);
    for (int g = 0; g < 4; g++)
      blob.insert(blob.end(), wih.data.begin() + (size_t)order[g] * H * I, wih.data.begin() + (size_t)(order[g] + 1) * H * I);
    for (int g = 0; g < 4; g++)
      for (int i = 0; i < H; i++)
        blob.push_back(bhh.data[(size_t)order[g] * H + i] + bih.data[(size_t)order[g] * H + i]);
    for (const char* init : { "lstm/h0", "lstm/c0" })
    {
      auto it = d.find(prefix + init);
      for (int i = 0; i < H; i++)
        blob.push_back(it != d.end() && (int)it->second.size() == H ? it->second.data[i] : 0.0);
    }
    std::vector<double> out_blob;
    if (!fnnBlobFromNpz(d, prefix + "output/", out_blob, err))
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_load_npz: " + err);
    {
      // the reference sizes the network from the file (LSTMHelper(path, prefix), lstm_helper.cu:13-62): models that take a
      // "<network>_structure" blob get {H, H + I, output-network layer sizes ...} first; for the others the weight blobs
      // below are checked against the compiled shape
      std::vector<double> desc = { (double)H, (double)(H + I) };
      for (int i = 1;; i++)
      {
        auto b = d.find(prefix + "output/dynamics_b" + std::to_string(i));
        if (b == d.end())
          break;
        desc.push_back((double)b->second.size());
      }
      const mppi_status st = setBlobD(h, (blob_stem + "_structure").c_str(), desc, { (int)desc.size() });
      if (st == MPPI_ERR_INVALID_ARG && h->last_error.find("has no blob named") == std::string::npos)
        return st;  // the model knows the blob and refused this shape
    }
    MPPI_TRY(setBlobD(h, (blob_stem + "_weights").c_str(), blob, { (int)blob.size() }));
    return setBlobD(h, (blob_stem + "_output_weights").c_str(), out_blob, { (int)out_blob.size() });
  }
  if (k == "costmap")
  {
    // ARStandardCost::loadTrackData (ar_standard_cost.cu:84-142)
    for (const char* key : { "xBounds", "yBounds", "pixelsPerMeter", "channel0" })
      if (!d.count(key))
        return fail(h, MPPI_ERR_INVALID_ARG, std::string("mppi_load_npz: no key '") + key + "' in the archive");
    const float x_min = (float)d["xBounds"].data[0], x_max = (float)d["xBounds"].data[1];
    const float y_min = (float)d["yBounds"].data[0], y_max = (float)d["yBounds"].data[1];
    const float ppm = (float)d["pixelsPerMeter"].data[0];
    const int width = int((x_max - x_min) * ppm), height = int((y_max - y_min) * ppm);
    if (width <= 0 || height <= 0 || (size_t)width * height != d["channel0"].size())
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_load_npz: load track has invalid sizes");
    const float r_c1[3] = { 1.0f / (x_max - x_min), 0.0f, 0.0f };
    const float r_c2[3] = { 0.0f, 1.0f / (y_max - y_min), 0.0f };
    const float trs[3] = { -x_min / (x_max - x_min), -y_min / (y_max - y_min), 1.0f };
    if (h->model->setCostmapTransform(r_c1, r_c2, trs) != MPPI_OK)
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_load_npz: the model's cost has no costmap");
    return setBlobD(h, "costmap", d["channel0"].data, { height, width });
  }
  return fail(h, MPPI_ERR_INVALID_ARG, "mppi_load_npz: kind must be \"dynamics\", \"lstm\" or \"costmap\"");
}

mppi_status mppi_npz_read_array(const char* path, const char* key, double* out, size_t capacity, size_t* count, int* dims,
                                int* ndims)
{
  if (!path || !key)
    return MPPI_ERR_INVALID_ARG;
  std::map<std::string, npz::Array> d;
  std::string err;
  if (!npz::load(path, d, err))
  {
    g_create_error = "mppi_npz_read_array: " + err;
    return MPPI_ERR_INVALID_ARG;
  }
  auto it = d.find(key);
  if (it == d.end())
  {
    g_create_error = std::string("mppi_npz_read_array: no key '") + key + "'";
    return MPPI_ERR_INVALID_ARG;
  }
  if (count)
    *count = it->second.size();
  if (ndims)
    *ndims = (int)it->second.shape.size();
  if (dims)
    for (size_t i = 0; i < it->second.shape.size() && i < 8; i++)
      dims[i] = it->second.shape[i];
  if (out)
  {
    if (capacity < it->second.size())
      return MPPI_ERR_INVALID_ARG;
    std::copy(it->second.data.begin(), it->second.data.end(), out);
  }
  return MPPI_OK;
}

mppi_status mppi_set_seed(mppi_handle h, uint64_t seed)
{
  CHECK_HANDLE(h);
  h->cfg.seed = seed;
  h->generation = 0;  // controller.cu:200-207: offset reset on reseed
  return MPPI_OK;
}

/* ---------------------------------------------------------------- rocRAND host API ------------------------------- */

/** one rocrand_generate_normal per rollout launch into rocrand_eps_d: PHILOX4_32_10, seeded with the handle's seed, the
 *  offset advanced so that rank r of a sharded problem and every later generation draw disjoint stretches of the stream.
 *  (Statistically equivalent to the in-kernel Philox mode, not bit-identical: rocRAND orders its counter / Box-Muller
 *  differently — the reference's own sampler tests are statistical too, tests/sampling_distributions/.) */
static mppi_status rocrandFill(mppi_handle h)
{
  const size_t n = (epsFloatsPerIteration(h) + 1) & ~(size_t)1;  // the generator wants an even count
  if (!h->rocrand_lib)
  {
    h->rocrand_lib = dlopen("librocrand.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h->rocrand_lib)
      h->rocrand_lib = dlopen("librocrand.so", RTLD_NOW | RTLD_LOCAL);
    if (!h->rocrand_lib)
      return fail(h, MPPI_ERR_UNSUPPORTED, std::string("MPPI_NOISE_ROCRAND_HOST: cannot load librocrand.so: ") + dlerror());
    g_rocrand.create = (int (*)(void**, int))dlsym(h->rocrand_lib, "rocrand_create_generator");
    g_rocrand.destroy = (int (*)(void*))dlsym(h->rocrand_lib, "rocrand_destroy_generator");
    g_rocrand.set_seed = (int (*)(void*, unsigned long long))dlsym(h->rocrand_lib, "rocrand_set_seed");
    g_rocrand.set_offset = (int (*)(void*, unsigned long long))dlsym(h->rocrand_lib, "rocrand_set_offset");
    g_rocrand.set_stream = (int (*)(void*, hipStream_t))dlsym(h->rocrand_lib, "rocrand_set_stream");
    g_rocrand.normal = (int (*)(void*, float*, size_t, float, float))dlsym(h->rocrand_lib, "rocrand_generate_normal");
    if (!g_rocrand.create || !g_rocrand.destroy || !g_rocrand.set_seed || !g_rocrand.set_offset || !g_rocrand.set_stream ||
        !g_rocrand.normal)
      return fail(h, MPPI_ERR_UNSUPPORTED, "MPPI_NOISE_ROCRAND_HOST: librocrand.so lacks an expected entry point");
  }
  if (!h->rocrand_gen)
  {
    if (g_rocrand.create(&h->rocrand_gen, /*ROCRAND_RNG_PSEUDO_PHILOX4_32_10*/ 404) != 0 ||
        g_rocrand.set_stream(h->rocrand_gen, h->stream) != 0)
      return fail(h, MPPI_ERR_HIP, "rocrand_create_generator / rocrand_set_stream failed");
  }
  if (!h->rocrand_eps_d)
    HIP_TRY(h, hipMalloc((void**)&h->rocrand_eps_d, n * sizeof(float)));
  // stretch of the stream for (generation, rank): offsets count Philox outputs; a normal consumes one 32-bit output
  const unsigned long long per_gen = (unsigned long long)n * (unsigned long long)h->cfg.world_size;
  const unsigned long long offset = (unsigned long long)h->generation * per_gen + (unsigned long long)h->cfg.rank * n;
  if (g_rocrand.set_seed(h->rocrand_gen, (unsigned long long)h->cfg.seed) != 0 ||
      g_rocrand.set_offset(h->rocrand_gen, offset) != 0 || g_rocrand.normal(h->rocrand_gen, h->rocrand_eps_d, n, 0.0f, 1.0f) != 0)
    return fail(h, MPPI_ERR_HIP, "rocrand_generate_normal failed");
  return MPPI_OK;
}

/* ---------------------------------------------------------------- roctx ranges --------------------------------------- */
/**
 * Marker ranges around the enqueue of the rollout, merge and post-processing kernels (the reference has no
 * profiler ranges; `rocprofv3 --marker-trace --kernel-trace` then attributes the kernels of an iteration).  Opt-in:
 * MPPI_AMD_ROCTX=1 — libroctx64 is dlopen'ed on first use; when the variable is unset a range is one predictable branch.
 */
namespace
{
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
static const Roctx& roctx()
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
}  // namespace

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

static mppi_status launchCombine(mppi_handle h, const float* records, int num_records, int finalize, float* record_out,
                                 int k_total, bool world_major = false, const unsigned* wait_flags = nullptr,
                                 unsigned wait_seq = 0, const kernels::PostTargets* post = nullptr)
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

static inline bool tsallisActive(const mppi_handle_s* h);
/** may the NEXT rollout launch merge the previous launch's records itself (rolloutPipelineKernel STREAM_MERGE)? */
static bool streamMergeApplies(const mppi_handle_s* h)
{
  return h->stream_merge_enabled && h->pipeline && h->bz == 1 && h->D == 1 && h->by == 1 && h->bx == 64 && !h->rows_in_hbm &&
         !exchangeActive(h) && h->noise_source == MPPI_NOISE_PHILOX_FUSED && h->reduction_mode == MPPI_REDUCTION_FUSED &&
         !tsallisActive(h) && h->cfg.controller != MPPI_CONTROLLER_ROBUST && (h->TC & 3) == 0 && h->num_blocks <= 256 &&
         h->model->supportsStreamedMerge();
}
static mppi_status launchCombine(mppi_handle h, const float* records, int num_records, int finalize, float* record_out,
                                 int k_total, bool world_major, const unsigned* wait_flags, unsigned wait_seq,
                                 const kernels::PostTargets* post);
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
/** the records of the last rollout launch are still un-merged: merge them now (combineKernel -> mean_d, stats_d) */
static mppi_status flushMerge(mppi_handle h)
{
  if (!h->pending_records_d)
    return MPPI_OK;
  const float* rec = h->pending_records_d;
  h->pending_records_d = nullptr;
  return launchCombine(h, rec, h->num_blocks, 1, nullptr, h->cfg.num_rollouts, false, nullptr, 0, nullptr);
}

static mppi_status launchRollout(mppi_handle h, int iteration, int stride)
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
  if (h->pending_records_d)
  {
    if (!streamMergeApplies(h))
      MPPI_TRY(flushMerge(h));  // (a setting changed between two launches: merge the pending records the ordinary way)
    else
    {
      a.prev_records_d = h->pending_records_d;
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

typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static nccl_allgather_fn g_ncclAllGather = nullptr;

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

static inline bool tsallisActive(const mppi_handle_s* h)
{  // colored_mppi_controller.cu:198: the exponential weights unless BOTH parameters are set
  return h->cfg.controller == MPPI_CONTROLLER_COLORED && h->tsallis_gamma != 0.0f && h->tsallis_r != 0.0f;
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

static mppi_status iterationLocal(mppi_handle h, int iteration, int stride)
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

static mppi_status iterationMerge(mppi_handle h)
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

static mppi_status iteration(mppi_handle h, int it, int stride)
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

static void parseStats(mppi_handle h, const float* st);
static mppi_status fetchStats(mppi_handle h)
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

static bool allFinite(const std::vector<float>& v)
{
  for (float f : v)
    if (!std::isfinite(f))
      return false;
  return true;
}

/** smoothing / state trajectories / constraints for the D systems in ctrl_in_d, results to the host vectors */
static mppi_status finalize(mppi_handle h, const float* ctrl_in_d, int smooth_mask, int constrain_mask,
                            std::vector<float>* ctrl_out[2], std::vector<float>* state_out[2], int num_systems = 0)
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
static void parseStats(mppi_handle h, const float* st)
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
    MPPI_TRY(flushMerge(h));  // the last iteration's records (streamed merge): everything below reads mean_d / stats_d
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
    const mppi_status st = h->model->launchFinalize(1, a, h->stream, err);
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
static mppi_status modelStepInPlace(mppi_handle h, float* x, float* u, float dt, int enforce)
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

/* ---------------------------------------------------------------- multi-GPU -------------------------------------- */
mppi_status mppi_get_exchange_buffers(mppi_handle h, void** send, void** recv, size_t* floats_per_rank)
{
  CHECK_HANDLE(h);
  if (send)
    *send = h->send_d;
  if (recv)
    *recv = h->recv_d;
  if (floats_per_rank)
    *floats_per_rank = (size_t)h->D * h->PS;
  return MPPI_OK;
}
mppi_status mppi_read_send_record(mppi_handle h, float* out)
{
  CHECK_HANDLE(h);
  if (!out)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(out, h->send_d, sizeof(float) * h->D * h->PS, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}
mppi_status mppi_write_recv_records(mppi_handle h, const float* in)
{
  CHECK_HANDLE(h);
  if (!in)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(h->recv_d, in, sizeof(float) * h->cfg.world_size * h->D * h->PS, hipMemcpyHostToDevice,
                            h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // `in` is the caller's
  return MPPI_OK;
}
mppi_status mppi_iteration_local(mppi_handle h)
{
  CHECK_HANDLE(h);
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  // the caller drives the optimisation loop: its iteration index (std_dev_decay) restarts with mppi_upload_state
  MPPI_TRY(iterationLocal(h, h->external_iteration++, h->last_stride));
  return flushMerge(h);  // (a caller-driven loop sees every iteration's mean: no streamed merge across its calls)
}
mppi_status mppi_iteration_merge(mppi_handle h)
{
  CHECK_HANDLE(h);
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  return iterationMerge(h);
}

static void* loadRccl(std::string& err)
{
  static void* lib = nullptr;
  if (lib)
    return lib;
  // The communicator must live on the SAME HIP runtime as this library's streams and buffers.  A host application may
  // carry a second ROCm stack (PyTorch wheels bundle their own libamdhip64 / librccl), and a plain dlopen("librccl.so")
  // would hand back that copy.  So: first the librccl that sits next to the libamdhip64 this library is linked to (by
  // absolute path), then the usual names.
  std::vector<std::string> candidates;
  Dl_info info{};
  if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname)
  {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos)
    {
      dir.resize(slash);
      candidates.push_back(dir + "/librccl.so.1");
      candidates.push_back(dir + "/librccl.so");
    }
  }
  candidates.push_back("/opt/rocm/lib/librccl.so.1");
  candidates.push_back("/opt/rocm/lib/librccl.so");
  candidates.push_back("librccl.so.1");
  candidates.push_back("librccl.so");
  for (const std::string& n : candidates)
  {
    lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (lib)
      return lib;
  }
  err = std::string("cannot dlopen librccl: ") + dlerror();
  return nullptr;
}

/* ---------------------------------------------------------------- P2P mailbox exchange ---------------------------- */
static mppi_status ensureMailbox(mppi_handle h)
{
  if (h->mbox_d)
    return MPPI_OK;
  const int world = h->cfg.world_size;
  if (world > 16)
    return fail(h, MPPI_ERR_UNSUPPORTED, "P2P mailbox exchange supports up to 16 ranks");
  const size_t dps = (size_t)h->D * h->PS;
  // records | flags [2][world], ticket counter (+ 3 words of padding) | aux arrays [2][MAILBOX_AUX_FLOATS] | aux flags [2][world]
  h->mbox_aux_off = 2 * world * dps + (size_t)(2 * world + 4);  // in 4-byte words from the base
  h->mbox_aux_off = (h->mbox_aux_off + 3) & ~(size_t)3;
  h->mbox_bytes = sizeof(float) * (h->mbox_aux_off + 2 * (size_t)kernels::MAILBOX_AUX_FLOATS + 2 * (size_t)world);
  h->mbox_bytes = (h->mbox_bytes + 4095) & ~(size_t)4095;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  // uncached device memory where the runtime offers it (the mailbox is written by other agents); every access to it is a
  // system-scope atomic anyway, so ordinary device memory is a correct fallback
  hipError_t e = hipExtMallocWithFlags((void**)&h->mbox_d, h->mbox_bytes, hipDeviceMallocUncached);
  h->mbox_uncached = (e == hipSuccess);
  if (e != hipSuccess)
  {
    (void)hipGetLastError();
    HIP_TRY(h, hipMalloc((void**)&h->mbox_d, h->mbox_bytes));
  }
  HIP_TRY(h, hipMemsetAsync(h->mbox_d, 0, h->mbox_bytes, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

/**
 * A new exchange session starts at sequence number 1 again, and the merge kernel waits for flag == sequence number: flags
 * and records left by an earlier session (one that ended after 1-3 iterations would match sequence 1 / 2 of the new one)
 * are cleared here.  Called where a session begins BEFORE a peer of the new session can reach the mailbox: when its IPC
 * handle is exported (peers map it after that), and by mppi_p2p_connect_local (in-process ranks connect before their first
 * exchange).
 */
static mppi_status resetMailboxSession(mppi_handle h)
{
  if (!h->mbox_d || (h->xseq == 0 && h->aseq == 0))
    return MPPI_OK;  // fresh (zeroed at allocation) or never used since the last reset
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemsetAsync(h->mbox_d, 0, h->mbox_bytes, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->xseq = 0;
  h->aseq = 0;
  return MPPI_OK;
}

mppi_status mppi_p2p_mailbox_handle(mppi_handle h, void* out_bytes, size_t capacity, size_t* nbytes)
{
  CHECK_HANDLE(h);
  if (!out_bytes || capacity < sizeof(hipIpcMemHandle_t))
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_p2p_mailbox_handle: buffer too small (needs 64 bytes)");
  MPPI_TRY(ensureMailbox(h));
  // Exporting the handle has no side effect on a LIVE session (a caller that asks twice, a peer that maps late): the mailbox
  // is only cleared when no session is connected — a new one starts with mppi_p2p_reset (or on a handle that never ran)
  if (!h->p2p_ready)
    MPPI_TRY(resetMailboxSession(h));
  hipIpcMemHandle_t ipc;
  hipError_t e = hipIpcGetMemHandle(&ipc, h->mbox_d);
  if (e != hipSuccess && h->mbox_uncached)
  {  // this runtime does not export uncached allocations: fall back to ordinary device memory
    (void)hipGetLastError();
    (void)hipFree(h->mbox_d);
    h->mbox_d = nullptr;
    HIP_TRY(h, hipMalloc((void**)&h->mbox_d, h->mbox_bytes));
    h->mbox_uncached = false;
    HIP_TRY(h, hipMemsetAsync(h->mbox_d, 0, h->mbox_bytes, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    e = hipIpcGetMemHandle(&ipc, h->mbox_d);
  }
  if (e != hipSuccess)
    return fail(h, MPPI_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e) +
                                     " (multi-process GPU sharing needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver)");
  memcpy(out_bytes, &ipc, sizeof(ipc));
  if (nbytes)
    *nbytes = sizeof(ipc);
  return MPPI_OK;
}

/** the exchange-failure mark stats_d[z][6] is sticky on the device (no kernel clears it, reduce_kernels.hpp: combineWave): a new
 *  session starts without it */
static mppi_status clearExchangeFailure(mppi_handle h)
{
  h->exchange_failed = false;
  if (!h->stats_d)
    return MPPI_OK;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  for (int z = 0; z < h->D; z++)
    HIP_TRY(h, hipMemsetAsync(h->stats_d + (size_t)z * kernels::STATS_STRIDE + 6, 0, sizeof(float), h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

mppi_status mppi_p2p_reset(mppi_handle h)
{
  CHECK_HANDLE(h);
  h->p2p_ready = false;
  MPPI_TRY(clearExchangeFailure(h));
  return resetMailboxSession(h);
}

mppi_status mppi_p2p_connect(mppi_handle h, const void* handles, size_t stride_bytes)
{
  CHECK_HANDLE(h);
  const int world = h->cfg.world_size;
  if (!handles || stride_bytes < sizeof(hipIpcMemHandle_t))
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_p2p_connect: handles[world] with a stride of at least 64 bytes expected");
  MPPI_TRY(ensureMailbox(h));
  /* A session's sequence numbers start at 1 and the merge waits for flag == sequence number, so a mailbox that still holds the
   * flags and records of an earlier session would let this one pass its waits early (stale or half-written peer records merged
   * silently).  The mailbox cannot be cleared HERE — a peer of the new session that connected first may already have posted
   * into it — only before its handle is exported, which mppi_p2p_mailbox_handle does on a handle without a live session.  A
   * live or used session therefore has to be ended explicitly first: mppi_p2p_reset, then export, then connect. */
  if (h->p2p_ready || h->xseq != 0 || h->aseq != 0)
    return fail(h, MPPI_ERR_STATE, "mppi_p2p_connect: this handle has a live (or used) exchange session; call mppi_p2p_reset on "
                                   "every rank, export the mailbox handles again (mppi_p2p_mailbox_handle) and then connect");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  for (int p = 0; p < world; p++)
  {
    if (p == h->cfg.rank)
    {
      h->peer_mbox[p] = h->mbox_d;
      continue;
    }
    if (h->peer_opened[p] && h->peer_mbox[p])
    {  // a reconnect: the mapping of the previous session goes first
      (void)hipIpcCloseMemHandle(h->peer_mbox[p]);
      h->peer_opened[p] = false;
      h->peer_mbox[p] = nullptr;
    }
    hipIpcMemHandle_t ipc;
    memcpy(&ipc, (const char*)handles + (size_t)p * stride_bytes, sizeof(ipc));
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, ipc, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess)
      return fail(h, MPPI_ERR_COMM, "hipIpcOpenMemHandle for rank " + std::to_string(p) + ": " + hipGetErrorString(e));
    h->peer_mbox[p] = (float*)ptr;
    h->peer_opened[p] = true;
  }
  h->xseq = 0;
  h->aseq = 0;
  MPPI_TRY(clearExchangeFailure(h));
  h->p2p_ready = true;
  return MPPI_OK;
}

mppi_status mppi_p2p_connect_local(mppi_handle h, const mppi_handle* peers)
{
  CHECK_HANDLE(h);
  const int world = h->cfg.world_size;
  if (!peers)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_p2p_connect_local: null");
  MPPI_TRY(ensureMailbox(h));
  MPPI_TRY(resetMailboxSession(h));
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  for (int p = 0; p < world; p++)
  {
    mppi_handle q = peers[p];
    if (!q || q->cfg.world_size != world || q->cfg.rank != p || q->D != h->D || q->PS != h->PS)
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_p2p_connect_local: peers[p] must be the handle of rank p of the same problem");
    if (q != h)
    {
      std::lock_guard<std::recursive_mutex> peer_lock(q->mu);
      MPPI_TRY(ensureMailbox(q) == MPPI_OK ? MPPI_OK : fail(h, MPPI_ERR_HIP, "peer mailbox allocation failed"));
      HIP_TRY(h, hipSetDevice(h->cfg.device));
      if (q->cfg.device != h->cfg.device)
      {
        int can = 0;
        HIP_TRY(h, hipDeviceCanAccessPeer(&can, h->cfg.device, q->cfg.device));
        if (!can)
          return fail(h, MPPI_ERR_COMM, "no peer access between device " + std::to_string(h->cfg.device) + " and " +
                                            std::to_string(q->cfg.device));
        const hipError_t e = hipDeviceEnablePeerAccess(q->cfg.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          return fail(h, MPPI_ERR_COMM, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
      }
    }
    h->peer_mbox[p] = q->mbox_d;
  }
  h->xseq = 0;
  h->aseq = 0;
  MPPI_TRY(clearExchangeFailure(h));
  h->p2p_ready = true;
  return MPPI_OK;
}

mppi_status mppi_rccl_unique_id(void* out_bytes, size_t capacity, size_t* nbytes)
{
  if (!out_bytes || capacity < 128)
    return MPPI_ERR_INVALID_ARG;
  std::string err;
  void* lib = loadRccl(err);
  if (!lib)
    return fail(nullptr, MPPI_ERR_COMM, err);
  typedef int (*fn_t)(void*);
  fn_t f = (fn_t)dlsym(lib, "ncclGetUniqueId");
  if (!f)
    return fail(nullptr, MPPI_ERR_COMM, "ncclGetUniqueId not found");
  const int rc = f(out_bytes);  // ncclUniqueId is 128 bytes
  if (nbytes)
    *nbytes = 128;
  return rc == 0 ? MPPI_OK : fail(nullptr, MPPI_ERR_COMM, "ncclGetUniqueId failed");
}

mppi_status mppi_comm_init_rccl(mppi_handle h, const void* unique_id, size_t nbytes)
{
  CHECK_HANDLE(h);
  if (!unique_id || nbytes != 128)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_comm_init_rccl: unique id must be 128 bytes");
  std::string err;
  void* lib = loadRccl(err);
  if (!lib)
    return fail(h, MPPI_ERR_COMM, err);
  struct Id
  {
    char b[128];
  } id;
  memcpy(id.b, unique_id, 128);
  typedef int (*init_fn)(void**, int, Id, int);
  init_fn f = (init_fn)dlsym(lib, "ncclCommInitRank");
  g_ncclAllGather = (nccl_allgather_fn)dlsym(lib, "ncclAllGather");
  if (!f || !g_ncclAllGather)
    return fail(h, MPPI_ERR_COMM, "ncclCommInitRank / ncclAllGather not found in librccl");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int rc = f(&h->comm, h->cfg.world_size, id, h->cfg.rank);
  if (rc != 0)
    return fail(h, MPPI_ERR_COMM, "ncclCommInitRank failed with code " + std::to_string(rc));
  h->rccl_lib = lib;
  return MPPI_OK;
}

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

}  // extern "C"
