/**
 * rollout_pipeline_kernel.hpp — the rollout as a three-stage pipeline of role-specialised waves (MI355X design).
 *
 * Why: at the benchmark sizes (K = 16384 rollouts = 256 waves on 1024 SIMDs) the fused rollout is bound by the LENGTH
 * of one wave's serial instruction stream (T steps x ~300 instructions x ~2 ns, dependent or not —
 * tools/ubench/op_latency.hip), not by throughput: three of the four SIMDs of every CU idle.  The reference's plugin split happens to cut a step into three parts with a one-way data
 * flow — SamplingDistribution -> Dynamics -> Cost (reference call order: core/mppi_common.cu:98-137) — and only the
 * Dynamics part carries the step-to-step dependence.  So a block of 64 rollouts runs as three waves on three SIMDs:
 *
 *   wave S (sampler)   for every t: draw eps (Philox quad / pre-filled row), apply the setGaussianControls rule,
 *                      store the sample in the rollout's LDS row                      -> sampling->drawQuad / shape...
 *   wave D (dynamics)  for every t: fetch the sample, step, push the output y_t into an LDS ring  -> dynamics->...
 *                      (enforceConstraints + write-back of the clamped control, mppi_common.cu:110-117, happen here only
 *                      when the plugin's constraints depend on the state; otherwise in the sampler waves)
 *   wave C (cost)      for every t: pop y_t, read u_t, running += computeRunningCost + likelihoodRatioCost
 *                                                                              -> costs->..., sampling->...
 *
 * The stages are decoupled by monotonic progress counters in LDS (one writer each, polled with s_sleep by the consumer,
 * release/acquire at workgroup scope); S may run arbitrarily far ahead (the rows hold the whole horizon), C lags D by at
 * most RING steps (back-pressure).  The critical path per step is wave D's instructions (Cartpole: 83) instead of ~300.
 * Blocks with one set of role waves run TWO sampler waves on alternate trips of four steps (the CU's fourth SIMD).
 * Every rollout is still evaluated with exactly the arithmetic of rolloutKernel (same plugin methods, same order of the
 * cost additions), so the results are bit-identical; the epilogue is shared (blockSoftminEpilogue).
 *
 * Requirements on the plugins: one lane per rollout (no threadIdx.y cooperation) and no block-wide barrier inside the
 * per-step methods (mppi::lane_sync() is fine — it is a no-op for blockDim.y == 1).  Models are registered for this
 * variant explicitly (csrc/models.hpp).
 *
 * Launch: grid = ceil(K / 64), block = (pipelineBlockX(), 1, BZ): waves 0 / 1 / 2 = sampler / dynamics / cost, wave 3 (blocks
 * with one z slice) = second sampler.  FOLD_Z: see the kernel.
 */
#ifndef MPPI_AMD_ROLLOUT_PIPELINE_KERNEL_HPP_
#define MPPI_AMD_ROLLOUT_PIPELINE_KERNEL_HPP_

#include <type_traits>
#include "rollout_kernel.hpp"
#include "merge_wave.hpp"
#include "kernarg_view.hpp"

namespace mppi
{
namespace kernels
{
constexpr int PIPE_ROLES = 3;
/** outputs buffered between the dynamics and the cost wave (steps): 32 for the small analytic models; wide output
 *  vectors (the RACER models carry 28 floats) get a shorter ring — 8 steps = two groups of the dynamics wave */
__host__ __device__ constexpr int pipeRingSteps(int output_dim)
{
  return output_dim <= 8 ? 32 : (output_dim <= 16 ? 16 : 8);
}

/** fold_z: the systems of a rollout are folded into the LANE dimension (64 / bz rollouts x bz systems per wave) instead of
 *  the workgroup's z dimension — one ring and one counter set per block, half the sample rows per block */
template <class DYN_T, class COST_T, class SAMPLING_T>
__host__ inline size_t pipelineSharedBytes(const DYN_T& dyn, const COST_T& cost, const SAMPLING_T& smp, int bz,
                                           bool fold_z = false)
{
  const int slots = fold_z ? 64 : 64 * bz;
  const int rings = fold_z ? 1 : bz;
  size_t n = 0;
  n += calcClassSharedMemSize(&dyn, slots);
  n += calcClassSharedMemSize(&cost, slots);
  n += calcClassSharedMemSize(&smp, slots);
  n += sizeof(float) * 2 * math::nearest_multiple_4(slots);                     // cost_s, w_s
  // output ring [z][slot][i][lane]; with the rows in HBM the clamped control travels through it as well
  n += sizeof(float) * (size_t)rings * pipeRingSteps(DYN_T::OUTPUT_DIM) *
       (DYN_T::OUTPUT_DIM + (smp.rows_global_d_ ? DYN_T::CONTROL_DIM : 0)) * 64;
  n += sizeof(int) * 4 * 4 * rings;                                              // progress counters (padded)
  // STREAM_MERGE: the control mean the sampler waves merge for themselves, [T][C], for the cost wave's likelihood-ratio term
  n += sizeof(float) * math::nearest_multiple_4(smp.params_.num_timesteps * DYN_T::CONTROL_DIM);
  return n;
}

/** progress counters live in LDS: address-space-3 pointers keep the polls on ds_read instead of flat loads */
typedef volatile __attribute__((address_space(3))) int* lds_counter_t;
typedef int pipe_int4 __attribute__((ext_vector_type(4)));
typedef volatile __attribute__((address_space(3))) pipe_int4* lds_counter4_t;

/**
 * Blocks until *ctr >= need.  `cached` is the consumer's last observed value: the producer usually runs ahead, so most
 * calls return without touching LDS.
 */
__device__ inline void pipeWait(lds_counter_t ctr, const int need, int& cached)
{
  if (cached >= need)
    return;
  int v = __builtin_amdgcn_readfirstlane(*ctr);
  while (v < need)
  {
    __builtin_amdgcn_s_sleep(1);
    v = __builtin_amdgcn_readfirstlane(*ctr);
  }
  cached = v;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ inline void pipePublish(lds_counter_t ctr, const int value, const int lane)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0)
    *ctr = value;
}
/**
 * pipePublish for a producer whose data went to LDS ONLY (ring slots, LDS sample rows).  The LDS unit executes the DS
 * instructions of a wave in issue order (that order is what lgkmcnt counts), so the counter store below cannot become visible
 * before the data stores in front of it: no s_waitcnt is needed, only the compiler must keep the order.  The release fence of
 * pipePublish cost the dynamics wave of rolloutPipelineKernel an exposed LDS write round trip per trip (in-kernel timers,
 * profiles/r04_cartpole_pipe_timing*.json).  NOT for data stored to global memory (the HBM-row variants' sampler waves).
 */
__device__ inline void pipePublishLds(lds_counter_t ctr, const int value, const int lane)
{
  asm volatile("" ::: "memory");
  if (lane == 0)
    *ctr = value;
  asm volatile("" ::: "memory");
}
/* ---- A/B instrumentation (tools/pipe_timing.py; never defined in a product build): where the role waves of
 * rolloutPipelineRepKernel spend their time.  Every wave of the first PIPE_TIMING_BLOCKS blocks accumulates s_memtime ticks
 * (constant 100 MHz on gfx950: 10 ns — coarse per event, unbiased over the hundreds of events of a launch) per category. */
#if defined(MPPI_PIPE_TIMING)
constexpr int PIPE_TIMING_BLOCKS = 256, PIPE_TIMING_WAVES = 24, PIPE_TIMING_SLOTS = 8;
static __device__ unsigned long long g_pipe_timing[PIPE_TIMING_BLOCKS * PIPE_TIMING_WAVES * PIPE_TIMING_SLOTS];
struct PipeTimer
{
  unsigned long long acc[PIPE_TIMING_SLOTS] = {};
  unsigned long long t0 = 0;
  __device__ inline void start()
  {
    t0 = __builtin_amdgcn_s_memtime();
  }
  __device__ inline void stop(const int slot)
  {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    acc[slot] += t1 - t0;
    t0 = t1;
  }
  /** per-trip record of one wave (the dynamics wave of rolloutPipelineKernel): row 4 + kind of the block's table, entry trip */
  __device__ inline void trip(const int block, const int kind, const int trip_idx, const int lane, const unsigned long long v) const
  {
    if (block < PIPE_TIMING_BLOCKS && lane == 0 && trip_idx < 4 * PIPE_TIMING_SLOTS && kind < 5)
      g_pipe_timing[(block * PIPE_TIMING_WAVES + 4 + 4 * kind) * PIPE_TIMING_SLOTS + trip_idx] = v;
  }
  __device__ inline void flush(const int block, const int wave, const int lane) const
  {
    if (block < PIPE_TIMING_BLOCKS && wave < PIPE_TIMING_WAVES && lane == 0)
      for (int i = 0; i < PIPE_TIMING_SLOTS; i++)
        g_pipe_timing[(block * PIPE_TIMING_WAVES + wave) * PIPE_TIMING_SLOTS + i] = acc[i];
  }
};
#define PIPE_T(x) x
#else
#define PIPE_T(x)
#endif

/** sampler waves of blocks that hold ONE set of role waves (BZ == 1, or the folded Tube variant): the draw (Philox rounds
 *  + two Box-Muller pairs per quad, ~320 dependent instructions) is as long as the dynamics of four cart-pole steps, and a
 *  block of three waves leaves the CU's fourth SIMD idle — a second sampler wave taking alternate trips is free */
constexpr int PIPE_SAMPLERS = 2;
/** workgroup x extent of rolloutPipelineKernel */
__host__ __device__ constexpr int pipelineBlockX(int bz, bool fold_z)
{
  return (bz == 1 || fold_z) ? 64 * (2 + PIPE_SAMPLERS) : 64 * PIPE_ROLES;
}

/** ROWS_HBM: the sample rows of the block in the sampler's HBM buffer (long horizons) — see rolloutPipelineRepKernel: the
 *  dynamics wave fetches the next trip's samples while it computes the current one, the clamped control reaches the cost
 *  wave through the output ring */
/**
 * STREAM_MERGE (round 4; one system, plain draw, rows in LDS): the sampler waves merge the PREVIOUS iteration's block records
 * (args.prev_records_d) into the control mean themselves, four columns — the steps of one trip — at a time, with the arithmetic
 * of the merge kernel (merge_wave.hpp: the same functions, the same bits): each sampler wave loads the 256 record tails once
 * (rho, the scale factors and eta stay in its registers) and, one of its own trips ahead, the four 16-byte column quads a lane
 * owns; a trip then costs four DPP all-reduces and four divisions on top of its draw.  The merge launch between two
 * iterations (3.3 us of body + a 1.6 us boundary on a 27 us Cartpole iteration) disappears; what remains of it is that the
 * first trip waits for the records' memory round trip instead of only for its own draw.  The last iteration of a sequence is
 * merged by combineKernel as before (the engine flushes before anything reads the mean or the statistics).
 */
template <class DYN_T, class COST_T, class SAMPLING_T, int BZ, bool DRAW_IN_LOOP, bool FOLD_Z = false, bool ROWS_HBM = false,
          bool STREAM_MERGE = false>
__global__ void __launch_bounds__(pipelineBlockX(BZ, FOLD_Z) * (FOLD_Z ? 1 : BZ))
    rolloutPipelineKernel(DYN_T dynamics_obj, COST_T costs_obj, SAMPLING_T sampling_obj, const RolloutArgs args)
{
  static_assert(!STREAM_MERGE || (BZ == 1 && !FOLD_Z && !ROWS_HBM && DRAW_IN_LOOP), "STREAM_MERGE: one system, plain draw, LDS rows");
  // FOLD_Z (Tube: BZ == 2): a wave carries 32 rollouts x 2 systems in its 64 lanes.  Half the sample rows per block
  // (two systems with T*C floats per rollout each are what overflows the LDS at 64 rollouts), twice the blocks — at
  // K = 8192 exactly one block per CU — and the two systems of a rollout advance in the same instruction stream.
  static_assert(!FOLD_Z || 64 % BZ == 0, "systems must divide the wave");
  constexpr int BX = FOLD_Z ? 64 / BZ : 64;
  constexpr int WZ = FOLD_Z ? 1 : BZ;  // z extent of the workgroup
  // The folded variant draws with PAIRS of lanes: the two systems of a rollout see the same noise, so lane z = 0 draws
  // Philox quad q, lane z = 1 quad q + 1, and one v_permlane32_swap per value hands both quads to both lanes — a trip is
  // 8 row elements (8 / C steps) for the price of one draw.
  // A sampler trip is always 4 steps (= C Philox quads = one group of the dynamics wave); NS sampler waves take
  // alternate trips.
  constexpr bool PAIR_DRAW = FOLD_Z && DRAW_IN_LOOP && BZ == 2 && DYN_T::CONTROL_DIM == 2;
  constexpr int WX = pipelineBlockX(BZ, FOLD_Z);
  constexpr int NS = WX / 64 - 2;  // sampler waves
  __builtin_assume(__builtin_amdgcn_workgroup_size_x() == WX);
  __builtin_assume(__builtin_amdgcn_workgroup_size_y() == 1);
  __builtin_assume(__builtin_amdgcn_workgroup_size_z() == WZ);
  __builtin_assume(__builtin_amdgcn_workitem_id_x() < WX);
  __builtin_assume(__builtin_amdgcn_workitem_id_y() == 0);
  __builtin_assume(__builtin_amdgcn_workitem_id_z() < WZ);

  PIPE_T(PipeTimer tm; tm.start(); const unsigned long long t_entry = wall_clock64();)
  DYN_T* dynamics = &dynamics_obj;
  COST_T* costs = &costs_obj;
  SAMPLING_T* sampling = &sampling_obj;
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  constexpr int PIPE_RING = pipeRingSteps(O);
  constexpr int SLOTS = BX * BZ;
  constexpr int NTHREADS = WX * WZ;
  // state-independent control constraints are applied by the sampler waves, which have issue slots to spare
  constexpr bool SMP_CONSTRAINS = !DYN_T::CONSTRAINTS_DEPEND_ON_STATE;

  const int tid_x = (int)__builtin_amdgcn_workitem_id_x();
  const int wave_x = __builtin_amdgcn_readfirstlane(tid_x >> 6);  // wave-uniform
  // waves 0 / 1 / 2: sampler / dynamics / cost; wave 3 (if present): second sampler
  const int role = wave_x < PIPE_ROLES ? wave_x : 0;
  const int smp_id = wave_x < PIPE_ROLES ? 0 : wave_x - PIPE_ROLES + 1;
  const int lane = tid_x & 63;
  const int thread_idx = FOLD_Z ? lane % BX : lane;
  const int thread_idz = FOLD_Z ? lane / BX : (int)__builtin_amdgcn_workitem_id_z();
  const int ring_z = FOLD_Z ? 0 : thread_idz;  // which ring / counter set this thread uses
  const int block_idx = (int)blockIdx.x;
  const int global_idx = BX * block_idx + thread_idx;
  const int shared_idx = BX * thread_idz + thread_idx;
  const int distribution_idx = thread_idz;
  sampling->setNoiseStream(distribution_idx);  // Philox stream of this thread's draws (independent-noise option)
  const int tid_flat = tid_x + WX * ring_z;
  const int num_timesteps = args.num_timesteps;
  const int num_rollouts = args.num_rollouts;
  const float dt = args.dt;
  const bool valid = global_idx < num_rollouts;
  const int nrows = min(BX, num_rollouts - BX * block_idx);

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s_shared = reinterpret_cast<float*>(smem_raw);
  float* theta_c_shared = theta_s_shared + calcClassSharedMemSize(dynamics, SLOTS) / (int)sizeof(float);
  float* theta_d_lds = theta_c_shared + calcClassSharedMemSize(costs, SLOTS) / (int)sizeof(float);
  float* theta_d_shared = theta_d_lds;
  if constexpr (ROWS_HBM)
  {
    theta_d_shared = sampling->blockRows(theta_d_lds, (int)blockIdx.x, SLOTS);
    sampling->setStagingBase(theta_d_lds);
  }
  constexpr int F = O + (ROWS_HBM ? C : 0);  // floats per step and rollout in the output ring
  float* cost_s = theta_d_lds + calcClassSharedMemSize(sampling, SLOTS) / (int)sizeof(float);
  float* w_s = cost_s + math::nearest_multiple_4(SLOTS);
  float* ring_all = w_s + math::nearest_multiple_4(SLOTS);
  float* ring = ring_all + (size_t)ring_z * PIPE_RING * F * 64;  // [slot][i][lane]
  lds_counter_t counters =
      (lds_counter_t)(reinterpret_cast<int*>(ring_all + (size_t)WZ * PIPE_RING * F * 64) + 16 * ring_z);
  float* mean_row_s = ring_all + (size_t)WZ * PIPE_RING * F * 64 + 16 * WZ;  // [T][C] (STREAM_MERGE), behind the counters
  // the three counters the dynamics wave consumes sit in one 16-byte word group: it reads them with a single ds_read_b128
  lds_counter_t smp_prog = counters + 0;   // steps whose shaped sample is in the row
  lds_counter_t smp_prog1 = counters + 1;  // second sampler wave (trips 1, 3, 5, ...)
  lds_counter_t cost_prog = counters + 2;  // steps the cost wave has consumed
  lds_counter_t dyn_prog = counters + 4;   // steps whose output is in the ring
  static_assert(NS <= 2, "one spare counter");

  sampling->setThreadMapping(shared_idx, BX, BZ);

  float x[S], x_next[S], xdot[S], u[C], y[O];
  int crash_status = 0;
#pragma unroll
  for (int i = 0; i < S; i++)
  {
    x[i] = 0.0f;  // the initial state is fetched behind the samplers' first trip (below): nothing in front of it reads the state
    xdot[i] = 0.0f;
    x_next[i] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < C; i++)
    u[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < O; i++)
    y[i] = 0.0f;
  /* One trip of a sampler wave (plain draw): four steps = C Philox quads, shaped by the setGaussianControls rule, clamped when
   * the constraints do not depend on the state, stored in the rollout's row.  The FIRST trip of every sampler wave runs here,
   * in front of the block's barriers: it needs nothing but kernel arguments, and the dynamics wave — which can do nothing
   * until steps 0..3 exist — would otherwise sit through prologue + draw in sequence (0.76 + 1.0 us of a 26 us launch,
   * in-kernel timers of round 4); now the other waves' prologue and the first draw overlap. */
  constexpr int SMP_STEPS = 4;
  constexpr int TRIP_COLS = SMP_STEPS * C;                 // columns of the mean a trip shapes with
  constexpr int TRIP_GROUPS = TRIP_COLS / MERGE_COLS;      // = C quads of four columns
  // STREAM_MERGE state of a sampler wave: what the record tails gave (once) and the column quads of its NEXT trip (in flight)
  // The loaded quads stay RAW until the trip that consumes them: a select on a padding record next to the load is a use, and
  // the compiler waits for the load right there — the tails, the first quads and every trip's "prefetch" each cost their own
  // exposed memory round trip that way (ISA of round 4: three back-to-back vmcnt(0) in front of the first draw).
  typedef float merge_f4 __attribute__((ext_vector_type(4)));
  MergeTails mt;
  merge_f4 vq[STREAM_MERGE ? TRIP_GROUPS : 1][MERGE_LANE_RECORDS];
  // the records' tails (rho_b, eta_b; the sum of w^2 behind them is the statistics' business): loaded at kernel entry.  Eight
  // bytes, not the whole 16-byte tail: the register allocator hands the dead half of a wider load's destination to the next
  // load while the first is still in flight, and the hazard costs a wait in between (ISA of round 4)
  typedef float merge_f2 __attribute__((ext_vector_type(2)));
  merge_f2 tailq[MERGE_LANE_RECORDS];
  const int TC_all = num_timesteps * C;
  auto load_trip_columns = [&](const int t) {
    if constexpr (STREAM_MERGE)
    {
      // the transposed copy of the records (RolloutArgs::records_t_d): quad q of record b at [q][b][4] — a load instruction's 64
      // lanes read 1 KB of contiguous memory
      static_assert(MERGE_COLS == 4, "the transposed copy is laid out in column quads");
#pragma unroll
      for (int g = 0; g < TRIP_GROUPS; g++)
      {
        const int col0 = t * C + MERGE_COLS * g;  // T*C is a multiple of 4 (engine): a quad lies inside the record or behind it
#pragma unroll
        for (int i = 0; i < MERGE_LANE_RECORDS; i++)
        {
          const int b = lane + 64 * i;
          const bool ok = b < args.prev_num_records && col0 < TC_all;  // (else: any valid address; the value is not used)
          vq[g][i] = *reinterpret_cast<const merge_f4*>(args.prev_records_t_d +
                                                        ((size_t)(ok ? col0 >> 2 : 0) * args.prev_num_records + (ok ? b : 0)) * 4);
        }
      }
    }
  };
  auto sampler_trip = [&](const int t, auto first_trip) {
    constexpr int QUADS = C;  // SMP_STEPS * C / 4
    float zq[4 * QUADS];
    if (DRAW_IN_LOOP)
    {
#pragma unroll
      for (int q = 0; q < QUADS; q++)
        sampling->drawQuad(global_idx, t * C / 4 + q, &zq[4 * q]);
    }
    PIPE_T(if (decltype(first_trip)::value && smp_id == 0) tm.trip(block_idx, 4, 1, lane, __builtin_amdgcn_s_memtime() - tm.t0);)  // stage stamps of the first trip: ticks since kernel entry
    float mu[TRIP_COLS];
    if constexpr (STREAM_MERGE)
    {
      // first trip: the tails' loads were issued at kernel entry and are first touched here, BEHIND the draw — the Philox
      // rounds run under the memory round trip instead of after it
      if constexpr (decltype(first_trip)::value)
      {
        // The draw stays HERE, between the loads' issue and their first use: left alone the compiler sinks the whole draw below
        // the merge (its results are only needed by the shaping at the end) and consumes the loads at once — the wave then sits
        // out the memory round trip and draws afterwards instead of drawing under it.  The empty asm uses the normals, so they
        // exist at this point; the scheduling barrier keeps the merge's arithmetic below it.
#pragma unroll
        for (int q = 0; q < 4 * QUADS; q++)
          asm volatile("" : "+v"(zq[q]));
        __builtin_amdgcn_sched_barrier(0);
        float rho_b[MERGE_LANE_RECORDS], eta_b[MERGE_LANE_RECORDS], eta2_b[MERGE_LANE_RECORDS];
#pragma unroll
        for (int i = 0; i < MERGE_LANE_RECORDS; i++)
        {
          const bool ok = lane + 64 * i < args.prev_num_records;
          rho_b[i] = ok ? tailq[i].x : INFINITY;
          eta_b[i] = ok ? tailq[i].y : 0.0f;
          eta2_b[i] = 0.0f;  // (MergeTails::eta2 is not used here)
        }
        mergeTails<false>(rho_b, eta_b, eta2_b, (float)(1.0 / (double)args.lambda), mt);
      }
      PIPE_T(if (decltype(first_trip)::value && smp_id == 0) tm.trip(block_idx, 4, 2, lane, __builtin_amdgcn_s_memtime() - tm.t0);)
      // this trip's part of u* of the previous iteration: the merge kernel's arithmetic on the quads loaded a trip ago
#pragma unroll
      for (int g = 0; g < TRIP_GROUPS; g++)
      {
        float v[MERGE_LANE_RECORDS][MERGE_COLS], tot[MERGE_COLS];
#pragma unroll
        for (int i = 0; i < MERGE_LANE_RECORDS; i++)
        {
          const bool ok = lane + 64 * i < args.prev_num_records;  // a padding record counts as zeros (combineWave's rule)
          v[i][0] = ok ? vq[g][i].x : 0.0f;
          v[i][1] = ok ? vq[g][i].y : 0.0f;
          v[i][2] = ok ? vq[g][i].z : 0.0f;
          v[i][3] = ok ? vq[g][i].w : 0.0f;
        }
        mergeColumns(mt.s, v, tot);
#pragma unroll
        for (int c = 0; c < MERGE_COLS; c++)
        {
          mu[MERGE_COLS * g + c] = tot[c] / mt.eta_f;
          if (t * C + MERGE_COLS * g + c < TC_all)
            mean_row_s[t * C + MERGE_COLS * g + c] = mu[MERGE_COLS * g + c];  // every lane: same word, same value
        }
      }
      PIPE_T(if (decltype(first_trip)::value && smp_id == 0) tm.trip(block_idx, 4, 3, lane, __builtin_amdgcn_s_memtime() - tm.t0);)
      load_trip_columns(t + SMP_STEPS * NS);  // the quads of this wave's next trip: in flight until then
    }
#pragma unroll
    for (int s2 = 0; s2 < SMP_STEPS; s2++)
    {
      if (t + s2 < num_timesteps)
      {
        if constexpr (STREAM_MERGE)
          sampling->template shapeControlSampleMean<FOLD_Z>(global_idx, t + s2, distribution_idx, &zq[s2 * C], &mu[s2 * C], u);
        else if (DRAW_IN_LOOP)
          sampling->template shapeControlSample<FOLD_Z>(global_idx, t + s2, distribution_idx, &zq[s2 * C], u);
        else
          sampling->template readControlSample<FOLD_Z>(global_idx, t + s2, distribution_idx, u, theta_d_shared, 1, 0, y);
        if (SMP_CONSTRAINS)
          dynamics->enforceConstraints(x, u);
        sampling->writeControlSample(global_idx, t + s2, distribution_idx, u, theta_d_shared, 1, 0, y);
      }
    }
    PIPE_T(if (decltype(first_trip)::value && smp_id == 0) tm.trip(block_idx, 4, 4, lane, __builtin_amdgcn_s_memtime() - tm.t0);)
  };
  constexpr bool EARLY_FIRST_TRIP = DRAW_IN_LOOP && !PAIR_DRAW;
  static_assert(!STREAM_MERGE || EARLY_FIRST_TRIP, "STREAM_MERGE: the record tails are merged inside the early first trip");
  if constexpr (STREAM_MERGE)
  {
    if (role == 0)
    {
      // the record tails (one 8-byte load per record and lane) and the first trip's quads: one memory round trip, under which
      // the first draw runs (the loads' results are first touched by mergeTails inside the first trip)
      const float* tails_t = args.prev_records_t_d + (size_t)TC_all * args.prev_num_records;  // [record][2] behind the quads
#pragma unroll
      for (int i = 0; i < MERGE_LANE_RECORDS; i++)
      {
        const int b = lane + 64 * i;
        tailq[i] = *reinterpret_cast<const merge_f2*>(tails_t + 2 * (b < args.prev_num_records ? b : 0));
      }
      load_trip_columns(SMP_STEPS * smp_id);
      asm volatile("" ::: "memory");  // the loads are issued here, not sunk to their first use behind the draw
      PIPE_T(if (smp_id == 0) tm.trip(block_idx, 4, 0, lane, __builtin_amdgcn_s_memtime() - tm.t0);)
    }
  }
  if (EARLY_FIRST_TRIP && role == 0 && SMP_STEPS * smp_id < num_timesteps)
    sampler_trip(SMP_STEPS * smp_id, std::true_type{});
  // The initial state: a scalar load through a pointer out of the kernel arguments, i.e. a second dependent round trip after
  // the arguments' own.  Up here it stood between kernel entry and the sampler waves' first loads (STREAM_MERGE: the record
  // tails, the critical path of the launch); the waves that need the state wait at the barrier below anyway.
#pragma unroll
  for (int i = 0; i < S; i++)
    x[i] = args.init_x_d[S * thread_idz + i];

  if (lane == 0 && wave_x == 0)
  {
    *smp_prog = 0;
    *dyn_prog = 0;
    *cost_prog = 0;
    *smp_prog1 = 0;
  }
  __syncthreads();

  // every wave runs the (cheap, possibly barrier-containing) initialisers: block-uniform control flow
  dynamics->initializeDynamics(x, u, y, theta_s_shared, 0.0f, dt);
  sampling->initializeDistributions(y, 0.0f, dt, theta_d_shared);
  costs->initializeCosts(y, u, theta_c_shared, 0.0f, dt);
  __syncthreads();

  float running_cost = 0.0f;
  float* row = sampling->sampleRow(theta_d_shared, shared_idx);
  PIPE_T(tm.stop(3);)  // slot 3: kernel entry -> role loops (argument loads, counter reset, initialisers, two barriers)

  if (role == 0 && PAIR_DRAW)
  {
    /* ------------------------------------------------ sampler waves, pair draw ------------------------------------- */
    constexpr int STEPS = 8 / C;  // == 4
    lds_counter_t my_prog = smp_id == 0 ? smp_prog : smp_prog1;
    for (int t = STEPS * smp_id; t < num_timesteps; t += STEPS * NS)
    {
      float zq[4], e[8];
      if (sampling->independentNoise())
      {  // every system draws its own stream: both quads of the trip on this lane, nothing to share (wave-uniform branch)
        sampling->drawQuad(global_idx, t * C / 4, &e[0]);
        sampling->drawQuad(global_idx, t * C / 4 + 1, &e[4]);
      }
      else
      {
        sampling->drawQuad(global_idx, t * C / 4 + thread_idz, zq);
#pragma unroll
        for (int l = 0; l < 4; l++)
        {  // r[0]: the value held by the pair's z = 0 lane, r[1]: by its z = 1 lane
          const unsigned v = __float_as_uint(zq[l]);
          auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
          e[l] = __uint_as_float(r[0]);
          e[4 + l] = __uint_as_float(r[1]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < STEPS; s2++)
      {
        if (t + s2 < num_timesteps)
        {
          sampling->template shapeControlSample<FOLD_Z>(global_idx, t + s2, distribution_idx, &e[s2 * C], u);
          if (SMP_CONSTRAINS)
            dynamics->enforceConstraints(x, u);
          sampling->writeControlSample(global_idx, t + s2, distribution_idx, u, theta_d_shared, 1, 0, y);
        }
      }
      pipePublish(my_prog, min(t + STEPS, num_timesteps), lane);
    }
  }
  else if (role == 0)
  {
    /* ------------------------------------------------ sampler wave ------------------------------------------------ */
    constexpr int STEPS = SMP_STEPS;
    lds_counter_t my_prog = smp_id == 0 ? smp_prog : smp_prog1;
    auto publish = [&](const int value) {
      if constexpr (ROWS_HBM)
        pipePublish(my_prog, value, lane);  // the rows are in global memory: a real release fence
      else
        pipePublishLds(my_prog, value, lane);
    };
    int t = STEPS * smp_id;
    if (EARLY_FIRST_TRIP && t < num_timesteps)
    {  // drawn, shaped and stored in front of the barriers
      publish(min(t + STEPS, num_timesteps));
      t += STEPS * NS;
    }
    for (; t < num_timesteps; t += STEPS * NS)
    {
      sampler_trip(t, std::false_type{});
      publish(min(t + STEPS, num_timesteps));
      PIPE_T(if (smp_id == 0) { const unsigned long long ts0 = tm.t0; tm.stop(0); tm.trip(block_idx, 3, t / (STEPS * NS), lane, tm.t0 - ts0); })
    }
    PIPE_T(tm.stop(0);)  // slot 0: the whole sampler loop (it never waits)
  }
  else if (role == 1)
  {
    /* ------------------------------------------------ dynamics wave ----------------------------------------------- */
    // the four samples of a trip are fetched up front (their LDS latency overlaps the first step's arithmetic)
    auto dyn_step = [&](float* xc, float* xn, int t, const float* u_in) {
#pragma unroll
      for (int i = 0; i < C; i++)
        u[i] = u_in[i];
      if (!SMP_CONSTRAINS)
      {
        dynamics->enforceConstraints(xc, u);
#pragma unroll
        for (int i = 0; i < C; i++)
          row[t * C + i] = u[i];
      }
      dynamics->step(xc, xn, xdot, u, y, theta_s_shared, t, dt);
      float* slot = ring + (size_t)(t % PIPE_RING) * F * 64 + lane;
#pragma unroll
      for (int i = 0; i < O; i++)
        slot[i * 64] = y[i];
      if constexpr (ROWS_HBM)
      {
#pragma unroll
        for (int i = 0; i < C; i++)
          slot[(O + i) * 64] = u[i];
      }
    };
    int seen_smp = 0, seen_smp1 = 0, seen_cost = 0;
    int t = 0;
    // The three counters this wave consumes, read with ONE ds_read_b128 inside the previous trip's arithmetic (`ahead`): the
    // poll's LDS round trip is off the critical path, and only when even that (slightly stale, hence conservative — counters
    // only grow) value does not cover a request does the wave really poll.
    pipe_int4 ahead = { 0, 0, 0, 0 };
    auto look_ahead = [&]() { ahead = *(lds_counter4_t)counters; };
    auto need = [&](lds_counter_t ctr, int& cached, const int ahead_value, const int n) {
      if (cached >= n)
        return;
      cached = max(cached, __builtin_amdgcn_readfirstlane(ahead_value));
      if (cached >= n)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      else
        pipeWait(ctr, n, cached);
    };
    auto wait_trip = [&](const int tp, const int n) {  // the samples of the trip that starts at step tp are in the rows
      if (NS == 2 && (tp & 4))
        need(smp_prog1, seen_smp1, ahead.y, n);
      else
        need(smp_prog, seen_smp, ahead.x, n);
    };
    auto wait_ring = [&](const int n) { need(cost_prog, seen_cost, ahead.z, n); };
    float unext[4 * C];  // ROWS_HBM: the next trip's samples, in flight while the current trip runs
    auto prefetch = [&](const int tp) {
      if (tp < num_timesteps)
      {
        wait_trip(tp, min(tp + 4, num_timesteps));
#pragma unroll
        for (int j = 0; j < 4 * C; j++)
          unext[j] = (tp * C + j < num_timesteps * C) ? row[tp * C + j] : 0.0f;
      }
    };
    if constexpr (ROWS_HBM)
      prefetch(0);
    // Full groups of steps form ONE basic block (no tail test between the steps): the recurrence is skewed — the angle
    // of step t + 1 depends only on the state at t, not on step t's derivative — so the scheduler can start the next
    // step's argument reduction / sincos underneath the current step's division chain.
    //
    // Blocks with two sampler waves and the rows in LDS take EIGHT steps per trip (one trip of each sampler wave): the
    // in-kernel timers of round 4 put ~60 ns of a 4-step trip's ~750 ns into what surrounds the steps — counter checks, the
    // release fence's exposed LDS round trip before every publish, the loop's scalar bookkeeping — and that part halves.
    if constexpr (!ROWS_HBM && NS == 2)
    {
      for (; t + 7 < num_timesteps; t += 8)
      {
        need(smp_prog, seen_smp, ahead.x, t + 4);        // sampler wave 0: steps t .. t+3
        need(smp_prog1, seen_smp1, ahead.y, t + 8);      // sampler wave 1: steps t+4 .. t+7
        PIPE_T(const unsigned long long tw0 = tm.t0; tm.stop(1); tm.trip(block_idx, 0, t / 8, lane, tm.t0 - tw0);)  // slot 1: waiting for the samplers
        wait_ring(t + 8 - PIPE_RING);                    // their ring slots have been consumed
        PIPE_T(const unsigned long long tw1 = tm.t0; tm.stop(2); tm.trip(block_idx, 1, t / 8, lane, tm.t0 - tw1);)  // slot 2: waiting for the cost wave
        float ubuf[8 * C];
#pragma unroll
        for (int j = 0; j < 8 * C; j++)
          ubuf[j] = row[t * C + j];
        dyn_step(x, x_next, t, &ubuf[0]);
        dyn_step(x_next, x, t + 1, &ubuf[C]);
        dyn_step(x, x_next, t + 2, &ubuf[2 * C]);
        dyn_step(x_next, x, t + 3, &ubuf[3 * C]);
        dyn_step(x, x_next, t + 4, &ubuf[4 * C]);
        dyn_step(x_next, x, t + 5, &ubuf[5 * C]);
        dyn_step(x, x_next, t + 6, &ubuf[6 * C]);
        look_ahead();                                    // lands during the last step; consumed by the next trip's checks
        dyn_step(x_next, x, t + 7, &ubuf[7 * C]);
        pipePublishLds(dyn_prog, t + 8, lane);
        PIPE_T(const unsigned long long tw2 = tm.t0; tm.stop(0); tm.trip(block_idx, 2, t / 8, lane, tm.t0 - tw2);)  // slot 0: eight steps of work
      }
    }
    for (; t + 3 < num_timesteps; t += 4)
    {
      if constexpr (!ROWS_HBM)
        wait_trip(t, t + 4);                             // samples for steps t .. t+3 are in the rows
      PIPE_T(tm.stop(1);)                                // slot 1: waiting for the sampler
      wait_ring(t + 4 - PIPE_RING);                      // their ring slots have been consumed
      PIPE_T(tm.stop(2);)                                // slot 2: waiting for the cost wave (ring back-pressure)
      float ubuf[4 * C];
      if constexpr (ROWS_HBM)
      {
#pragma unroll
        for (int j = 0; j < 4 * C; j++)
          ubuf[j] = unext[j];
        prefetch(t + 4);
      }
      else
      {
#pragma unroll
        for (int j = 0; j < 4 * C; j++)
          ubuf[j] = row[t * C + j];
      }
      dyn_step(x, x_next, t, &ubuf[0]);
      dyn_step(x_next, x, t + 1, &ubuf[C]);
      dyn_step(x, x_next, t + 2, &ubuf[2 * C]);
      look_ahead();
      dyn_step(x_next, x, t + 3, &ubuf[3 * C]);
      pipePublishLds(dyn_prog, t + 4, lane);             // ring (and LDS rows) only: no fence needed, see pipePublishLds
      PIPE_T(tm.stop(0);)                                // slot 0: four steps of work
    }
    if (t < num_timesteps)
    {  // tail of 1..3 steps
      const int hi = num_timesteps;
      if constexpr (!ROWS_HBM)
        wait_trip(t, hi);
      wait_ring(hi - PIPE_RING);
      float ubuf[4 * C];
#pragma unroll
      for (int j = 0; j < 4 * C; j++)
        ubuf[j] = ROWS_HBM ? unext[j] : ((t * C + j < num_timesteps * C) ? row[t * C + j] : 0.0f);
      dyn_step(x, x_next, t, &ubuf[0]);
      if (t + 1 < num_timesteps)
        dyn_step(x_next, x, t + 1, &ubuf[C]);
      if (t + 2 < num_timesteps)
        dyn_step(x, x_next, t + 2, &ubuf[2 * C]);
      pipePublishLds(dyn_prog, hi, lane);
      PIPE_T(tm.stop(0);)
    }
  }
  else if (role == 2)
  {
    /* ------------------------------------------------ cost wave --------------------------------------------------- */
    int seen_dyn = 0;
    auto cost_step = [&](const int tt) {
      const float* slot = ring + (size_t)(tt % PIPE_RING) * F * 64 + lane;
#pragma unroll
      for (int i = 0; i < O; i++)
        y[i] = slot[i * 64];
#pragma unroll
      for (int i = 0; i < C; i++)
        u[i] = ROWS_HBM ? slot[(O + i) * 64] : row[tt * C + i];
      running_cost += costs->computeRunningCost(y, u, tt, theta_c_shared, &crash_status) +
                      sampling->template computeLikelihoodRatioCost<FOLD_Z, STREAM_MERGE>(u, theta_d_shared, global_idx, tt,
                                                                                          distribution_idx, args.lambda,
                                                                                          args.alpha, mean_row_s);
    };
    int t = 0;
    // full groups as one basic block: the four steps' ring / row reads and mean loads issue ahead of the arithmetic
    for (; t + 3 < num_timesteps; t += 4)
    {
      pipeWait(dyn_prog, t + 4, seen_dyn);
      PIPE_T(tm.stop(1);)  // slot 1: waiting for the dynamics wave
      cost_step(t);
      cost_step(t + 1);
      cost_step(t + 2);
      cost_step(t + 3);
      pipePublishLds(cost_prog, t + 4, lane);  // the ring reads above are DS instructions: ordered before this store
      PIPE_T(tm.stop(0);)  // slot 0: four steps of cost
    }
    if (t < num_timesteps)
    {
      pipeWait(dyn_prog, num_timesteps, seen_dyn);
      PIPE_T(tm.stop(1);)
      for (int tt = t; tt < num_timesteps; tt++)
        cost_step(tt);
      pipePublishLds(cost_prog, num_timesteps, lane);
      PIPE_T(tm.stop(0);)
    }
  }
  __syncthreads();
  PIPE_T(tm.stop(4);)  // slot 4: role loop done -> every wave of the block done

  // the cost wave holds the running cost and the last output: it publishes the rollout's result
  const bool writer = (role == 2);
  const float terminal = costs->terminalCost(y, theta_c_shared);
  blockSoftminEpilogue<SAMPLING_T, C, BX, BZ, NTHREADS>(sampling, args, terminal, running_cost, writer, valid, global_idx,
                                                        shared_idx, thread_idz, tid_flat, block_idx, nrows,
                                                        theta_d_shared, cost_s, w_s);
  PIPE_T(tm.stop(5); tm.acc[6] = t_entry; tm.acc[7] = wall_clock64(); tm.flush(block_idx, wave_x + 4 * ring_z, lane);)
  // slot 5: block softmin epilogue; slots 6 / 7: s_memrealtime (100 MHz, chip-wide) at entry / exit: launch ramp and tail across blocks
}


/* =====================================================================================================================
 * The same pipeline for models whose Dynamics runs REPLICATED_LANES > 1 lanes per rollout (the MFMA network forwards).
 *
 * Why it pays even more there: a lone wave issues one VALU instruction every ~2 ns whatever its dependences
 * (tools/ubench/op_latency.hip), so a step costs its instruction count on the slowest wave.  In the fused kernel the
 * 4 lanes of a rollout ALL execute the sampler and the cost function redundantly (~600 of the ~1150 instructions of an
 * AutoRally step) and every SIMD holds exactly one wave at K = 16384.  Here a block of 64 rollouts runs as
 *     REP dynamics waves   16 rollouts x REP lanes each: enforceConstraints, network + kinematics, Euler step
 *     1 sampler wave       64 rollouts x 1 lane: Philox draw / colored row, setGaussianControls rule
 *     1 cost wave          64 rollouts x 1 lane: running cost + likelihood ratio, terminal cost, epilogue
 * so the sampler and the cost are evaluated once per rollout instead of REP times, and they fill issue slots next to the
 * dynamics waves instead of lengthening them.  Protocol, counters and arithmetic are those of rolloutPipelineKernel
 * (bit-identical costs); one progress counter per dynamics wave.  RING (runtime, power of two) is sized by the host to
 * what the sample rows leave of the 160 KiB.
 * Launch: grid = ceil(K / 64), block = ((REP + 2) * 64, 1, 1); wave w < REP: dynamics, w == REP: sampler, w == REP + 1: cost.
 * ===================================================================================================================== */
/** the sampler's block-shared (Grd) bytes sit at the END of its LDS region ([slot rows][Grd], see colored_noise.hpp) and
 *  are only used by initializeDistributions(): the output ring, only used by the step loop, overlays them */
template <class SAMPLING_T>
__host__ __device__ inline int samplerGrdBytes(const SAMPLING_T* smp)
{
  return ((smp->getGrdSharedSizeBytes() + 15) / 16) * 16;
}

/** helper waves of rolloutPipelineRepKernel: a block of REP = 4 dynamics waves puts one on each SIMD of the CU and every SIMD
 *  gets one helper wave next to it.  The SIMD is the unit that saturates (in-kernel timers, tools/pipe_timing.py, round 3:
 *  fp32 MFMA and fp32 VALU share the SIMD's multipliers — their cycles ADD whichever wave issues them), so no helper may
 *  need the whole launch.  Measured for AutoRally-NN (K = 16384, T = 150; s_memtime ticks per launch, the dynamics waves
 *  work 389 k of them): ONE sampler wave is busy for 417 k ticks — it sets the pace — and a cost wave of THREE for 217 k;
 *  TWO samplers 205 k each, a cost wave of TWO 343 k.  Two and two it is (185.4 us per launch against 187.9 with one and
 *  three); the cost waves evaluate ahead of their relay, so they work concurrently either way.  A/B builds:
 *  -DMPPI_PIPE_REP_NS=.. -DMPPI_PIPE_REP_NC=.. */
#if !defined(MPPI_PIPE_REP_NS)
#define MPPI_PIPE_REP_NS 2
#define MPPI_PIPE_REP_NC 2
#endif
constexpr int PIPE_REP_SAMPLERS = MPPI_PIPE_REP_NS;
constexpr int PIPE_REP_COSTS = MPPI_PIPE_REP_NC;

template <class DYN_T, class COST_T, class SAMPLING_T>
__host__ inline size_t pipelineRepSharedBytes(const DYN_T& dyn, const COST_T& cost, const SAMPLING_T& smp, int ring)
{
  const int slots = 64;
  size_t n = 0;
  n += calcClassSharedMemSize(&dyn, slots);
  n += calcClassSharedMemSize(&cost, slots);
  n += calcClassSharedMemSize(&smp, slots) - samplerGrdBytes(&smp);       // the sample rows (none when they live in HBM)
  // output ring [slot][i][rollout]; with the rows in HBM the clamped control travels through it as well
  const size_t ring_bytes = sizeof(float) * (size_t)ring * (DYN_T::OUTPUT_DIM + (smp.rows_global_d_ ? DYN_T::CONTROL_DIM : 0)) * 64;
  n += ring_bytes > (size_t)samplerGrdBytes(&smp) ? ring_bytes : (size_t)samplerGrdBytes(&smp);
  n += sizeof(float) * 4 * math::nearest_multiple_4(slots);              // cost_s, w_s, relay_cost, relay_status
  n += sizeof(int) * 4 * (replicated_lanes<DYN_T>::value + PIPE_REP_SAMPLERS + 1);  // progress counters (padded)
  return n;
}

/** largest power-of-two ring (<= 32 steps) that fits next to the sample rows; 0 if not even 4 steps fit */
template <class DYN_T, class COST_T, class SAMPLING_T>
__host__ inline int pipelineRepRingSteps(const DYN_T& dyn, const COST_T& cost, const SAMPLING_T& smp, size_t max_lds)
{
  for (int ring = 32; ring >= 4; ring >>= 1)
    if (pipelineRepSharedBytes(dyn, cost, smp, ring) <= max_lds)
      return ring;
  return 0;
}

/** pins every 32-bit word of a (trivially copyable) object to a VGPR: see the cost wave of rolloutPipelineRepKernel */
template <class T>
__device__ inline void vgprResident(T& obj)
{
  static_assert(sizeof(T) % 4 == 0, "whole words");
  uint32_t* w = reinterpret_cast<uint32_t*>(&obj);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); i++)
    asm volatile("" : "+v"(w[i]));
}

/**
 * ROWS_HBM (long horizons: T * C floats per rollout do not fit the LDS next to the ring): the block's sample rows live in the
 * sampler's HBM buffer (blockRows).  The sampler waves write them there (write-through stores, nobody waits for them), the
 * dynamics waves fetch the samples of the NEXT pair of steps while they compute the current one (the rows are complete far
 * ahead of them), and the clamped control travels to the cost waves through the output ring next to the outputs — so no wave
 * waits on a global load inside its step loop.  Same arithmetic, same bits.
 */
template <class DYN_T, class COST_T, class SAMPLING_T, bool DRAW_IN_LOOP, bool ROWS_HBM = false>
/* A block of this kernel is 512 threads = two waves per SIMD, and its LDS request (sample rows + ring) leaves room for one
 * block per CU: the second __launch_bounds__ argument tells the register allocator that two waves per SIMD is all there will
 * ever be, i.e. that 256 VGPRs per wave are there to be used (without it it aims at four and spills at 128).
 * A/B: -DMPPI_PIPE_REP_MIN_WAVES=0 restores the default. */
#if !defined(MPPI_PIPE_REP_MIN_WAVES)
#define MPPI_PIPE_REP_MIN_WAVES 2
#endif
#if MPPI_PIPE_REP_MIN_WAVES > 0
#define MPPI_PIPE_REP_BOUNDS(n) __launch_bounds__(n, MPPI_PIPE_REP_MIN_WAVES)
#else
#define MPPI_PIPE_REP_BOUNDS(n) __launch_bounds__(n)
#endif
__global__ void MPPI_PIPE_REP_BOUNDS(64 * (replicated_lanes<DYN_T>::value + PIPE_REP_SAMPLERS + PIPE_REP_COSTS))
    rolloutPipelineRepKernel(DYN_T dynamics_obj, COST_T costs_obj, SAMPLING_T sampling_obj, const RolloutArgs args,
                             const int ring_steps)
{
  constexpr int BX = 64;
  constexpr int REP = replicated_lanes<DYN_T>::value;
  static_assert(REP > 1 && 64 % REP == 0, "this variant is for replicated-lane dynamics");
  constexpr int DW = BX * REP / 64;  // dynamics waves
  constexpr int NS = PIPE_REP_SAMPLERS, NC = PIPE_REP_COSTS;
  // wave order: dynamics 0 .. DW-1, then the sampler(s), then the cost waves: with the waves of a workgroup dealt to the
  // CU's four SIMDs in turn, every SIMD gets one dynamics wave and one helper
  constexpr int NWAVES = DW + NS + NC;
  constexpr int NTHREADS = 64 * NWAVES;
  constexpr int PER_WAVE = 64 / REP;
  __builtin_assume(__builtin_amdgcn_workgroup_size_x() == NTHREADS);
  __builtin_assume(__builtin_amdgcn_workgroup_size_y() == 1);
  __builtin_assume(__builtin_amdgcn_workgroup_size_z() == 1);
  __builtin_assume(__builtin_amdgcn_workitem_id_x() < NTHREADS);
  __builtin_assume(__builtin_amdgcn_workitem_id_y() == 0);
  __builtin_assume(__builtin_amdgcn_workitem_id_z() == 0);

  DYN_T* dynamics = &dynamics_obj;
  COST_T* costs = &costs_obj;
  SAMPLING_T* sampling = &sampling_obj;
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  constexpr int SLOTS = BX;
  constexpr bool SMP_CONSTRAINS = !DYN_T::CONSTRAINTS_DEPEND_ON_STATE;  // see rolloutPipelineKernel

  const int tid_x = (int)__builtin_amdgcn_workitem_id_x();
  const int wave = __builtin_amdgcn_readfirstlane(tid_x >> 6);  // wave-uniform
  const int lane = tid_x & 63;
  const bool is_dyn = wave < DW;
  const bool is_sampler = !is_dyn && wave < DW + NS;
  const int helper_id = is_dyn ? 0 : (is_sampler ? wave - DW : wave - DW - NS);  // which of the samplers / cost waves
  // rollout slot of this thread: dynamics waves carry PER_WAVE rollouts x REP lanes, the other two one lane per rollout
  const int thread_idx = is_dyn ? wave * PER_WAVE + (lane % PER_WAVE) : lane;
  const int rep_lane = is_dyn ? lane / PER_WAVE : 0;
  const int block_idx = (int)blockIdx.x;
  const int global_idx = BX * block_idx + thread_idx;
  const int shared_idx = thread_idx;
  const int num_timesteps = args.num_timesteps;
  const int num_rollouts = args.num_rollouts;
  const float dt = args.dt;
  const bool valid = global_idx < num_rollouts;
  const int nrows = min(BX, num_rollouts - BX * block_idx);
  const int ring_mask = ring_steps - 1;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s_shared = reinterpret_cast<float*>(smem_raw);
  float* theta_c_shared = theta_s_shared + calcClassSharedMemSize(dynamics, SLOTS) / (int)sizeof(float);
  float* theta_d_lds = theta_c_shared + calcClassSharedMemSize(costs, SLOTS) / (int)sizeof(float);
  float* theta_d_shared = theta_d_lds;
  if constexpr (ROWS_HBM)
  {
    theta_d_shared = sampling->blockRows(theta_d_lds, block_idx, SLOTS);
    sampling->setStagingBase(theta_d_lds);
  }
  constexpr int F = O + (ROWS_HBM ? C : 0);  // floats per step and rollout in the output ring
  // [sample rows][sampler Grd (prologue only)  U  output ring (step loop only)][cost_s][w_s][counters]
  float* ring = theta_d_lds + (calcClassSharedMemSize(sampling, SLOTS) - samplerGrdBytes(sampling)) / (int)sizeof(float);
  const int overlay_floats = max(ring_steps * F * 64, samplerGrdBytes(sampling) / (int)sizeof(float));
  float* cost_s = ring + overlay_floats;
  float* w_s = cost_s + math::nearest_multiple_4(SLOTS);
  float* relay_cost = w_s + math::nearest_multiple_4(SLOTS);  // running cost / status handed from cost wave to cost wave
  int* relay_status = reinterpret_cast<int*>(relay_cost + math::nearest_multiple_4(SLOTS));
  lds_counter_t counters = (lds_counter_t)(relay_status + math::nearest_multiple_4(SLOTS));
  // counters + s, s < NS (<= 2): sampler s — steps (of ITS trips) whose shaped sample is in the row
  // counters + 2: steps the cost waves have consumed          } one 16-byte group: a dynamics wave reads all it waits for
  // counters + 4 + w, w < DW: steps whose output dynamics wave w has put into the ring   (the cost waves' group)
  static_assert(NS <= 2, "the sampler counters share a 16-byte group with the cost counter");
  lds_counter_t cost_prog = counters + 2;

  sampling->setThreadMapping(shared_idx, BX);

  float x[S], x_next[S], xdot[S], u[C], y[O];
  int crash_status = 0;
#pragma unroll
  for (int i = 0; i < S; i++)
  {
    x[i] = args.init_x_d[i];
    xdot[i] = 0.0f;
    x_next[i] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < C; i++)
    u[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < O; i++)
    y[i] = 0.0f;
  if (tid_x < 4 + DW)
    counters[tid_x] = 0;
  __syncthreads();

  dynamics->initializeDynamics(x, u, y, theta_s_shared, 0.0f, dt);
  sampling->initializeDistributions(y, 0.0f, dt, theta_d_shared);
  costs->initializeCosts(y, u, theta_c_shared, 0.0f, dt);
  __syncthreads();

  float running_cost = 0.0f;
  float* row = sampling->sampleRow(theta_d_shared, shared_idx);
  constexpr int STEPS = (C % 2 == 0) ? 2 : 4;  // steps per sampler trip (whole Philox quads)

  if (is_sampler)
  {
    /* ------------------------------------------------ sampler waves ----------------------------------------------- */
    // sampler s takes trips s, s + NS, ...
    constexpr int QUADS = STEPS * C / 4;
    lds_counter_t smp_prog = counters + helper_id;
    PIPE_T(PipeTimer tm; tm.start();)
    for (int t = STEPS * helper_id; t < num_timesteps; t += STEPS * NS)
    {
      float zq[4 * QUADS];
      if (DRAW_IN_LOOP)
      {
#pragma unroll
        for (int q = 0; q < QUADS; q++)
          sampling->drawQuad(global_idx, t * C / 4 + q, &zq[4 * q]);
      }
#pragma unroll
      for (int s2 = 0; s2 < STEPS; s2++)
      {
        if (t + s2 < num_timesteps)
        {
          if (DRAW_IN_LOOP)
            sampling->shapeControlSample(global_idx, t + s2, 0, &zq[s2 * C], u);
          else
            sampling->readControlSample(global_idx, t + s2, 0, u, theta_d_shared, 1, 0, y);
          if (SMP_CONSTRAINS)
            dynamics->enforceConstraints(x, u);
          sampling->writeControlSample(global_idx, t + s2, 0, u, theta_d_shared, 1, 0, y);
        }
      }
      if constexpr (ROWS_HBM)
        pipePublish(smp_prog, min(t + STEPS, num_timesteps), lane);  // rows in global memory: a real release fence
      else
        pipePublishLds(smp_prog, min(t + STEPS, num_timesteps), lane);
    }
    PIPE_T(tm.stop(0); tm.flush(block_idx, wave, lane);)  // slot 0: the whole sampler loop (it never waits)
  }
  else if (is_dyn)
  {
    /* ------------------------------------------------ dynamics waves ---------------------------------------------- */
    lds_counter_t my_prog = counters + 4 + wave;
    auto dyn_step = [&](float* xc, float* xn, int t, const float* u_in) {
#pragma unroll
      for (int i = 0; i < C; i++)
        u[i] = u_in[i];
      // The REP replicas of a rollout hold identical copies of u and y: ALL of them store (same word, same value) instead
      // of `if (rep_lane == 0) { ... }`.  A region that narrows EXEC to one replica inside a spilling four-lane kernel gave
      // wrong covariance outputs in one round-3 build (the register allocator works on the wave-level control
      // flow and may place spill code of a live-through value inside such a region, where it only covers the active
      // lanes); with no replica-divergent region in the step loop there is no place for that to happen — and the
      // s_and_saveexec / s_or pair around the stores is gone as well.
      if (!SMP_CONSTRAINS)
      {
        dynamics->enforceConstraints(xc, u);
#pragma unroll
        for (int i = 0; i < C; i++)
          row[t * C + i] = u[i];
      }
      dynamics->step(xc, xn, xdot, u, y, theta_s_shared, t, dt);
      {
        float* slot = ring + (size_t)(t & ring_mask) * F * 64 + thread_idx;
#pragma unroll
        for (int i = 0; i < O; i++)
          slot[i * 64] = y[i];
        if constexpr (ROWS_HBM)
        {
#pragma unroll
          for (int i = 0; i < C; i++)
            slot[(O + i) * 64] = u[i];
        }
      }
    };
    int seen_smp[NS], seen_cost = 0;
#pragma unroll
    for (int q = 0; q < NS; q++)
      seen_smp[q] = 0;
    // the counters this wave waits for (samplers, cost waves) in ONE 16-byte read issued inside the previous pair's
    // arithmetic: see rolloutPipelineKernel
    pipe_int4 ahead = { 0, 0, 0, 0 };
    auto look_ahead = [&]() { ahead = *(lds_counter4_t)counters; };
    auto need = [&](lds_counter_t ctr, int& cached, const int ahead_value, const int n) {
      if (cached >= n)
        return;
      cached = max(cached, __builtin_amdgcn_readfirstlane(ahead_value));
      if (cached >= n)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      else
        pipeWait(ctr, n, cached);
    };
    // the pair (t, t + 1) lies inside one sampler trip (STEPS is even): wait for the sampler that owns the trip
    auto wait_samples = [&](const int t, const int n) {
      if constexpr (NS == 1)
        need(counters, seen_smp[0], ahead.x, n);
      else
      {
        const int owner = (t / STEPS) % NS;
        if (owner == 0)
          need(counters, seen_smp[0], ahead.x, n);
        else
          need(counters + 1, seen_smp[NS - 1], ahead.y, n);
      }
    };
    int t = 0;
    // full pairs of steps as one basic block (see rolloutPipelineKernel): lets the scheduler overlap the second step's
    // independent work with the first step's MFMA chains
    PIPE_T(PipeTimer tm; tm.start();)
    float unext[2 * C];  // ROWS_HBM: the samples of the pair after the current one, in flight while the current one runs
    auto prefetch = [&](const int tp) {
      if (tp < num_timesteps)
      {
        wait_samples(tp, min(tp + 2, num_timesteps));
#pragma unroll
        for (int j = 0; j < 2 * C; j++)
          unext[j] = (tp * C + j < num_timesteps * C) ? row[tp * C + j] : 0.0f;
      }
    };
    if constexpr (ROWS_HBM)
      prefetch(0);
    for (; t + 1 < num_timesteps; t += 2)
    {
      if constexpr (!ROWS_HBM)
        wait_samples(t, t + 2);
      PIPE_T(tm.stop(1);)  // slot 1: waiting for the sampler
      need(cost_prog, seen_cost, ahead.z, t + 2 - ring_steps);
      PIPE_T(tm.stop(2);)  // slot 2: waiting for the cost waves (ring back-pressure)
      float ubuf[2 * C];
      if constexpr (ROWS_HBM)
      {
#pragma unroll
        for (int j = 0; j < 2 * C; j++)
          ubuf[j] = unext[j];
        prefetch(t + 2);
      }
      else
      {
#pragma unroll
        for (int j = 0; j < 2 * C; j++)
          ubuf[j] = row[t * C + j];
      }
      dyn_step(x, x_next, t, &ubuf[0]);
      look_ahead();  // lands during the second step; consumed by the next pair's checks
      dyn_step(x_next, x, t + 1, &ubuf[C]);
      pipePublishLds(my_prog, t + 2, lane);  // ring (and LDS rows) only: no fence needed, see pipePublishLds
      PIPE_T(tm.stop(0);)  // slot 0: two steps of work
    }
    PIPE_T(tm.flush(block_idx, wave, lane);)
    if (t < num_timesteps)
    {
      if constexpr (!ROWS_HBM)
        wait_samples(t, num_timesteps);
      need(cost_prog, seen_cost, ahead.z, num_timesteps - ring_steps);
      float ubuf[C];
#pragma unroll
      for (int j = 0; j < C; j++)
        ubuf[j] = ROWS_HBM ? unext[j] : row[t * C + j];
      dyn_step(x, x_next, t, &ubuf[0]);
      pipePublishLds(my_prog, num_timesteps, lane);
    }
  }
  else
  {
    /* ------------------------------------------------ cost waves -------------------------------------------------- */
    // The cost plugin's parameters arrive as kernel arguments, i.e. in SGPRs, and this kernel's three roles together want
    // more than the ~100 a wave has: the step loop of this wave was reloading spilled SGPRs with ~45 v_readlane per step.
    // A private copy whose words are pinned to VGPRs (which this wave has to spare) takes the parameters out of that
    // competition; every lane holds the same values, the arithmetic is unchanged.
    //
    // NC cost waves take the PAIRS of steps in turn.  The running cost is a sum in step order and the plugin's status word
    // (crash flags) threads through the steps, so a pair's result can only be FINAL after the pair before it — but almost
    // all of a pair's work (map lookups, trigonometry, the terms of the sum) does not wait for that: the wave evaluates its
    // pair AHEAD of its turn with the status it last saw (its own result NC pairs ago), the waves therefore work
    // concurrently, and what is left for the turn itself — the RELAY: running cost and status of the 64 rollouts handed on
    // in LDS — is two additions.  Only when the status some rollout arrives with differs from the guess (a crash flag
    // raised during the NC - 1 pairs in between) is the pair evaluated again, now with the true status, for the whole wave:
    // same plugin calls with the same inputs as the in-order evaluation, so the costs are the same bits.
    // (Round 2 evaluated inside the relay: the in-kernel timers showed that chain — 2 x 3100 cycles per pair — busy for
    // the whole launch, as long as the dynamics waves themselves.)
    // (Round 6: a cost class that is a pure parameter block — kernarg_viewable — is not copied at all: every evaluation runs on
    // the kernel's argument block behind an opaque pointer, the parameters arrive by s_load where they are used and occupy
    // neither SGPRs across the loop nor VGPRs.)
    constexpr bool COST_VIEW = MPPI_KERNARG_RELOAD && MPPI_COST_KERNARG_VIEW && kernarg_viewable<COST_T>::value;
    constexpr size_t COST_OFFSET = KernargLayout<DYN_T, COST_T>::template offset<1>();
    COST_T costs_v = *costs;
    if constexpr (!COST_VIEW)
      vgprResident(costs_v);
    COST_T* costs_w = &costs_v;
    int seen_dyn[DW], seen_cost = 0;
#pragma unroll
    for (int w = 0; w < DW; w++)
      seen_dyn[w] = 0;
    int status_guess = 0;  // the status this wave expects its next pair to start from
    PIPE_T(PipeTimer tm; tm.start();)
    for (int t = 2 * helper_id; t < num_timesteps; t += 2 * NC)
    {
      const int hi = min(t + 2, num_timesteps);
#pragma unroll
      for (int w = 0; w < DW; w++)
        pipeWait(counters + 4 + w, hi, seen_dyn[w]);
      PIPE_T(tm.stop(1);)  // slot 1: waiting for the dynamics waves
      float yb[2][O], ub[2][C];
#pragma unroll
      for (int q = 0; q < 2; q++)
      {
        const int tt = min(t + q, num_timesteps - 1);
        const float* slot = ring + (size_t)(tt & ring_mask) * F * 64 + lane;
#pragma unroll
        for (int i = 0; i < O; i++)
          yb[q][i] = slot[i * 64];
#pragma unroll
        for (int i = 0; i < C; i++)
          ub[q][i] = ROWS_HBM ? slot[(O + i) * 64] : row[tt * C + i];
      }
      PIPE_T(tm.stop(3);)  // slot 3: fetching outputs / controls of the pair
      float cq[2] = { 0.0f, 0.0f };
      auto evaluate = [&](int status) {
        if constexpr (COST_VIEW)
          costs_w = kernargObject<COST_T>(kernargBase(), COST_OFFSET);
#pragma unroll
        for (int q = 0; q < 2; q++)
        {
          if (t + q < num_timesteps)
            cq[q] = costs_w->computeRunningCost(yb[q], ub[q], t + q, theta_c_shared, &status) +
                    sampling->computeLikelihoodRatioCost(ub[q], theta_d_shared, global_idx, t + q, 0, args.lambda, args.alpha);
        }
        return status;
      };
      int status_out = evaluate(status_guess);
      PIPE_T(tm.stop(0);)  // slot 0: the pair's cost evaluation, ahead of the relay
      if (t > 0)
      {
        pipeWait(cost_prog, t, seen_cost);
        running_cost = relay_cost[lane];
        crash_status = relay_status[lane];
      }
      PIPE_T(tm.stop(2);)  // slot 2: waiting for the relay
      if (__builtin_amdgcn_ballot_w64(crash_status != status_guess) != 0ull)
      {  // wave-uniform: some rollout's status changed in between
        status_out = evaluate(crash_status);
        PIPE_T(tm.stop(4);)  // slot 4: re-evaluation inside the relay
      }
      running_cost += cq[0];
      if (t + 1 < num_timesteps)
        running_cost += cq[1];
      crash_status = status_out;
      status_guess = status_out;
      relay_cost[lane] = running_cost;
      relay_status[lane] = crash_status;
      pipePublishLds(cost_prog, hi, lane);  // relay slots and ring reads are DS instructions: ordered before this store
      PIPE_T(tm.stop(5);)  // slot 5: the relay itself
    }
    PIPE_T(tm.flush(block_idx, wave, lane);)
  }
  __syncthreads();

  // cost wave 0 publishes the rollouts' results: the last pair's wave left the running cost in the relay slots and the
  // last step's outputs are still in the ring (terminalCost reads them)
  const bool writer = !is_dyn && !is_sampler && helper_id == 0;
  if (writer)
  {
    running_cost = relay_cost[lane];
    const float* slot = ring + (size_t)((num_timesteps - 1) & ring_mask) * F * 64 + lane;
#pragma unroll
    for (int i = 0; i < O; i++)
      y[i] = slot[i * 64];
  }
  const float terminal = costs->terminalCost(y, theta_c_shared);
  blockSoftminEpilogue<SAMPLING_T, C, BX, 1, NTHREADS>(sampling, args, terminal, running_cost, writer, valid, global_idx,
                                                       shared_idx, 0, tid_x, block_idx, nrows, theta_d_shared, cost_s, w_s);
}

}  // namespace kernels
}  // namespace mppi

#endif
