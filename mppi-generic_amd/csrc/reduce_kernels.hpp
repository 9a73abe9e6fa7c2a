/**
 * reduce_kernels.hpp — second pass of the two-pass softmin reduction.
 *
 * Replaces the reference's host baseline scan + normExpKernel + host normaliser sum + weightedReductionKernel tail
 * (include/mppi/core/mppi_common.cu:858-900, 686-701, 1055-1063, 1138-1160; three blocking D2H copies per iteration in
 * controllers/MPPI/mppi_controller.cu:187-219) with one small launch that never leaves the device.
 *
 * Input: per-(system, block) records {U_b[T*C], rho_b, eta_b, sum w^2} from rolloutKernel (or per-GPU records after the
 * exchange).  Merge rule (exact in real arithmetic, ~1e-7 relative in fp32 — SURVEY.md §8e):
 *   rho = min_b rho_b;  s_b = exp(-(rho_b - rho)/lambda);  eta = sum_b s_b eta_b (double);
 *   U[j] = sum_b s_b U_b[j];   u*[j] = U[j] / eta.
 * finalize == 0 writes the merged record (per-GPU partial, layout identical to the input records) instead of u*.
 * Also produces the reference's free-energy statistics (mppi_common.cu:1065-1081) from eta and sum w^2.
 *
 * Launch: grid = (D systems, combineGridY(T*C)), block = MERGE_THREADS (4 waves), no LDS.  The kernel is a chain of
 * latencies (the records were just written by other CUs), so it is organised as ONE memory round trip per wave and nothing
 * else that waits: a wave owns MERGE_COLS columns, a lane the records l, l + 64, ... — see combineWave().
 */
#ifndef MPPI_AMD_REDUCE_KERNELS_HPP_
#define MPPI_AMD_REDUCE_KERNELS_HPP_

#include <hip/hip_runtime.h>
#include <math.h>
#include "mppi_amd/det_math.h"
#include "mppi_amd/utils/wave_ops.hpp"
#include "mppi_amd/engine/merge_wave.hpp"

namespace mppi
{
namespace kernels
{
constexpr int COMBINE_THREADS = 1024;  ///< 16 waves: the merge is a chain of memory round trips, so it wants loads in flight, not ALUs
/** floats per system in the stats buffer: rho, eta, fe_mean, fe_var, fe_modified_var, sum w^2, pad, pad */
constexpr int STATS_STRIDE = 8;

/**
 * P2P exchange over xGMI (SURVEY.md §8e second stage): where this GPU's merged record [D][PS] goes — straight into the mailbox
 * slot [rank] of every peer (peer memory mapped through hipIpc / peer access), written by the local merge itself
 * (combineKernel, finalize == 0) with write-through system-scope stores; the block that finishes last raises one flag per
 * peer, and each peer's global merge (combineKernel with wait_flags_d) spins on its own flags.  Replaces an ncclAllGather of
 * ~1 KB (10-20 us of latency on a ~30 us iteration) by one hop inside a launch that exists anyway.
 */
struct PostTargets
{
  float* peer_slot[16];     ///< peer p's mailbox slot for THIS rank's record ([D][PS]); this rank's own mailbox included
  unsigned* peer_flag[16];  ///< peer p's flag word for this rank
  int world;                ///< 0: no posting (the record goes to record_out_d only)
  unsigned seq;
  unsigned* ticket_d;       ///< device counter (zero between launches): the last block to arrive raises the flags
};

struct CombineArgs
{
  const float* records_d;  ///< record b of system z at records_d + z * z_stride + b * rec_stride
  int z_stride;            ///< [D][num_records][PS] (block records): num_records * PS;  [world][D][PS] (gathered): PS
  int rec_stride;          ///<                                       PS;                                          D * PS
  int num_records;
  int TC;                  ///< T * C
  int PS;                  ///< record stride in floats (TC + 4)
  float lambda;
  int num_rollouts_total;  ///< K over everything merged so far (free-energy normalisation)
  int finalize;
  float* mean_out_d;       ///< finalize: [D][T*C] new control mean (= u*)
  float* record_out_d;     ///< !finalize: [D][PS] merged record
  float* stats_out_d;      ///< finalize: [D][STATS_STRIDE]
  /* records delivered by peers into this GPU's mailbox (P2P exchange over xGMI, postRecordsKernel): wait until every
   * peer's flag shows `wait_seq`, then read the records with system-scope loads (they were written by other agents) */
  const unsigned* wait_flags_d;  ///< [num_records] or nullptr
  unsigned wait_seq;
  unsigned long long wait_limit_ticks;  ///< give up after this many wall_clock64 ticks (100 MHz): stats[6] = 1 marks the failure
  PostTargets post;        ///< !finalize: also deliver the merged record to the peers' mailboxes
};

/** a float another agent (peer GPU over xGMI, or a kernel of another process) may have written: system-scope load, which
 *  bypasses the caches that are not coherent with that agent */
__device__ inline float loadPeerWritten(const float* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ inline float blockMin(float v, float* red_s)
{
  // wave64 shuffle tree, then across the waves of the block through LDS
  for (int off = 32; off > 0; off >>= 1)
    v = fminf(v, __shfl_xor(v, off, 64));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0)
    red_s[wave] = v;
  __syncthreads();
  float r = red_s[0];
  for (int i = 1; i < COMBINE_THREADS / 64; i++)
    r = fminf(r, red_s[i]);
  __syncthreads();
  return r;
}

__device__ inline double blockSum(double v, double* red_s)
{
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0)
    red_s[wave] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < COMBINE_THREADS / 64; i++)
    r += red_s[i];
  __syncthreads();
  return r;
}

constexpr int COMBINE_COLS = 64;  ///< columns per block of the Tsallis mean kernel below

#if defined(MPPI_COMBINE_TIMING)
/* A/B instrumentation (tools/combine_timing.py; never defined in a product build): s_memtime at the phases of combineKernel,
 * [wave][8] for the first waves of the grid */
static __device__ unsigned long long g_combine_timing[4 * 8];
#define COMBINE_T(i)                                                                                                    \
  if ((threadIdx.x & 63) == 0 && wave_global < 4)                                                                      \
  g_combine_timing[wave_global * 8 + (i)] = __builtin_amdgcn_s_memtime()
#else
#define COMBINE_T(i)
#endif

using mppi::wave::waveAllMin;  // wave64 all-reduces on DPP + v_readlane (include/mppi_amd/utils/wave_ops.hpp)
using mppi::wave::waveAllSum;

constexpr int MERGE_WAVES = 1;      ///< waves per block: ONE — a wave's loads touch 64 cache lines per instruction (lane = record, rows
                                    ///< 416 B apart), and four waves queueing on one CU's address unit took 3.3 us to get their loads out
constexpr int MERGE_THREADS = 64 * MERGE_WAVES;
/** y extent of combineKernel's grid: ceil(T C / MERGE_COLS) column waves + one wave for the record tail / the statistics */
__host__ __device__ inline int combineGridY(const int TC)
{
  return ((TC + MERGE_COLS - 1) / MERGE_COLS + 1 + MERGE_WAVES - 1) / MERGE_WAVES;
}

/**
 * The merge, one WAVE per MERGE_COLS columns of u* and no data exchanged between waves: lane l owns records l, l + 64, ...
 * (their tails and their MERGE_COLS column values in registers), every wave forms rho, the scale factors, eta and sum w^2 for
 * itself — the same instructions on the same data in every wave, block and rank, hence the same bits — then its columns'
 * sums over the lanes.  One memory round trip (all loads of a wave are issued before anything is waited for), no LDS, no
 * barrier.  Round 4 replaced the block-cooperative form (2 blocks x 16 waves, record tails through LDS, __shfl_xor trees, a
 * barrier pair) whose body took 5.1-5.5 us of a ~29 us Cartpole iteration.
 * One extra wave (the last one of the grid's y extent) writes the statistics (finalize) or the merged record's tail.
 */
template <bool MAILBOX>
__device__ inline float mergeLoad(const float* p)
{
  return MAILBOX ? loadPeerWritten(p) : *p;
}

template <bool MAILBOX>
__device__ inline void combineWave(const CombineArgs& a, const int z, const int wave_global, const int lane)
{
  const int col_waves = (a.TC + MERGE_COLS - 1) / MERGE_COLS;
  const bool stats_wave = wave_global == col_waves;
  if (wave_global > col_waves)
    return;
  const float* rec = a.records_d + (size_t)z * a.z_stride;
  const float lambda_inv = (float)(1.0 / (double)a.lambda);
  const int n = a.num_records;
  const int col0 = wave_global * MERGE_COLS;
  auto scale = [&](const float rho_b, const float rho_) { return mergeScale(rho_b, rho_, lambda_inv); };
  float rho, tot[MERGE_COLS];
  double eta = 0.0, eta2 = 0.0;
  if (n <= 64 * MERGE_LANE_RECORDS)
  {
    // everything this wave needs, requested at once: one round trip to records other CUs (or other GPUs) have just written
    float rho_b[MERGE_LANE_RECORDS], eta_b[MERGE_LANE_RECORDS], eta2_b[MERGE_LANE_RECORDS], v[MERGE_LANE_RECORDS][MERGE_COLS];
    typedef float merge_f4 __attribute__((ext_vector_type(4)));
    // 16-byte loads where the records allow it (T C a multiple of 4: tail and column group are aligned quads): two load
    // instructions per record instead of seven
    const bool quads = !MAILBOX && MERGE_COLS == 4 && (a.TC & 3) == 0 && (a.rec_stride & 3) == 0 && (a.z_stride & 3) == 0;
#pragma unroll
    for (int i = 0; i < MERGE_LANE_RECORDS; i++)
    {
      const int b = lane + 64 * i;
      const bool ok = b < n;
      const float* r = rec + (size_t)(ok ? b : 0) * a.rec_stride;
      if (quads)
      {
        const merge_f4 tail = *reinterpret_cast<const merge_f4*>(r + a.TC);
        const merge_f4 cols = *reinterpret_cast<const merge_f4*>(r + (stats_wave ? 0 : col0));
        rho_b[i] = ok ? tail.x : INFINITY;
        eta_b[i] = ok ? tail.y : 0.0f;
        eta2_b[i] = ok ? tail.z : 0.0f;
        const bool okc = ok && !stats_wave;
        v[i][0] = okc ? cols.x : 0.0f;
        v[i][1] = okc ? cols.y : 0.0f;
        v[i][2] = okc ? cols.z : 0.0f;
        v[i][3] = okc ? cols.w : 0.0f;
        continue;
      }
      rho_b[i] = ok ? mergeLoad<MAILBOX>(r + a.TC) : INFINITY;
      eta_b[i] = ok ? mergeLoad<MAILBOX>(r + a.TC + 1) : 0.0f;
      eta2_b[i] = ok ? mergeLoad<MAILBOX>(r + a.TC + 2) : 0.0f;
#pragma unroll
      for (int c = 0; c < MERGE_COLS; c++)
        v[i][c] = (ok && !stats_wave && col0 + c < a.TC) ? mergeLoad<MAILBOX>(r + col0 + c) : 0.0f;
    }
    COMBINE_T(1);
    MergeTails mt;
    // (the first use of a loaded value waits for the round trip; only the statistics wave needs the sum of w^2)
    if (stats_wave)
      mergeTails<true>(rho_b, eta_b, eta2_b, lambda_inv, mt);
    else
      mergeTails<false>(rho_b, eta_b, eta2_b, lambda_inv, mt);
    COMBINE_T(3);
    rho = mt.rho;
    eta = mt.eta;
    eta2 = mt.eta2;
    COMBINE_T(4);
    mergeColumns(mt.s, v, tot);
    COMBINE_T(5);
  }
  else
  {
    // many records (K / 64 > 256): two passes over the tails, lane-strided in ascending order (the second one hits the L2)
    float m = INFINITY;
    for (int b = lane; b < n; b += 64)
      m = fminf(m, mergeLoad<MAILBOX>(rec + (size_t)b * a.rec_stride + a.TC));
    rho = waveAllMin(m);
    float acc[MERGE_COLS];
#pragma unroll
    for (int c = 0; c < MERGE_COLS; c++)
      acc[c] = 0.0f;
    for (int b = lane; b < n; b += 64)
    {
      const float* r = rec + (size_t)b * a.rec_stride;
      const float s = scale(mergeLoad<MAILBOX>(r + a.TC), rho);
      eta += (double)s * (double)mergeLoad<MAILBOX>(r + a.TC + 1);
      eta2 += (double)s * (double)s * (double)mergeLoad<MAILBOX>(r + a.TC + 2);
#pragma unroll
      for (int c = 0; c < MERGE_COLS; c++)
        acc[c] += s * ((!stats_wave && col0 + c < a.TC) ? mergeLoad<MAILBOX>(r + col0 + c) : 0.0f);
    }
    eta = waveAllSum(eta);
    eta2 = waveAllSum(eta2);
#pragma unroll
    for (int c = 0; c < MERGE_COLS; c++)
      tot[c] = waveAllSum(acc[c]);
  }
  const float eta_f = (float)eta;

  if (!stats_wave)
  {
    if (lane < MERGE_COLS && col0 + lane < a.TC)
    {
      // lane c takes column c (the sums are wave-uniform)
      float mine = tot[0];
#pragma unroll
      for (int c = 1; c < MERGE_COLS; c++)
        mine = lane == c ? tot[c] : mine;
      const int j = col0 + lane;
      if (a.finalize)
        a.mean_out_d[(size_t)z * a.TC + j] = mine / eta_f;
      else
      {
        a.record_out_d[(size_t)z * a.PS + j] = mine;
        for (int p = 0; p < a.post.world; p++)
          __hip_atomic_store(a.post.peer_slot[p] + (size_t)z * a.PS + j, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  else if (lane == 0)
  {
    if (a.finalize)
    {
      // free energy and its variance terms; st[6], the sticky exchange-failure mark, is left alone (merge_wave.hpp)
      mergeStatistics(rho, eta_f, eta2, a.lambda, a.num_rollouts_total, a.stats_out_d + (size_t)z * STATS_STRIDE);
    }
    else
    {
      const float tail[4] = { rho, eta_f, (float)eta2, 0.0f };
      float* o = a.record_out_d + (size_t)z * a.PS + a.TC;
      for (int i = 0; i < 4; i++)
        o[i] = tail[i];
      for (int p = 0; p < a.post.world; p++)
        for (int i = 0; i < 4; i++)
          __hip_atomic_store(a.post.peer_slot[p] + (size_t)z * a.PS + a.TC + i, tail[i], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  COMBINE_T(6);
}

static __global__ void __launch_bounds__(MERGE_THREADS) combineKernel(const CombineArgs a)
{
  const int z = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_global = __builtin_amdgcn_readfirstlane((int)blockIdx.y * MERGE_WAVES + (tid >> 6));
  COMBINE_T(0);
  const bool mailbox = a.wait_flags_d != nullptr;
  if (mailbox)
  {
    // one lane per peer spins on that peer's flag (bounded: a peer that never posts must not wedge the GPU)
    __shared__ int wait_failed_s;
    if (tid == 0)
      wait_failed_s = 0;
    __syncthreads();
    if (tid < a.num_records)
    {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(a.wait_flags_d + tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.wait_seq)
      {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > a.wait_limit_ticks)
        {
          wait_failed_s = 1;
          break;
        }
      }
    }
    __syncthreads();
    if (wait_failed_s)
    {  // leave a mark the host checks (mppi_synchronize / result getters) and no result: the mean is left untouched
      if (tid == 0 && blockIdx.y == 0 && a.stats_out_d)
        a.stats_out_d[(size_t)z * STATS_STRIDE + 6] = 1.0f;
      return;
    }
    combineWave<true>(a, z, wave_global, lane);
  }
  else
  {
    combineWave<false>(a, z, wave_global, lane);
  }
  if (!a.finalize && a.post.world > 0)
  {
    // every block's slice is out (barrier waits for the block's stores, the fence makes them visible system-wide) before it
    // takes a ticket; the block that takes the last one publishes the sequence number to every peer
    __syncthreads();
    if (tid == 0)
    {
      __threadfence_system();
      const unsigned nblocks = gridDim.x * gridDim.y;
      const unsigned t = __hip_atomic_fetch_add(a.post.ticket_d, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (t == nblocks - 1)
      {
        __hip_atomic_store(a.post.ticket_d, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        for (int p = 0; p < a.post.world; p++)
          __hip_atomic_store(a.post.peer_flag[p], a.post.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

/**
 * The merge of a K-SHARDED iteration in ONE launch (P2P mailbox exchange): every wave
 *   1. merges this rank's block records for its columns (combineWave, finalize = 0) and writes the result into slot [rank] of
 *      EVERY peer's mailbox (system-scope stores over xGMI; this rank's own mailbox included),
 *   2. takes a ticket; the wave that takes the last one raises this exchange's flag at every peer,
 *   3. waits (bounded) until every peer's flag shows this exchange, and
 *   4. merges the world's records from its own mailbox into u* (combineWave, finalize = 1).
 * A sharded iteration is then two launches — rollout, merge — plus one hop.  Round 3 ran steps 1-2
 * and 3-4 as two launches (5.6 us apart on a ~32 us iteration).  Step 3 includes this rank's OWN flag, which the launch's last
 * ticket raises: every wave therefore waits until all one-wave blocks of this launch have taken their tickets, i.e. the grid
 * must be CO-RESIDENT (a block that cannot start while the resident ones spin would run them into the 2 s limit).  The grid is
 * D x combineGridY(T*C) blocks of one wave without LDS — 101 at Cartpole's T*C = 100 — and the host refuses the fused form
 * above COMBINE_SHARDED_MAX_BLOCKS (engine_iteration.hip: launchCombineSharded falls back to the two-launch form there).
 */
/** upper limit of the co-resident grid; the host lowers it to half of (CUs x blocks per CU the runtime reports) of the device at
 *  hand (engine_iteration.hip: launchCombineSharded) */
constexpr int COMBINE_SHARDED_MAX_BLOCKS = 2048;
static __global__ void __launch_bounds__(MERGE_THREADS) combineShardedKernel(const CombineArgs loc, const CombineArgs glob)
{
  const int z = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_global = __builtin_amdgcn_readfirstlane((int)blockIdx.y * MERGE_WAVES + (tid >> 6));
  combineWave<false>(loc, z, wave_global, lane);
  if (MERGE_WAVES > 1)
    __syncthreads();
  if (tid == 0)
  {
    __threadfence_system();  // this block's slice of the record is visible to every peer before its ticket counts
    const unsigned nblocks = gridDim.x * gridDim.y;
    const unsigned t = __hip_atomic_fetch_add(loc.post.ticket_d, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == nblocks - 1)
    {
      __hip_atomic_store(loc.post.ticket_d, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int p = 0; p < loc.post.world; p++)
        __hip_atomic_store(loc.post.peer_flag[p], loc.post.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // one lane per peer spins on that peer's flag (bounded: a peer that never posts must not wedge the GPU)
  bool failed = false;
  if (lane < glob.num_records)
  {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(glob.wait_flags_d + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != glob.wait_seq)
    {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > glob.wait_limit_ticks)
      {
        failed = true;
        break;
      }
    }
  }
  if (__builtin_amdgcn_ballot_w64(failed) != 0ull)
  {  // leave a mark the host checks (mppi_synchronize / result getters) and no result: the mean is left untouched
    if (tid == 0 && blockIdx.y == 0 && glob.stats_out_d)
      glob.stats_out_d[(size_t)z * STATS_STRIDE + 6] = 1.0f;
    return;
  }
  combineWave<true>(glob, z, wave_global, lane);
}

/* ---- a second, small mailbox channel for values every rank needs IN FULL on its host (Robust MPPI: the costs of the
 * candidate evaluation, sharded over the ranks by candidates — SURVEY.md §8e "RMPPI init-eval shards the same way"): every rank
 * writes its slice at its position of the array in every peer's aux region, raises its flag there; gatherAuxKernel waits for
 * all flags of its own region and hands the assembled array on. */
constexpr int MAILBOX_AUX_FLOATS = 4096;  ///< per parity; candidates x samples_per_candidate must fit (else: replicated evaluation)
struct AuxTargets
{
  float* peer_aux[16];      ///< peer p's aux array of this exchange's parity (this rank's own mailbox included)
  unsigned* peer_flag[16];  ///< peer p's aux flag word for this rank
  int world;
  unsigned seq;
};
static __global__ void __launch_bounds__(256) postAuxKernel(const float* __restrict__ src, const int offset, const int count, const AuxTargets t)
{
  for (int p = 0; p < t.world; p++)
    for (int i = (int)threadIdx.x; i < count; i += 256)
      __hip_atomic_store(t.peer_aux[p] + offset + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (threadIdx.x == 0)
  {
    __threadfence_system();
    for (int p = 0; p < t.world; p++)
      __hip_atomic_store(t.peer_flag[p], t.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
/** dst[0..n) <- the assembled array; a peer that never posts: dst is filled with NaN (the host reports MPPI_ERR_COMM) */
static __global__ void __launch_bounds__(256)
    gatherAuxKernel(const float* __restrict__ aux, const unsigned* __restrict__ flags, const int world, const unsigned seq,
                    const unsigned long long wait_limit_ticks, const int n, float* __restrict__ dst)
{
  __shared__ int failed_s;
  if (threadIdx.x == 0)
    failed_s = 0;
  __syncthreads();
  if ((int)threadIdx.x < world)
  {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq)
    {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > wait_limit_ticks)
      {
        failed_s = 1;
        break;
      }
    }
  }
  __syncthreads();
  for (int i = (int)threadIdx.x; i < n; i += 256)
    dst[i] = failed_s ? __builtin_nanf("") : loadPeerWritten(aux + i);
}

/** every store of the kernels in front of it on the stream is out: publish `seq` where the host spins (device-mapped host
 *  memory, system scope) — the hand-over of a small kernel whose signature has no flag argument (model step) */
static __global__ void __launch_bounds__(64) raiseFlagKernel(unsigned* flag, unsigned seq)
{
  if (threadIdx.x == 0)
  {
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

/**
 * First kernel of a low-latency mppi_compute_control: the inputs of the call (x0 | nominal control | control history, the
 * handle's input block) are read straight from host memory mapped into the device and written to the device-resident
 * block the rollout kernels read — one launch boundary (~1.5 us) instead of a copy command (~3 us, tools/ubench/handover.hip).
 */
static __global__ void __launch_bounds__(256) ingestKernel(const float* __restrict__ host_mapped, float* __restrict__ dst, int n)
{
  for (int i = (int)threadIdx.x; i < n; i += 256)
    dst[i] = host_mapped[i];
}
/** two ranges of the same block in one launch: [0, n0) and [off1, off1 + n1) — the device-resident copies of x0 and the control
 *  history behind a computeControl whose kernels read the inbox themselves (the nominal control between them holds u* by then) */
static __global__ void __launch_bounds__(256) ingestRangesKernel(const float* __restrict__ src, float* __restrict__ dst, int n0, int off1,
                                                          int n1)
{
  for (int i = (int)threadIdx.x; i < n0; i += 256)
    dst[i] = src[i];
  for (int i = (int)threadIdx.x; i < n1; i += 256)
    dst[off1 + i] = src[off1 + i];
}

/**
 * Tube-MPPI's choice between the two systems after an optimisation pass, on the device
 * (controllers/Tube-MPPI/tube_mppi_controller.cu:264-277): when the actual system's baseline is below the nominal one's plus
 * the threshold, the nominal system restarts from the actual one — its control sequence and its initial state are overwritten
 * with the actual system's.  stats[1][7] carries two bits for the host: bit 0 = this pass's choice (0: the actual state was
 * taken over, 1: the nominal state was kept, the reference's nominalStateUsed), bit 1 = the nominal system's initial state now
 * EQUALS the actual one, bit for bit.  Bit 1 is what the host's copy of the nominal state follows (nominal_state_trajectory_
 * persists across the passes of one call, :268-277): a take-over in pass 0 followed by a pass that keeps the nominal system
 * leaves bit 0 = 1 with x0_d[S..2S) = x0, and the combine of every pass clears the slot, so the fact is re-derived from the
 * states themselves instead of remembered (x0_d[0..S) does not change within a call; when the two states happen to be equal
 * without any take-over, copying one over the other changes nothing).
 * The reference (and rounds 2-4 here) made this choice on the host, between two device
 * passes, which cost mppi_compute_control a second hand-over and a wait for trajectories nobody needed.
 */
static __global__ void __launch_bounds__(256) tubeSelectKernel(float* __restrict__ stats_d, float* __restrict__ mean_d,
                                                        float* __restrict__ x0_d, const int TC, const int S,
                                                        const float nominal_threshold)
{
  const bool take_actual = stats_d[0] < stats_d[STATS_STRIDE] + nominal_threshold;  // block-uniform
  int same = 1;
  if (take_actual)
  {
    for (int i = (int)threadIdx.x; i < TC; i += 256)
      mean_d[TC + i] = mean_d[i];
    for (int i = (int)threadIdx.x; i < S; i += 256)
      x0_d[S + i] = x0_d[i];
  }
  else
  {
    for (int i = (int)threadIdx.x; i < S; i += 256)
      same &= __float_as_uint(x0_d[S + i]) == __float_as_uint(x0_d[i]) ? 1 : 0;
  }
  same = __syncthreads_and(same);
  if (threadIdx.x == 0)
    stats_d[STATS_STRIDE + 7] = (take_actual ? 0.0f : 1.0f) + (same ? 2.0f : 0.0f);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Unfused kernel-level operators with the reference's launch-wrapper semantics, exported through the C ABI for the
 * kernel-level parity tests (reference tests: tests/mppi_core/normexp_kernel_tests.cu, weightedreduction_kernel_tests.cu)
 * and for callers that hold their own cost / sample buffers.
 * ------------------------------------------------------------------------------------------------------------------ */
/** in-place w_i = exp(-lambda_inv * (S_i - baseline)); reference: normExpKernel mppi_common.cu:686-701, :958-966 */
static __global__ void normExpKernel(int num_rollouts, float* trajectory_costs_d, float lambda_inv, float baseline)
{
  const int stride = blockDim.x * gridDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < num_rollouts; i += stride)
  {
    const float cost_dif = trajectory_costs_d[i] - baseline;
    trajectory_costs_d[i] = mppi::det::exp(-lambda_inv * cost_dif);
  }
}

/**
 * Single-block baseline + normaliser of a cost vector: out = {min_k S_k, sum_k exp(-lambda_inv (S_k - min))} — the
 * two-pass device reduction the reference sketches in fullGPUcomputeWeights (mppi_common.cu:1031-1053) but never uses;
 * also applies the exp transform in place.
 */
static __global__ void __launch_bounds__(COMBINE_THREADS)
    computeWeightsKernel(int num_rollouts, float* trajectory_costs_d, float lambda_inv, float* baseline_and_normalizer)
{
  __shared__ double red_d[COMBINE_THREADS / 64];
  __shared__ float red_f[COMBINE_THREADS / 64];
  const int tid = threadIdx.x;
  float m = INFINITY;
  for (int i = tid; i < num_rollouts; i += COMBINE_THREADS)
    m = fminf(m, trajectory_costs_d[i]);
  m = blockMin(m, red_f);
  double s = 0.0;
  for (int i = tid; i < num_rollouts; i += COMBINE_THREADS)
  {
    const float w = mppi::det::exp(-lambda_inv * (trajectory_costs_d[i] - m));
    trajectory_costs_d[i] = w;
    s += (double)w;
  }
  s = blockSum(s, red_d);
  if (tid == 0)
  {
    baseline_and_normalizer[0] = m;
    baseline_and_normalizer[1] = (float)s;
  }
}

/**
 * ColoredMPPI's Tsallis weights (core/mppi_common.cu:968-985 TsallisTransform, launched by
 * controllers/ColoredMPPI/colored_mppi_controller.cu:198-206 when gamma != 0 and r != 0):
 *     w_k = (S_k - rho < gamma) ? exp(log(1 - (S_k - rho) / gamma) / (r - 1)) : 0
 * The weights are not shift-invariant, so the block-local softmin records of the rollout kernel cannot be rescaled into them:
 * this path takes the GLOBAL baseline first (one block over the K costs), writes the weights, the normaliser (double) and the
 * free-energy statistics, and tsallisMeanKernel then forms the weighted mean from the samples the rollout kernel dumped to HBM.
 */
/**
 * K-sharded handles: the baseline must be the GLOBAL one before any weight exists (the weights are not shift-invariant), so
 * the ranks exchange their minima first — `peer_records_d` are the world's merged records of that first exchange ([world][PS]
 * rows of system 0, tail[0] = the rank's own minimum; nullptr: un-sharded) — and `record_tail_out_d` receives {rho, eta_local,
 * sum w^2 local, 0}: the tail of the record of the SECOND exchange, whose merge then needs no rescaling (all rho equal).
 */
static __global__ void __launch_bounds__(COMBINE_THREADS)
    tsallisWeightsKernel(int num_rollouts, const float* __restrict__ costs_d, float gamma, float r, float lambda,
                         float* __restrict__ weights_d, float* __restrict__ stats_out_d,
                         const float* __restrict__ peer_records_d = nullptr, int world = 0, int peer_stride = 0, int TC = 0,
                         const unsigned* __restrict__ wait_flags_d = nullptr, unsigned wait_seq = 0,
                         unsigned long long wait_limit_ticks = 0, float* __restrict__ record_tail_out_d = nullptr)
{
  __shared__ double red_d[COMBINE_THREADS / 64];
  __shared__ float red_f[COMBINE_THREADS / 64];
  __shared__ int wait_failed_s;
  const int tid = threadIdx.x;
  float m = INFINITY;
  if (peer_records_d)
  {
    if (wait_flags_d)
    {  // P2P mailbox: the peers' records of the first exchange must have arrived (bounded wait, as in combineKernel)
      if (tid == 0)
        wait_failed_s = 0;
      __syncthreads();
      if (tid < world)
      {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(wait_flags_d + tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != wait_seq)
        {
          __builtin_amdgcn_s_sleep(2);
          if (wall_clock64() - t0 > wait_limit_ticks)
          {
            wait_failed_s = 1;
            break;
          }
        }
      }
      __syncthreads();
      if (wait_failed_s)
      {
        if (tid == 0)
          stats_out_d[6] = 1.0f;
        return;
      }
    }
    for (int p = tid; p < world; p += COMBINE_THREADS)
      m = fminf(m, loadPeerWritten(peer_records_d + (size_t)p * peer_stride + TC));
  }
  else
  {
    for (int i = tid; i < num_rollouts; i += COMBINE_THREADS)
      m = fminf(m, costs_d[i]);
  }
  m = blockMin(m, red_f);
  double s = 0.0, s2 = 0.0;
  const float inv = 1.0f / (r - 1.0f);
  for (int i = tid; i < num_rollouts; i += COMBINE_THREADS)
  {
    const float cost_dif = costs_d[i] - m;
    float w = 0.0f;
    if (cost_dif < gamma)
      w = mppi::det::exp(mppi::det::log(1.0f - cost_dif / gamma) * inv);
    weights_d[i] = w;
    s += (double)w;
    s2 += (double)w * (double)w;
  }
  s = blockSum(s, red_d);
  s2 = blockSum(s2, red_d);
  if (tid == 0 && record_tail_out_d)
  {
    record_tail_out_d[0] = m;
    record_tail_out_d[1] = (float)s;
    record_tail_out_d[2] = (float)s2;
    record_tail_out_d[3] = 0.0f;
  }
  if (tid == 0)
  {
    // the statistics of combineKernel (mppi_common.cu:1065-1081 computeFreeEnergy) on these weights
    const float K = (float)num_rollouts, eta_f = (float)s, norm = eta_f / K, var = (float)s2;
    const float fe_var = lambda * (var / K - norm * norm);
    const float weird = fe_var / (norm * mppi::det::sqrt(K));
    stats_out_d[0] = m;
    stats_out_d[1] = eta_f;
    stats_out_d[2] = -lambda * mppi::det::log(norm) + m;
    stats_out_d[3] = fe_var;
    stats_out_d[4] = lambda * (weird + 0.5f * (weird * weird));
    stats_out_d[5] = var;
    stats_out_d[7] = 0.0f;  // ([6], the exchange-failure mark, is sticky: see combineWave)
  }
}

/** u*[j] = sum_k w_k v[k][j] / eta over samples in HBM (v: [K][T*C]); block = 64 columns x 16 waves, wave w takes rollouts
 *  w, w + 16, ... in ascending order and the sixteen partials are added in a fixed order (reproducible run to run) */
static __global__ void __launch_bounds__(COMBINE_THREADS)
    tsallisMeanKernel(const float* __restrict__ weights_d, const float* __restrict__ v_d, const float* __restrict__ stats_d,
                      int TC, int num_rollouts, float* __restrict__ mean_out_d, int normalize = 1)
{
  __shared__ float part_s[COMBINE_THREADS / 64][COMBINE_COLS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = COMBINE_THREADS / 64;
  const int j = blockIdx.x * COMBINE_COLS + lane;
  float acc = 0.0f;
  if (j < TC)
    for (int k = wave; k < num_rollouts; k += NW)
      acc += weights_d[k] * v_d[(size_t)k * TC + j];
  part_s[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && j < TC)
  {
    float tot = part_s[0][lane];
#pragma unroll
    for (int w = 1; w < NW; w++)
      tot += part_s[w][lane];
    mean_out_d[j] = normalize ? tot / stats_d[1] : tot;  // !normalize: this rank's part of a K-sharded sum (a record's columns)
  }
}

/**
 * u*[t][c] = sum_k (w_k / eta) v[k][t][c] for samples in HBM (v: [K][T][C], reference layout).
 * reference: weightedReductionKernel mppi_common.cu:710-737.  MI355X mapping: one block per chunk of rollouts reads its
 * rows fully coalesced (a row is contiguous), accumulates T*C columns in registers per thread, and merges chunks with
 * one float atomicAdd per (block, column) into a zeroed output.
 */
static __global__ void __launch_bounds__(256)
    weightedReductionKernel(const float* __restrict__ exp_costs_d, const float* __restrict__ v_d,
                            float* __restrict__ new_u_d, const float normalizer, const int TC, const int num_rollouts,
                            const int rollouts_per_block)
{
  const int k0 = blockIdx.x * rollouts_per_block;
  const int k1 = min(num_rollouts, k0 + rollouts_per_block);
  for (int j = threadIdx.x; j < TC; j += blockDim.x)
  {
    float acc = 0.0f;
    for (int k = k0; k < k1; k++)
    {
      const float weight = exp_costs_d[k] / normalizer;
      acc += weight * v_d[(size_t)k * TC + j];
    }
    atomicAdd(&new_u_d[j], acc);
  }
}

}  // namespace kernels
}  // namespace mppi

#endif
