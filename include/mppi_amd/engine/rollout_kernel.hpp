/**
 * rollout_kernel.hpp — the fused MPPI rollout kernel for gfx950.
 *
 * Replaces the reference's rolloutKernel / rolloutDynamicsKernel + rolloutCostKernel (include/mppi/core/
 * mppi_common.cu:28-146, 269-362, 148-267), setGaussianControls (sampling_distributions/gaussian/gaussian.cu:17-277),
 * normExpKernel (mppi_common.cu:686-701) and the sample-reading half of weightedReductionKernel (mppi_common.cu:710-737)
 * with ONE launch per optimisation iteration:
 *
 *   prologue  sampler.initializeDistributions(): eps rows loaded (coalesced) / drawn into LDS — skipped in the
 *             DRAW_IN_LOOP variant, where each lane draws its Philox quads inside the step loop into registers
 *   loop      per (rollout x, lane y, system z) exactly the reference's call sequence
 *               readControlSample -> enforceConstraints -> writeControlSample -> step -> runningCost + likelihoodRatio
 *             same plugin methods, same argument meaning, same thread-index conventions (x = rollout, y = lane,
 *             z = system).  With BY == 1 the per-rollout state/control/output arrays are thread-private (VGPRs) and no
 *             barrier is executed; with BY > 1 they sit in per-slot LDS and phases are separated by block barriers.
 *   epilogue  cost = running/T + terminal/T (mppi_common.cu:144, :843-853) -> trajectory_costs_d;
 *             block-local softmin: rho_b = min_k S_k, w_k = exp(-(S_k - rho_b)/lambda), eta_b = sum w_k (double),
 *             U_b[t][c] = sum_k w_k v[k][t][c] from the LDS sample rows.  One record (U_b, rho_b, eta_b, sum w^2) per
 *             block and system goes to HBM; combineKernel (reduce_kernels.hpp) merges the records exactly like the
 *             multi-GPU merge: rho = min rho_b, s_b = exp(-(rho_b - rho)/lambda), u* = sum s_b U_b / sum s_b eta_b.
 *
 * HBM traffic per iteration and system: K costs + (K/BX)*(T*C+4) partial floats written, T*C means + S state read.
 * The K*T*C sample tensor never exists in HBM (unless the caller asks for a dump with save_samples).
 *
 * LDS per block: BX*BZ sample rows of (T*C | 1) floats + plugin scratch + (BY > 1: per-slot x, x_next, xdot, u, y).
 */
#ifndef MPPI_AMD_ROLLOUT_KERNEL_HPP_
#define MPPI_AMD_ROLLOUT_KERNEL_HPP_

#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>
#include "mppi_amd/plugin/managed.hpp"
#include "kernarg_view.hpp"
#include "mppi_amd/plugin/math_utils.hpp"
#include "mppi_amd/plugin/parallel_utils.hpp"

#include "mppi_amd/utils/wave_ops.hpp"

namespace mppi
{
namespace kernels
{
/** floats per (system, block) record: U_b[T*C], rho_b, eta_b, sum w^2, pad */
__host__ __device__ inline int partialStride(int num_timesteps, int control_dim)
{
  return num_timesteps * control_dim + 4;
}

/** SINGLE_STEP_SITE of a Dynamics plugin (default false): the fused kernel calls step() from ONE place in its time loop.
 *  For models whose step is thousands of instructions (the RACER suspension / uncertainty models) the usual unrolled pair
 *  of steps + tail would either triple the code (instruction cache) or, as the compiler then decides, turn the step into
 *  an out-of-line function whose state travels through scratch memory. */
template <class T, class = void>
struct single_step_site
{
  static constexpr bool value = false;
};
template <class T>
struct single_step_site<T, std::void_t<decltype(T::SINGLE_STEP_SITE)>>
{
  static constexpr bool value = T::SINGLE_STEP_SITE;
};

/** REPLICATED_LANES of a Dynamics plugin (default 1): the number of wave lanes that carry private copies of one rollout
 *  and cooperate only inside the plugin (e.g. as MFMA k-groups, utils/nn_helpers/fnn_mfma.hpp) */
template <class T, class = void>
struct replicated_lanes
{
  static constexpr int value = 1;
};
template <class T>
struct replicated_lanes<T, std::void_t<decltype(T::REPLICATED_LANES)>>
{
  static constexpr int value = T::REPLICATED_LANES;
};

struct RolloutArgs
{
  float dt;
  int num_timesteps;
  int num_rollouts;  ///< rollouts on this GPU
  float lambda;
  float alpha;
  const float* init_x_d;        ///< [D][S]
  float* trajectory_costs_d;    ///< [D][K_local]
  float* partials_d;            ///< [D][num_blocks][partialStride]
  int save_samples;             ///< != 0: also write the clamped samples to sampler.control_samples_d_
  /* STREAM_MERGE (rolloutPipelineKernel, round 4): the block records of the PREVIOUS iteration (one system, at most 256 of
   * them, T*C a multiple of 4) — this launch's sampler waves merge them into the control mean they shape with, four columns per
   * trip, instead of a merge launch in between; nullptr: the mean is in sampler.control_means_d_ */
  const float* prev_records_d;  ///< [prev_num_records][partialStride]
  int prev_num_records;
  /* The streamed merge reads the records lane = record: in [record][partialStride] every lane of a load instruction touches a
   * cache line of its own (64 line requests for 1 KB of data; 2 waves x 8 instructions per block at kernel entry, in front of the
   * first sample of the launch).  The epilogue therefore writes a second, TRANSPOSED copy of a one-system launch's records —
   * [T*C/4][num_blocks][4] column quads, then [num_blocks][2] tails {rho_b, eta_b}, then [num_blocks] sums of w^2 — which the
   * sampler waves read with 8 / 4 lines per instruction (tools/ubench/record_layout.hip: kernel entry -> data back 1.05 ->
   * 0.58 us), and which lets ONE block merge all records (finalize_kernel.hpp: mergeControlKernel — a hand-over's merge inside
   * its control phase).  Same values, same order of the sums.  nullptr: no copy written / (reader) not a streamed-merge launch. */
  float* records_t_d;
  const float* prev_records_t_d;
};

/** floats of the transposed copy of `num_blocks` one-system records (RolloutArgs::records_t_d) */
__host__ __device__ inline size_t transposedRecordFloats(int num_timesteps, int control_dim, int num_blocks)
{
  return (size_t)num_blocks * (num_timesteps * control_dim + 4);
}

template <class DYN_T, class COST_T, class SAMPLING_T>
__host__ inline size_t rolloutSharedBytes(const DYN_T& dyn, const COST_T& cost, const SAMPLING_T& smp, int bx, int by,
                                          int bz)
{
  const int slots = bx * bz;
  size_t n = 0;
  n += calcClassSharedMemSize(&dyn, slots);
  n += calcClassSharedMemSize(&cost, slots);
  n += calcClassSharedMemSize(&smp, slots);
  if (by > 1)
  {
    n += sizeof(float) * (3 * math::nearest_multiple_4(slots * DYN_T::STATE_DIM) +
                          math::nearest_multiple_4(slots * DYN_T::OUTPUT_DIM) +
                          math::nearest_multiple_4(slots * DYN_T::CONTROL_DIM));
    n += sizeof(float) * math::nearest_multiple_4(bx * by * bz);  // running cost per thread
    n += sizeof(int) * math::nearest_multiple_4(slots);           // crash status
  }
  n += sizeof(float) * 2 * math::nearest_multiple_4(slots);  // cost_s, w_s
  return n;
}

/**
 * Epilogue shared by the rollout kernels: given the trajectory cost `total` of the thread's rollout,
 * block-local softmin record {U_b, rho_b, eta_b, sum w^2} from the LDS sample rows, optional sample dump.
 * Block-uniform; contains barriers.  `writer` marks the one thread that publishes a rollout's result.
 */
template <class SAMPLING_T, int C, int BX, int BZ, int NTHREADS>
__device__ inline void blockSoftminEpilogueCost(SAMPLING_T* sampling, const RolloutArgs& args, const float total,
                                                const bool writer, const bool valid, const int global_idx,
                                                const int shared_idx, const int thread_idz, const int tid_flat,
                                                const int block_idx, const int nrows, float* theta_d_shared,
                                                float* cost_s, float* w_s)
{
  const int num_timesteps = args.num_timesteps;
  const int num_rollouts = args.num_rollouts;
  float traj_cost = INFINITY;
  if (writer)
  {
    if (valid)
    {
      traj_cost = total;
      args.trajectory_costs_d[(size_t)num_rollouts * thread_idz + global_idx] = total;
    }
    cost_s[shared_idx] = traj_cost;
  }
  __syncthreads();

  /* ---- block-local softmin record ---- */
  const float lambda_inv = (float)(1.0 / (double)args.lambda);  // mppi_controller.cu:201 passes 1.0 / lambda
  // Blocks made of whole waves reduce over the (at most 64) rollouts of a system with a shuffle tree; the serial loops
  // they replace — 64 dependent LDS reads per writer for the minimum, and one thread adding 64 doubles twice for eta and
  // sum w^2 — were ~2 us of a ~30 us Cartpole iteration.
  constexpr bool WAVE_REDUCE = (NTHREADS % 64 == 0) && (BX <= 64);
  const int lane = tid_flat & 63;
  float rho_mine = INFINITY;
  if (WAVE_REDUCE)
  {
#pragma unroll
    for (int z = 0; z < BZ; z++)
    {
      const float v = mppi::wave::waveAllMin((lane < BX) ? cost_s[BX * z + lane] : INFINITY);  // DPP, no LDS round trips
      rho_mine = (z == thread_idz) ? v : rho_mine;
    }
  }
  if (writer)
  {
    float rho_b = rho_mine;
    if (!WAVE_REDUCE)
    {
      rho_b = INFINITY;
      for (int i = 0; i < BX; i++)
        rho_b = fminf(rho_b, cost_s[BX * thread_idz + i]);
    }
    // S - rho_b is NaN when every valid rollout of the block has cost +inf (inf - inf), or when S itself is NaN: such a
    // rollout gets weight 0 — what the reference's single global baseline gives it (exp(-inf)) — instead of poisoning U_b
    const float dist = traj_cost - rho_b;
    w_s[shared_idx] = (valid && dist == dist) ? mppi::det::exp(-lambda_inv * dist) : 0.0f;
  }
  __syncthreads();

  const int TC = num_timesteps * C;
  const int PS = partialStride(num_timesteps, C);
  const int num_blocks = (int)gridDim.x;
  const int row_stride = sampling->rowStrideNow();
  if constexpr (WAVE_REDUCE && (BX % 2 == 0))
  {
    // U_b[z][j] = sum_i w_i v_i[j].  A column's BX rows are split between the two half-waves (lane l and l ^ 32 take the same
    // column, rows [0, BX/2) and [BX/2, BX)) and joined with one shuffle: with one thread per column only T*C of the block's
    // threads worked — 100 of 256 for Cartpole, each walking 64 rows at 13 instructions apiece (1.7 us of the kernel's ~4.5 us
    // fixed part); now every wave walks 32 rows with the addresses advanced by hand (4 instructions a row).
    constexpr int CPP = NTHREADS / 2;  // columns per pass
    const int c = (tid_flat >> 6) * 32 + (tid_flat & 31);
    const int half = (tid_flat >> 5) & 1;
    const int i0 = half * (BX / 2);
    const int i1 = min(nrows, i0 + BX / 2);
    for (int o0 = 0; o0 < BZ * TC; o0 += CPP)
    {
      const int o = o0 + c;
      const bool ok = o < BZ * TC;
      const int oc = ok ? o : 0;
      const int z = oc / TC;
      const int j = oc - z * TC;
      const float* rows = theta_d_shared + (size_t)(BX * z + i0) * row_stride + j;
      const float* wz = w_s + BX * z + i0;
      float acc = 0.0f;
#pragma unroll 8
      for (int i = i0; i < i1; i++)
      {
        acc += wz[0] * rows[0];
        wz += 1;
        rows += row_stride;
      }
      acc += __shfl_xor(acc, 32, 64);
      if (ok && half == 0)
      {
        args.partials_d[((size_t)z * num_blocks + block_idx) * PS + j] = acc;
        if (BZ == 1 && args.records_t_d)
          args.records_t_d[((size_t)(j >> 2) * num_blocks + block_idx) * 4 + (j & 3)] = acc;
      }
    }
  }
  else
  {
    for (int o = tid_flat; o < BZ * TC; o += NTHREADS)
    {
      const int z = o / TC;
      const int j = o - z * TC;
      const float* rows = theta_d_shared + (size_t)(BX * z) * row_stride + j;
      const float* wz = w_s + BX * z;
      float acc = 0.0f;
      for (int i = 0; i < nrows; i++)
        acc += wz[i] * rows[i * row_stride];
      args.partials_d[((size_t)z * num_blocks + block_idx) * PS + j] = acc;
      if (BZ == 1 && args.records_t_d)
        args.records_t_d[((size_t)(j >> 2) * num_blocks + block_idx) * 4 + (j & 3)] = acc;
    }
  }
  if (WAVE_REDUCE)
  {
    if (tid_flat < 64)
    {  // wave 0: eta_b and sum w^2 in double by shuffle tree (invalid rows carry w = 0, cost = inf)
#pragma unroll
      for (int z = 0; z < BZ; z++)
      {
        const float rho_b = mppi::wave::waveAllMin((lane < BX) ? cost_s[BX * z + lane] : INFINITY);
        const double w = (lane < BX) ? (double)w_s[BX * z + lane] : 0.0;
        // (three six-level __shfl_xor trees — 24 dependent ds_bpermute round trips for the two doubles — were ~0.5 us of the
        //  kernel's fixed part)
        const double eta = mppi::wave::waveAllSum(w);
        const double eta2 = mppi::wave::waveAllSum(w * w);
        if (lane == 0)
        {
          float* rec = args.partials_d + ((size_t)z * num_blocks + block_idx) * PS + TC;
          rec[0] = rho_b;
          rec[1] = (float)eta;
          rec[2] = (float)eta2;
          rec[3] = 0.0f;
          if (BZ == 1 && args.records_t_d)
          {
            float* tail_t = args.records_t_d + (size_t)TC * num_blocks + 2 * block_idx;
            tail_t[0] = rho_b;
            tail_t[1] = (float)eta;
            args.records_t_d[(size_t)(TC + 2) * num_blocks + block_idx] = (float)eta2;
          }
        }
      }
    }
  }
  else if (tid_flat < BZ)
  {
    const int z = tid_flat;
    float rho_b = INFINITY;
    double eta = 0.0, eta2 = 0.0;
    for (int i = 0; i < nrows; i++)
    {
      rho_b = fminf(rho_b, cost_s[BX * z + i]);
      const double w = (double)w_s[BX * z + i];
      eta += w;
      eta2 += w * w;
    }
    float* rec = args.partials_d + ((size_t)z * num_blocks + block_idx) * PS + TC;
    rec[0] = rho_b;
    rec[1] = (float)eta;
    rec[2] = (float)eta2;
    rec[3] = 0.0f;
    if (BZ == 1 && args.records_t_d)
    {
      float* tail_t = args.records_t_d + (size_t)TC * num_blocks + 2 * block_idx;
      tail_t[0] = rho_b;
      tail_t[1] = (float)eta;
      args.records_t_d[(size_t)(TC + 2) * num_blocks + block_idx] = (float)eta2;
    }
  }
  if (args.save_samples)
  {
    // dump v[z][k][t][c] (clamped) for getSampledControl-style readers and for the RNG-mode parity tests
    for (int o = tid_flat; o < BZ * nrows * TC; o += NTHREADS)
    {
      const int z = o / (nrows * TC);
      const int r = o - z * nrows * TC;
      const int i = r / TC;
      const int j = r - i * TC;
      sampling->control_samples_d_[((size_t)z * num_rollouts + BX * block_idx + i) * TC + j] =
          theta_d_shared[(size_t)(BX * z + i) * row_stride + j];
    }
  }
}

/** trajectory cost = running/T + terminal/T (mppi_common.cu:144, :843-853), then the shared block-softmin epilogue */
template <class SAMPLING_T, int C, int BX, int BZ, int NTHREADS>
__device__ inline void blockSoftminEpilogue(SAMPLING_T* sampling, const RolloutArgs& args, const float terminal_cost,
                                            const float running_cost, const bool writer, const bool valid,
                                            const int global_idx, const int shared_idx, const int thread_idz,
                                            const int tid_flat, const int block_idx, const int nrows,
                                            float* theta_d_shared, float* cost_s, float* w_s)
{
  const float total =
      running_cost / (float)args.num_timesteps + terminal_cost / (float)args.num_timesteps;
  blockSoftminEpilogueCost<SAMPLING_T, C, BX, BZ, NTHREADS>(sampling, args, total, writer, valid, global_idx, shared_idx,
                                                             thread_idz, tid_flat, block_idx, nrows, theta_d_shared,
                                                             cost_s, w_s);
}

template <class DYN_T, class COST_T, class SAMPLING_T, int BX, int BY, int BZ, bool DRAW_IN_LOOP>
__global__ void __launch_bounds__(BX* BY* BZ* replicated_lanes<DYN_T>::value)
    rolloutKernel(DYN_T dynamics_obj, COST_T costs_obj, SAMPLING_T sampling_obj, const RolloutArgs args)
{
  // BX rollouts per block.  REP > 1 (only with one contract lane per rollout, BY == 1): REP wave lanes carry private
  // register copies of each rollout; blockDim.x = BX * REP and a wave holds 64 / REP rollouts.
  constexpr int REP = replicated_lanes<DYN_T>::value;
  static_assert(REP == 1 || (BY == 1 && 64 % REP == 0 && (BX * REP) % 64 == 0), "replicated lanes need BY == 1 and whole waves");
  // The block shape is a template parameter; telling the compiler lets the plugins' threadIdx.y / blockDim.y loops
  // fold (see mppi_amd/plugin/parallel_utils.hpp).
  __builtin_assume(__builtin_amdgcn_workgroup_size_x() == BX * REP);
  __builtin_assume(__builtin_amdgcn_workgroup_size_y() == BY);
  __builtin_assume(__builtin_amdgcn_workgroup_size_z() == BZ);
  __builtin_assume(__builtin_amdgcn_workitem_id_x() < BX * REP);
  __builtin_assume(__builtin_amdgcn_workitem_id_y() < BY);
  __builtin_assume(__builtin_amdgcn_workitem_id_z() < BZ);

  DYN_T* dynamics = &dynamics_obj;
  COST_T* costs = &costs_obj;
  SAMPLING_T* sampling = &sampling_obj;

  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  constexpr int SLOTS = BX * BZ;
  const int tid_x = (int)__builtin_amdgcn_workitem_id_x();
  int thread_idx = tid_x;  // rollout within the block
  int rep_lane = 0;        // which of the REP copies of the rollout this thread is
  if (REP > 1)
  {
    constexpr int PER_WAVE = 64 / REP;
    const int l = tid_x & 63;
    thread_idx = (tid_x >> 6) * PER_WAVE + (l % PER_WAVE);
    rep_lane = l / PER_WAVE;
  }
  const int thread_idy = (int)__builtin_amdgcn_workitem_id_y();
  const int thread_idz = (int)__builtin_amdgcn_workitem_id_z();
  const int block_idx = (int)blockIdx.x;
  const int global_idx = BX * block_idx + thread_idx;
  const int shared_idx = BX * thread_idz + thread_idx;
  const int distribution_idx = thread_idz;
  sampling->setNoiseStream(distribution_idx);  // Philox stream of this thread's draws (independent-noise option)
  const int tid_flat = tid_x + BX * REP * (thread_idy + BY * thread_idz);
  constexpr int NTHREADS = BX * REP * BY * BZ;
  const bool writer = (rep_lane == 0) && (thread_idy == 0);  // the one thread that publishes a rollout's results
  const int num_timesteps = args.num_timesteps;
  const int num_rollouts = args.num_rollouts;
  const float dt = args.dt;
  const bool valid = global_idx < num_rollouts;
  const int nrows = min(BX, num_rollouts - BX * block_idx);

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s_shared = reinterpret_cast<float*>(smem_raw);
  float* theta_c_shared = theta_s_shared + calcClassSharedMemSize(dynamics, SLOTS) / (int)sizeof(float);
  float* theta_d_lds = theta_c_shared + calcClassSharedMemSize(costs, SLOTS) / (int)sizeof(float);
  float* next_shared = theta_d_lds + calcClassSharedMemSize(sampling, SLOTS) / (int)sizeof(float);
  // the block's sample rows: in LDS, or — horizons whose rows do not fit — in the sampler's HBM buffer (generic pointers)
  float* theta_d_shared = sampling->blockRows(theta_d_lds, block_idx, SLOTS);
  sampling->setStagingBase(theta_d_lds);  // block-shared LDS of the sampler (colored noise: the table staging tiles)

  float x_priv[S], xn_priv[S], xdot_priv[S], u_priv[C], y_priv[O];
  int crash_priv = 0;
  float *x, *x_next, *xdot, *u, *y;
  int* crash_status;
  float* running_cost_shared = nullptr;
  if (BY == 1)
  {
    x = x_priv;
    x_next = xn_priv;
    xdot = xdot_priv;
    u = u_priv;
    y = y_priv;
    crash_status = &crash_priv;
  }
  else
  {
    float* x_shared = next_shared;
    float* x_next_shared = x_shared + math::nearest_multiple_4(SLOTS * S);
    float* x_dot_shared = x_next_shared + math::nearest_multiple_4(SLOTS * S);
    float* y_shared = x_dot_shared + math::nearest_multiple_4(SLOTS * S);
    float* u_shared = y_shared + math::nearest_multiple_4(SLOTS * O);
    running_cost_shared = u_shared + math::nearest_multiple_4(SLOTS * C);
    int* crash_status_shared = (int*)(running_cost_shared + math::nearest_multiple_4(NTHREADS));
    next_shared = (float*)(crash_status_shared + math::nearest_multiple_4(SLOTS));
    x = &x_shared[shared_idx * S];
    x_next = &x_next_shared[shared_idx * S];
    xdot = &x_dot_shared[shared_idx * S];
    u = &u_shared[shared_idx * C];
    y = &y_shared[shared_idx * O];
    crash_status = &crash_status_shared[shared_idx];
  }
  float* cost_s = next_shared;
  float* w_s = cost_s + math::nearest_multiple_4(SLOTS);

  // loadGlobalToShared (mppi_common.cu:770-840): x <- x0[z], xdot <- 0, u <- 0
  for (int i = thread_idy; i < S; i += BY)
  {
    x[i] = args.init_x_d[S * thread_idz + i];
    xdot[i] = 0.0f;
    if (BY == 1)
      x_next[i] = 0.0f;
  }
  for (int i = thread_idy; i < C; i += BY)
    u[i] = 0.0f;
  for (int i = thread_idy; i < O; i += BY)
    y[i] = 0.0f;
  if (thread_idy == 0)
    crash_status[0] = 0;
  __syncthreads();

  if (REP > 1)
    sampling->setThreadMapping(shared_idx, BX);

  /*<----Start of simulation loop-----> */
  dynamics->initializeDynamics(x, u, y, theta_s_shared, 0.0f, dt);
  sampling->initializeDistributions(y, 0.0f, dt, theta_d_shared);
  costs->initializeCosts(y, u, theta_c_shared, 0.0f, dt);
  __syncthreads();

  static_assert(!DRAW_IN_LOOP || BY == 1, "the in-loop draw needs thread-private controls (BY == 1)");
  float running_cost = 0.0f;
  // eps == nullptr: the sample comes from the pre-filled LDS row (contract path); otherwise from registers
  auto one_step = [&](float* xc, float* xn, int t, const float* eps) {
    if (DRAW_IN_LOOP)
      sampling->shapeControlSample(global_idx, t, distribution_idx, eps, u);
    else
      sampling->readControlSample(global_idx, t, distribution_idx, u, theta_d_shared, BY, thread_idy, y);
    lane_sync();
    dynamics->enforceConstraints(xc, u);
    lane_sync();
    // every replica of a rollout writes (same word, same value): no region that narrows EXEC to one replica in the step loop
    // (see rolloutPipelineRepKernel)
    sampling->writeControlSample(global_idx, t, distribution_idx, u, theta_d_shared, BY, thread_idy, y);
    dynamics->step(xc, xn, xdot, u, y, theta_s_shared, t, dt);
    lane_sync();
    running_cost += costs->computeRunningCost(y, u, t, theta_c_shared, crash_status) +
                    sampling->computeLikelihoodRatioCost(u, theta_d_shared, global_idx, t, distribution_idx,
                                                         args.lambda, args.alpha);
    lane_sync();
  };
  // STEPS steps per trip (even, so that x / x_next keep fixed roles and stay in registers when BY == 1) chosen such
  // that a trip consumes whole Philox quads whose lanes are indexed statically: 4 steps for odd C, 2 for even C.
  constexpr int STEPS = (C % 2 == 0) ? 2 : 4;
  constexpr int QUADS = STEPS * C / 4;
  static_assert(STEPS * C % 4 == 0, "a loop trip must consume whole quads");
  int t = 0;
  if constexpr (REP > 1 && REP % 2 == 0)
  {
    /* Replicated-lane (MFMA) dynamics: a wave holds its 64 / REP rollouts REP times over, and only the dynamics need the
     * replicas — one_step() would shape the sample and evaluate the cost REP times redundantly.  Instead the replicas
     * share that work in TIME:
     *   - sampler: lane (rollout j, replica r) shapes, clamps and stores the samples of steps t0 + STEPS * r .. of
     *     rollout j — one pass per STEPS * REP steps with all 64 lanes busy;
     *   - cost: after REP dynamics steps lane (j, r) evaluates the cost of step t + r from the outputs the wave kept.
     *     The running cost is accumulated in step order from the replicas (REP cross-lane reads): the sum the
     *     sequential code forms, bit for bit.
     * The plugin contract threads an integer status (crash flags) through computeRunningCost: step t + r must see what
     * steps t .. t + r - 1 left behind.  The slices run with the status at the start of the group; if one of the first
     * REP - 1 slices CHANGES it, the wave redoes that group step by step on every lane — for the sticky flags of the
     * reference's costs at most once per rollout.  (computeRunningCost has to be a pure function of output, control, t
     * and status, which the reference's cost classes are.)  AutoRally-NN, K = 16384, T = 150: 355 -> 257 us. */
    static_assert(BY == 1, "replicated lanes run with one contract lane");
    constexpr int PER_WAVE = 64 / REP;
    constexpr bool SMP_CONSTRAINS = !DYN_T::CONSTRAINTS_DEPEND_ON_STATE;
    constexpr int SGROUP = STEPS * REP;  // steps one sampling pass of the wave covers
    const int col = (tid_x & 63) % PER_WAVE;
    float* row = sampling->sampleRow(theta_d_shared, shared_idx);
    auto sample_pass = [&](const int t0) {
      const int ts = t0 + STEPS * rep_lane;
      float zq[4 * QUADS], us[C];
      if (DRAW_IN_LOOP)
      {
#pragma unroll
        for (int q = 0; q < QUADS; q++)
          sampling->drawQuad(global_idx, ts * C / 4 + q, &zq[4 * q]);
      }
#pragma unroll
      for (int s2 = 0; s2 < STEPS; s2++)
      {
        if (ts + s2 < num_timesteps)
        {
          if (DRAW_IN_LOOP)
            sampling->shapeControlSample(global_idx, ts + s2, distribution_idx, &zq[s2 * C], us);
          else
            sampling->readControlSample(global_idx, ts + s2, distribution_idx, us, theta_d_shared, 1, 0, y);
          if (SMP_CONSTRAINS)
            dynamics->enforceConstraints(x, us);
          sampling->writeControlSample(global_idx, ts + s2, distribution_idx, us, theta_d_shared, 1, 0, y);
        }
      }
      // the other replicas of the rollout read these rows: LDS operations of a wave execute in order, the fence keeps
      // the compiler (which reasons per lane) from moving a lane's row reads above its own, differently addressed, stores
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    };
    auto dyn_step = [&](float* xc, float* xn, int tt, const float* u_in) {
#pragma unroll
      for (int i = 0; i < C; i++)
        u[i] = u_in[i];
      if (!SMP_CONSTRAINS)
      {
        dynamics->enforceConstraints(xc, u);
#pragma unroll
        for (int i = 0; i < C; i++)
          row[tt * C + i] = u[i];  // all replicas: same word, same value
      }
      dynamics->step(xc, xn, xdot, u, y, theta_s_shared, tt, dt);
    };
    auto step_cost = [&](float* ys, int tt, int* status) {
      float us[C];
#pragma unroll
      for (int i = 0; i < C; i++)
        us[i] = row[tt * C + i];
      return costs->computeRunningCost(ys, us, tt, theta_c_shared, status) +
             sampling->computeLikelihoodRatioCost(us, theta_d_shared, global_idx, tt, distribution_idx, args.lambda,
                                                  args.alpha);
    };
    int sampled = 0;  // steps whose shaped samples are in the rows (a multiple of SGROUP, hence of REP)
    for (; t + REP <= num_timesteps; t += REP)
    {
      if (t >= sampled)
      {
        sample_pass(sampled);
        sampled += SGROUP;
      }
      float ubuf[REP * C];
#pragma unroll
      for (int j = 0; j < REP * C; j++)
        ubuf[j] = row[t * C + j];
      float ybuf[REP][O];
#pragma unroll
      for (int r = 0; r < REP; r += 2)
      {
        dyn_step(x, x_next, t + r, &ubuf[r * C]);
#pragma unroll
        for (int i = 0; i < O; i++)
          ybuf[r][i] = y[i];
        dyn_step(x_next, x, t + r + 1, &ubuf[(r + 1) * C]);
#pragma unroll
        for (int i = 0; i < O; i++)
          ybuf[r + 1][i] = y[i];
      }
      if (!SMP_CONSTRAINS)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // replica 0 stored the clamped controls
      float ys[O];
#pragma unroll
      for (int i = 0; i < O; i++)
      {
        ys[i] = ybuf[0][i];
#pragma unroll
        for (int r = 1; r < REP; r++)
          ys[i] = (rep_lane == r) ? ybuf[r][i] : ys[i];
      }
      int status = crash_status[0];
      const float val = step_cost(ys, t + rep_lane, &status);
      const bool stale = (rep_lane < REP - 1) && (status != crash_status[0]);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(stale) != 0, 0))
      {
        // a slice changed the status under a later slice of the same group: this group again, in order, on every lane
#pragma nounroll
        for (int r = 0; r < REP; r++)
        {
#pragma unroll
          for (int i = 0; i < O; i++)
          {
            ys[i] = ybuf[0][i];
#pragma unroll
            for (int q = 1; q < REP; q++)
              ys[i] = (r == q) ? ybuf[q][i] : ys[i];
          }
          running_cost += step_cost(ys, t + r, crash_status);
        }
      }
      else
      {
#pragma unroll
        for (int r = 0; r < REP; r++)
          running_cost += __shfl(val, r * PER_WAVE + col, 64);
        crash_status[0] = __shfl(status, (REP - 1) * PER_WAVE + col, 64);
      }
    }
    // the last num_timesteps % REP steps: step by step, cost on every lane
    for (; t < num_timesteps; t++)
    {
      if (t >= sampled)
      {
        sample_pass(sampled);
        sampled += SGROUP;
      }
      float ubuf[C];
#pragma unroll
      for (int j = 0; j < C; j++)
        ubuf[j] = row[t * C + j];
      dyn_step(x, x_next, t, ubuf);
#pragma unroll
      for (int i = 0; i < S; i++)
        x[i] = x_next[i];
      running_cost += step_cost(y, t, crash_status);
    }
  }
  else if constexpr (single_step_site<DYN_T>::value && BY == 1)
  {
    for (; t < num_timesteps; t += STEPS)
    {
      float zq[4 * QUADS];
      if (DRAW_IN_LOOP)
      {
#pragma unroll
        for (int q = 0; q < QUADS; q++)
          sampling->drawQuad(global_idx, t * C / 4 + q, &zq[4 * q]);
      }
#pragma unroll 1
      for (int s2 = 0; s2 < STEPS; s2++)
      {
        if (t + s2 < num_timesteps)
        {
          float eps[C];
#pragma unroll
          for (int i = 0; i < C; i++)
          {  // zq[s2 * C + i] without a run-time index into the register array
            float v = zq[i];
#pragma unroll
            for (int q = 1; q < STEPS; q++)
              v = (s2 == q) ? zq[q * C + i] : v;
            eps[i] = v;
          }
          one_step(x, x_next, t + s2, eps);
#pragma unroll
          for (int i = 0; i < S; i++)
            x[i] = x_next[i];
        }
      }
    }
  }
  else
  {
  for (; t + STEPS - 1 < num_timesteps; t += STEPS)
  {
    float zq[4 * QUADS];
    if (DRAW_IN_LOOP)
    {
#pragma unroll
      for (int q = 0; q < QUADS; q++)
        sampling->drawQuad(global_idx, t * C / 4 + q, &zq[4 * q]);
    }
#pragma unroll
    for (int s2 = 0; s2 < STEPS; s2 += 2)
    {
      one_step(x, x_next, t + s2, &zq[s2 * C]);
      one_step(x_next, x, t + s2 + 1, &zq[(s2 + 1) * C]);
    }
  }
  if (t < num_timesteps)
  {
    // tail of 1..STEPS-1 steps: the same quads, the unused lanes are simply not consumed
    float zq[4 * QUADS];
    if (DRAW_IN_LOOP)
    {
#pragma unroll
      for (int q = 0; q < QUADS; q++)
        sampling->drawQuad(global_idx, t * C / 4 + q, &zq[4 * q]);
    }
    const int rem = num_timesteps - t;
    one_step(x, x_next, t, &zq[0 * C]);
    if (STEPS > 2)
    {
      if (rem > 1)
        one_step(x_next, x, t + 1, &zq[1 * C]);
      if (rem > 2)
        one_step(x, x_next, t + 2, &zq[(STEPS > 2 ? 2 : 0) * C]);
    }
    // (an odd number of steps leaves the newest state in x_next; the epilogue reads y, the output, not x)
  }

  }

  /* ---- cost of the rollout: sum over the y lanes, running/T + terminal/T ---- */
  if (BY > 1)
  {
    running_cost_shared[thread_idx + BX * (thread_idy + BY * thread_idz)] = running_cost;
    __syncthreads();
    if (thread_idy == 0)
    {
      float acc = 0.0f;
      for (int j = 0; j < BY; j++)
        acc += running_cost_shared[thread_idx + BX * (j + BY * thread_idz)];
      running_cost = acc;
    }
  }
  blockSoftminEpilogue<SAMPLING_T, C, BX, BZ, NTHREADS>(sampling, args, costs->terminalCost(y, theta_c_shared), running_cost,
                                                        writer, valid, global_idx, shared_idx, thread_idz, tid_flat,
                                                        block_idx, nrows, theta_d_shared, cost_s, w_s);
}

}  // namespace kernels
}  // namespace mppi

#endif
