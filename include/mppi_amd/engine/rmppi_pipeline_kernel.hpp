/**
 * rmppi_pipeline_kernel.hpp — the Robust MPPI rollout (two coupled systems per rollout) as a pipeline of role-specialised
 * waves, for dynamics with replicated lanes (REP wave lanes per rollout: the MFMA / four-lane forms).
 *
 * Why: rolloutRMPPIKernel (rmppi_kernels.hpp) gives every rollout of every system REP lanes for the whole step, so
 * everything that is NOT the dynamics — the Philox draw, the sample shaping, the cost terms, the write-back of the
 * fed-back control — is issued for 64 / REP rollouts per instruction instead of 64, and a block-wide barrier per step
 * hands the nominal state to the real system.  With the SIMD's issue slots as the bound (AutoRally-NN, K = 16384, T = 150:
 * ~700 instructions per step and wave, two waves per SIMD, 618 us per launch — the step loop's instruction count times the
 * issue interval) the redundant work is what there is to remove.  A block of 64 rollouts x 2 systems runs as
 *
 *   2 x REP dynamics waves  (nominal: waves 0 .. REP-1, real: REP .. 2 REP-1; 64 / REP rollouts each) — fetch the shaped
 *                           sample, nominal: publish the state x*_t / real: u += K_t (x_t - x*_t), clamp, step, push
 *                           (y_t, u_t, feedback term) into the system's output ring        -> rmppi_kernels.cu:741-796
 *   NS sampler waves        (64 rollouts, BOTH systems: with the reference's default — the same noise for all
 *                           distributions — one Philox draw serves both)                   -> readControlSample
 *   2 x NC cost waves       (64 rollouts of one system) — running cost, likelihood-ratio and feedback cost of the two
 *                           accumulators (rmppi_kernels.cu:797-812), evaluated AHEAD of the relay exactly as in
 *                           rolloutPipelineRepKernel; they also write the fed-back, clamped control back into the sample
 *                           row (rmppi_kernels.cu:780-781) — one instruction per 64 rollouts
 *
 * coupled by monotonic progress counters in LDS (rollout_pipeline_kernel.hpp).  The block-wide barrier per step is gone:
 * the real wave of a group of rollouts waits for the nominal wave of the same rollouts only (it trails it by one pair of
 * steps), through a ring of nominal states.  Every rollout is evaluated with the plugin calls, in the order, of
 * rolloutRMPPIKernel: trajectory costs and control updates are the same bits (tests/test_rmppi.py compares the two).
 *
 * The sample rows (all T steps of 128 rollout-systems) live where the sampler says (blockRows): in HBM for the benchmark
 * horizon — next to the rings they do not fit the LDS — or in LDS for short horizons.
 */
#ifndef MPPI_AMD_RMPPI_PIPELINE_KERNEL_HPP_
#define MPPI_AMD_RMPPI_PIPELINE_KERNEL_HPP_

#include "rmppi_kernels.hpp"
#include "rollout_pipeline_kernel.hpp"
#include "kernarg_view.hpp"

namespace mppi
{
namespace kernels
{
/* helper waves per block; A/B builds: -DMPPI_RMPPI_PIPE_NS=.. -DMPPI_RMPPI_PIPE_NC=.. (cost waves PER SYSTEM).
 * AutoRally-NN, K = 16384, T = 150, Robust computeControl (tools/robust_latency.py, one session): NS / NC = 2 / 2 690 us,
 * 1 / 2 684, 2 / 1 736, 2 / 3 637; after the last-state feedback path 2 / 2 640, 1 / 3 622, 2 / 3 624.  The cost waves'
 * chain (a pair of steps is evaluated by one wave) is what a third wave per system shortens; one sampler keeps up with both
 * systems (one Philox draw per pair of steps serves both).  15 waves = 960 threads per block. */
#if !defined(MPPI_RMPPI_PIPE_NS)
#define MPPI_RMPPI_PIPE_NS 1
#endif
#if !defined(MPPI_RMPPI_PIPE_NC)
#define MPPI_RMPPI_PIPE_NC 3
#endif
/** replicated-lane dynamics: the tuned counts above.  One lane per rollout (analytic models: a dynamics wave carries 64
 *  rollouts and a step is ~40 instructions): what sets the pace there is the cost waves' chain again — double integrator,
 *  K = 8192, T = 150, Robust computeControl with NS1 / NC1 = 2 / 1 144 us, 3 / 1 144, 2 / 2 99, 4 / 2 98, 2 / 3 89, 2 / 4 88
 *  (the fused kernel: 253 us).  A/B builds: -DMPPI_RMPPI_PIPE_NS1=.. -DMPPI_RMPPI_PIPE_NC1=.. */
#if !defined(MPPI_RMPPI_PIPE_NS1)
#define MPPI_RMPPI_PIPE_NS1 2
#endif
#if !defined(MPPI_RMPPI_PIPE_NC1)
#define MPPI_RMPPI_PIPE_NC1 3
#endif
/** A dynamics class may name its own helper-wave counts for this kernel (`static constexpr int MPPI_RMPPI_PIPE_SAMPLERS = ..,
 *  MPPI_RMPPI_PIPE_COSTS = ..` — cost waves PER SYSTEM): the counts above were tuned on models whose step is short (AutoRally-NN:
 *  the cost waves' chain sets the pace).  A model whose dynamics step is thousands of instructions wants the opposite trade —
 *  FEWER waves per block, because the block's wave count sets the register file share of every wave (15 waves = 4 per SIMD =
 *  128 VGPRs each; 11 waves = 3 per SIMD = 168) and its dynamics waves spill. */
template <class T, class = void>
struct rmppi_pipe_samplers_of : std::integral_constant<int, 0>
{
};
template <class T>
struct rmppi_pipe_samplers_of<T, std::void_t<decltype(T::MPPI_RMPPI_PIPE_SAMPLERS)>> : std::integral_constant<int, T::MPPI_RMPPI_PIPE_SAMPLERS>
{
};
template <class T, class = void>
struct rmppi_pipe_costs_of : std::integral_constant<int, 0>
{
};
template <class T>
struct rmppi_pipe_costs_of<T, std::void_t<decltype(T::MPPI_RMPPI_PIPE_COSTS)>> : std::integral_constant<int, T::MPPI_RMPPI_PIPE_COSTS>
{
};
template <class T, class = void>
struct rmppi_cost_view_of : std::false_type
{
};
template <class T>
struct rmppi_cost_view_of<T, std::void_t<decltype(T::MPPI_RMPPI_COST_VIEW)>> : std::integral_constant<bool, T::MPPI_RMPPI_COST_VIEW>
{
};
template <class DYN_T>
__host__ __device__ constexpr int rmppiPipeSamplers()
{
  if (rmppi_pipe_samplers_of<DYN_T>::value > 0)
    return rmppi_pipe_samplers_of<DYN_T>::value;
  return replicated_lanes<DYN_T>::value > 1 ? MPPI_RMPPI_PIPE_NS : MPPI_RMPPI_PIPE_NS1;
}
template <class DYN_T>
__host__ __device__ constexpr int rmppiPipeCosts()
{
  if (rmppi_pipe_costs_of<DYN_T>::value > 0)
    return rmppi_pipe_costs_of<DYN_T>::value;
  return replicated_lanes<DYN_T>::value > 1 ? MPPI_RMPPI_PIPE_NC : MPPI_RMPPI_PIPE_NC1;
}

/** ring depths in steps, powers of two: outputs (dynamics -> cost), shaped samples (sampler -> dynamics), nominal states */
struct RMPPIPipeRings
{
  int out_steps;
  int sample_steps;
  int xnom_steps;
};

template <class DYN_T>
__host__ __device__ constexpr int rmppiPipelineWaves()
{
  return 2 * replicated_lanes<DYN_T>::value + rmppiPipeSamplers<DYN_T>() + 2 * rmppiPipeCosts<DYN_T>();
}

template <class DYN_T, class COST_T, class FB_T, class SAMPLING_T>
__host__ inline size_t rmppiPipelineSharedBytes(const DYN_T& dyn, const COST_T& cost, const FB_T& fb, const SAMPLING_T& smp,
                                                const RMPPIPipeRings r)
{
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  const int slots = 128;
  size_t n = 0;
  n += calcClassSharedMemSize(&dyn, slots);
  n += calcClassSharedMemSize(&cost, slots);
  n += calcClassSharedMemSize(&smp, slots);
  n += calcClassSharedMemSize(&fb, slots);
  n += sizeof(float) * 2 * (size_t)r.out_steps * (O + 2 * C) * 64;   // [z][slot][y | u | fb][rollout]
  n += sizeof(float) * 2 * (size_t)r.sample_steps * C * 64;          // [z][slot][c][rollout]
  n += sizeof(float) * (size_t)r.xnom_steps * S * 64;                // [slot][s][rollout]
  n += sizeof(float) * 7 * math::nearest_multiple_4(slots);          // cost_s, w_s, acc_a_s, acc_b_s, relay a / b / status
  n += sizeof(int) * 4 * (2 * replicated_lanes<DYN_T>::value + rmppiPipeSamplers<DYN_T>() + 2);  // progress counters (padded)
  return n;
}

/** deepest rings that fit: the output ring is the wide one (O + 2 C floats per step, rollout and system) */
template <class DYN_T, class COST_T, class FB_T, class SAMPLING_T>
__host__ inline RMPPIPipeRings rmppiPipelineRings(const DYN_T& dyn, const COST_T& cost, const FB_T& fb, const SAMPLING_T& smp,
                                                  size_t max_lds)
{
  for (int out = 16; out >= 4; out >>= 1)
  {
    const RMPPIPipeRings r{ out, 2 * out, 8 };
    if (rmppiPipelineSharedBytes(dyn, cost, fb, smp, r) <= max_lds)
      return r;
  }
  return RMPPIPipeRings{ 0, 0, 0 };
}

/**
 * The REP replica lanes of a rollout hold the same N values; replica r stores the fields r, r + REP, ... at
 * base[field * 64] (fields past the end: field j * REP again, same value).  Every lane of the wave stays active and the
 * stores of a replica group go to distinct addresses.  The dynamics waves of this kernel contain NO region that only the
 * lanes of one replica execute (`if (rep_lane == 0) store`): the kernel runs at the register limit of a 1024-thread block and
 * spills, and builds WITH such regions gave the suspension / complete RACER models covariance outputs a few ulp off in every
 * rollout from the second step on (deterministic per build; builds without them are exact).  The instruction at fault was
 * not identified — see DESIGN.md §5 for what was tried — so the pattern is avoided here.
 */
template <int REP, int N>
__device__ inline void stripedStore(float* base, const float (&vals)[N], const int rep_lane)
{
  // The selects below work on a COPY of the values, not on the caller's array.  Straight on `vals`, the optimiser folds
  // `select(c, load vals[j + q], load vals[j])` into `load vals[select(...)]` — the caller's array indexed by a lane-dependent
  // value, which cannot live in registers: the state / record arrays of the dynamics waves went to SCRATCH MEMORY (7 + 9
  // floats per lane and step in the Robust AutoRally kernel: the 64 B of private segment and the 1.45x HBM traffic that round 5
  // attributed to spilled kernel arguments).  With the copy: 0 B, 114.7 -> 83.5 MB per launch = 1.06x the algorithmic bytes
  // (profiles/r06_robust_hbm_traffic_pmc.json), the Robust complete RACER kernel 3399 -> 3099 us.  (Pinning the copy to VGPRs
  // with an empty asm as well was 2 % slower on the AutoRally kernel: 403 against 396 us.)
  float r[N];
#pragma unroll
  for (int j = 0; j < N; j++)
    r[j] = vals[j];
#pragma unroll
  for (int j = 0; j < N; j += REP)
  {
    float v = r[j];
    int field = j;
#pragma unroll
    for (int q = 1; q < REP; q++)
    {
      if (j + q < N)
      {
        v = (rep_lane == q) ? r[j + q] : v;
        field = (rep_lane == q) ? j + q : field;
      }
    }
    base[field * 64] = v;
  }
}

template <class DYN_T, class COST_T, class FB_T, class SAMPLING_T, bool DRAW_IN_LOOP>
__global__ void __launch_bounds__(64 * rmppiPipelineWaves<DYN_T>())
    rolloutRMPPIPipelineKernel(DYN_T dynamics_obj, COST_T costs_obj, FB_T fb_obj, SAMPLING_T sampling_obj, const RMPPIArgs rargs,
                               const RMPPIPipeRings rings)
{
  constexpr int BX = 64, BZ = 2;
  constexpr int REP = replicated_lanes<DYN_T>::value;
  static_assert(REP >= 1 && 64 % REP == 0, "whole waves of replica groups");
  constexpr int DW = BX * REP / 64;  // dynamics waves per system (one lane per rollout: 1)
  constexpr int NS = rmppiPipeSamplers<DYN_T>(), NC = rmppiPipeCosts<DYN_T>();
  // wave order: nominal dynamics 0 .. DW-1, real dynamics DW .. 2 DW-1 (the CU deals a workgroup's waves to its four SIMDs
  // in turn: with DW = 4 the two systems of a group of rollouts share a SIMD), then the samplers, then the cost waves
  constexpr int NWAVES = rmppiPipelineWaves<DYN_T>();
  constexpr int NTHREADS = 64 * NWAVES;
  constexpr int PER_WAVE = 64 / REP;
  __builtin_assume(__builtin_amdgcn_workgroup_size_x() == NTHREADS);
  __builtin_assume(__builtin_amdgcn_workgroup_size_y() == 1);
  __builtin_assume(__builtin_amdgcn_workgroup_size_z() == 1);
  __builtin_assume(__builtin_amdgcn_workitem_id_x() < NTHREADS);
  __builtin_assume(__builtin_amdgcn_workitem_id_y() == 0);
  __builtin_assume(__builtin_amdgcn_workitem_id_z() == 0);

  const RolloutArgs& args = rargs.base;
  DYN_T* dynamics = &dynamics_obj;
  COST_T* costs = &costs_obj;
  FB_T* fb_controller = &fb_obj;
  SAMPLING_T* sampling = &sampling_obj;
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  constexpr int F = O + 2 * C;  // floats per step and rollout in the output ring: y | u | feedback term
  constexpr int SLOTS = BX * BZ;

  const int tid_x = (int)__builtin_amdgcn_workitem_id_x();
  const int wave = __builtin_amdgcn_readfirstlane(tid_x >> 6);  // wave-uniform
  const int lane = tid_x & 63;
  const bool is_dyn = wave < 2 * DW;
  const bool is_sampler = !is_dyn && wave < 2 * DW + NS;
  const int helper = wave - 2 * DW - NS;  // cost waves: 0 .. 2 NC - 1, systems alternating
  // the system this wave works for (samplers: both) and which of its NC cost waves it is
  const int thread_idz = is_dyn ? wave / DW : (is_sampler ? 0 : (helper & 1));
  const int cost_id = helper >> 1;
  const bool is_nominal = thread_idz == RMPPI_NOMINAL_IDX;
  // rollout slot of this thread: dynamics waves carry PER_WAVE rollouts x REP lanes, the helpers one lane per rollout
  const int thread_idx = is_dyn ? (wave % DW) * PER_WAVE + (lane % PER_WAVE) : lane;
  const int rep_lane = is_dyn ? lane / PER_WAVE : 0;
  const int block_idx = (int)blockIdx.x;
  const int global_idx = BX * block_idx + thread_idx;
  const int shared_idx = BX * thread_idz + thread_idx;
  const int num_timesteps = args.num_timesteps;
  const int num_rollouts = args.num_rollouts;
  const float dt = args.dt;
  const bool valid = global_idx < num_rollouts;
  const int nrows = min(BX, num_rollouts - BX * block_idx);
  const int out_mask = rings.out_steps - 1, smp_mask = rings.sample_steps - 1, xnom_mask = rings.xnom_steps - 1;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s_shared = reinterpret_cast<float*>(smem_raw);
  float* theta_c_shared = theta_s_shared + calcClassSharedMemSize(dynamics, SLOTS) / (int)sizeof(float);
  float* theta_d_lds = theta_c_shared + calcClassSharedMemSize(costs, SLOTS) / (int)sizeof(float);
  float* theta_fb = theta_d_lds + calcClassSharedMemSize(sampling, SLOTS) / (int)sizeof(float);
  float* out_ring = theta_fb + calcClassSharedMemSize(fb_controller, SLOTS) / (int)sizeof(float);  // [z][slot][F][64]
  float* smp_ring = out_ring + (size_t)BZ * rings.out_steps * F * 64;                               // [z][slot][C][64]
  float* xnom_ring = smp_ring + (size_t)BZ * rings.sample_steps * C * 64;                           // [slot][S][64]
  float* cost_s = xnom_ring + (size_t)rings.xnom_steps * S * 64;
  float* w_s = cost_s + math::nearest_multiple_4(SLOTS);
  float* acc_a_s = w_s + math::nearest_multiple_4(SLOTS);
  float* acc_b_s = acc_a_s + math::nearest_multiple_4(SLOTS);
  float* relay_a = acc_b_s + math::nearest_multiple_4(SLOTS);  // the two accumulators and the status word handed from cost
  float* relay_b = relay_a + math::nearest_multiple_4(SLOTS);  // wave to cost wave, [z][rollout]
  int* relay_status = reinterpret_cast<int*>(relay_b + math::nearest_multiple_4(SLOTS));
  lds_counter_t counters = (lds_counter_t)(relay_status + math::nearest_multiple_4(SLOTS));
  // counters + 4 * w, w < 2 DW: steps dynamics wave w has completed (outputs in the ring; nominal: states x*_0 .. x*_{n-1} out)
  // counters + 4 * (2 DW + s): sampler s — steps (of ITS trips) whose shaped samples are in the sample ring
  // counters + 4 * (2 DW + NS + z): steps the cost waves of system z have consumed
  lds_counter_t cost_prog0 = counters + 4 * (2 * DW + NS);
  lds_counter_t cost_prog = cost_prog0 + 4 * thread_idz;
  // the block's sample rows: LDS, or the sampler's HBM buffer (see rolloutKernel)
  float* theta_d_shared = sampling->blockRows(theta_d_lds, block_idx, SLOTS);
  sampling->setStagingBase(theta_d_lds);
  sampling->setThreadMapping(shared_idx, BX, BZ);
  sampling->setNoiseStream(thread_idz);

  float x[S], x_next[S], xdot[S], u[C], y[O], fb_control[C];
  int crash_status = 0;
#pragma unroll
  for (int i = 0; i < S; i++)
  {
    x[i] = args.init_x_d[S * thread_idz + i];
    xdot[i] = 0.0f;
    x_next[i] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < C; i++)
  {
    u[i] = 0.0f;
    fb_control[i] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < O; i++)
    y[i] = 0.0f;
  if (tid_x < 2 * DW + NS + 2)
    counters[4 * tid_x] = 0;
  __syncthreads();

  dynamics->initializeDynamics(x, u, y, theta_s_shared, 0.0f, dt);
  sampling->initializeDistributions(y, 0.0f, dt, theta_d_shared);
  costs->initializeCosts(y, u, theta_c_shared, 0.0f, dt);
  fb_controller->initializeFeedback(x, u, theta_fb, 0.0f, dt);
  __syncthreads();

  float acc_a = 0.0f, acc_b = 0.0f;
  constexpr int STEPS = (C % 2 == 0) ? 2 : 4;  // steps per sampler trip (whole Philox quads)

  if (is_sampler)
  {
    /* -------------------------------------------------- sampler waves ------------------------------------------------- */
    // sampler s takes trips s, s + NS, ...; a slot of the sample ring is free again when the cost waves of BOTH systems have
    // consumed the step that last used it (they trail the dynamics waves)
    constexpr int QUADS = STEPS * C / 4;
    const int sid = wave - 2 * DW;
    lds_counter_t smp_prog = counters + 4 * (2 * DW + sid);
    int seen_c0 = 0, seen_c1 = 0;
    for (int t = STEPS * sid; t < num_timesteps; t += STEPS * NS)
    {
      const int hi = min(t + STEPS, num_timesteps);
      pipeWait(cost_prog0, hi - rings.sample_steps, seen_c0);
      pipeWait(cost_prog0 + 4, hi - rings.sample_steps, seen_c1);
      float zq[BZ][4 * QUADS];
      if (DRAW_IN_LOOP)
      {
        sampling->setNoiseStream(0);
#pragma unroll
        for (int q = 0; q < QUADS; q++)
          sampling->drawQuad(global_idx, t * C / 4 + q, &zq[0][4 * q]);
        if (sampling->independentNoise())
        {  // wave-uniform: one Philox stream per distribution (gaussian.cu:378-394)
          sampling->setNoiseStream(1);
#pragma unroll
          for (int q = 0; q < QUADS; q++)
            sampling->drawQuad(global_idx, t * C / 4 + q, &zq[1][4 * q]);
        }
        else
        {
#pragma unroll
          for (int j = 0; j < 4 * QUADS; j++)
            zq[1][j] = zq[0][j];
        }
      }
#pragma unroll
      for (int z = 0; z < BZ; z++)
      {
        if (!DRAW_IN_LOOP)
          sampling->setThreadMapping(BX * z + lane, BX, BZ);  // readControlSample reads the eps of THIS system's row
#pragma unroll
        for (int s2 = 0; s2 < STEPS; s2++)
        {
          if (t + s2 < num_timesteps)
          {
            if (DRAW_IN_LOOP)
              sampling->template shapeControlSample<true>(global_idx, t + s2, z, &zq[z][s2 * C], u);
            else
              sampling->template readControlSample<true>(global_idx, t + s2, z, u, theta_d_shared, 1, 0, y);
            float* slot = smp_ring + ((size_t)(z * rings.sample_steps + ((t + s2) & smp_mask)) * C) * 64 + lane;
#pragma unroll
            for (int i = 0; i < C; i++)
              slot[i * 64] = u[i];
          }
        }
      }
      pipePublishLds(smp_prog, hi, lane);
    }
  }
  else if (is_dyn)
  {
    /* -------------------------------------------------- dynamics waves ------------------------------------------------ */
    lds_counter_t my_prog = counters + 4 * wave;
    // the wave that carries the same rollouts in the other system
    lds_counter_t peer_prog = counters + 4 * (is_nominal ? wave + DW : wave - DW);
    float* my_out = out_ring + (size_t)thread_idz * rings.out_steps * F * 64;
    const float* my_smp = smp_ring + (size_t)thread_idz * rings.sample_steps * C * 64;
    const bool last_state_only = fb_controller->lastStateOnly();
    auto dyn_step = [&](float* xc, float* xn, int t, const float* u_in) {
#pragma unroll
      for (int i = 0; i < C; i++)
      {
        u[i] = u_in[i];
        fb_control[i] = 0.0f;
      }
      // (all stores of the dynamics waves are issued by every lane: see stripedStore)
      float* xs = xnom_ring + (size_t)(t & xnom_mask) * S * 64 + thread_idx;
      if (last_state_only)
      {  // wave-uniform: the reference's gain product keeps the last state's term only (ddp_feedback.hpp)
        if (is_nominal)
          xs[(S - 1) * 64] = xc[S - 1];  // the replicas write the same value to the same word
        else
          fb_controller->kLastState(xc[S - 1], xs[(S - 1) * 64], t, fb_control);
      }
      else if (is_nominal)
      {  // wave-uniform
        float xv[S];
#pragma unroll
        for (int i = 0; i < S; i++)
          xv[i] = xc[i];
        stripedStore<REP>(xs, xv, rep_lane);
      }
      else
      {
        float x_nom[S];
#pragma unroll
        for (int i = 0; i < S; i++)
          x_nom[i] = xs[i * 64];
        fb_controller->k(xc, x_nom, t, theta_fb, fb_control);
      }
#pragma unroll
      for (int i = 0; i < C; i++)
        u[i] += fb_control[i];
      dynamics->enforceConstraints(xc, u);
      dynamics->step(xc, xn, xdot, u, y, theta_s_shared, t, dt);
      float rec[F];  // y | u | feedback term (zero on the nominal system)
#pragma unroll
      for (int i = 0; i < O; i++)
        rec[i] = y[i];
#pragma unroll
      for (int i = 0; i < C; i++)
      {
        rec[O + i] = u[i];
        rec[O + C + i] = fb_control[i];
      }
      stripedStore<REP>(my_out + (size_t)(t & out_mask) * F * 64 + thread_idx, rec, rep_lane);
    };
    int seen_smp[NS], seen_cost = 0, seen_peer = 0;
#pragma unroll
    for (int q = 0; q < NS; q++)
      seen_smp[q] = 0;
    // the pair (t, t + 1) lies inside one sampler trip (STEPS is even): wait for the sampler that owns the trip
    auto wait_samples = [&](const int t, const int need) {
      const int owner = (t / STEPS) % NS;
#pragma unroll
      for (int q = 0; q < NS; q++)
        if (owner == q)
          pipeWait(counters + 4 * (2 * DW + q), need, seen_smp[q]);
    };
    // nominal: the real wave must be done with the states in the slots about to be rewritten; real: x*_t and x*_{t+1} are out
    auto wait_peer = [&](const int hi) {
      pipeWait(peer_prog, is_nominal ? hi - rings.xnom_steps : hi, seen_peer);
    };
    auto fetch = [&](const int t, float* ub) {
      const float* slot = my_smp + (size_t)(t & smp_mask) * C * 64 + thread_idx;
#pragma unroll
      for (int i = 0; i < C; i++)
        ub[i] = slot[i * 64];
    };
    int t = 0;
    for (; t + 1 < num_timesteps; t += 2)
    {
      wait_samples(t, t + 2);
      pipeWait(cost_prog, t + 2 - rings.out_steps, seen_cost);
      wait_peer(t + 2);
      float ubuf[2 * C];
      fetch(t, &ubuf[0]);
      fetch(t + 1, &ubuf[C]);
      dyn_step(x, x_next, t, &ubuf[0]);
      dyn_step(x_next, x, t + 1, &ubuf[C]);
      pipePublishLds(my_prog, t + 2, lane);
    }
    if (t < num_timesteps)
    {
      wait_samples(t, num_timesteps);
      pipeWait(cost_prog, num_timesteps - rings.out_steps, seen_cost);
      wait_peer(num_timesteps);
      float ubuf[C];
      fetch(t, &ubuf[0]);
      dyn_step(x, x_next, t, &ubuf[0]);
      pipePublishLds(my_prog, num_timesteps, lane);
    }
  }
  else
  {
    /* -------------------------------------------------- cost waves ---------------------------------------------------- */
    // As in rolloutPipelineRepKernel: a private copy of the cost plugin pinned to VGPRs, the NC waves of a system take the
    // PAIRS of steps in turn, evaluate ahead of the relay with the status they last saw and redo the pair (wave-uniform) when
    // a rollout arrives with another one.  Two accumulators travel through the relay here (rmppi_kernels.cu:797-812):
    //   nominal: A += running cost, B += likelihood-ratio cost;  real: A += running + likelihood ratio, B += running + feedback
    // the cost class off the argument block instead of a VGPR-pinned copy (rolloutPipelineRepKernel; kernarg_view.hpp): a loss
    // wherever registers are not what is short — so only where the dynamics form asks for it (MPPI_RMPPI_COST_VIEW)
    constexpr bool COST_VIEW = MPPI_KERNARG_RELOAD && (MPPI_COST_KERNARG_VIEW || rmppi_cost_view_of<DYN_T>::value) &&
                               kernarg_viewable<COST_T>::value;
    constexpr size_t COST_OFFSET = KernargLayout<DYN_T, COST_T>::template offset<1>();
    COST_T costs_v = *costs;
    if constexpr (!COST_VIEW)
      vgprResident(costs_v);
    COST_T* costs_w = &costs_v;
    const float* my_out = out_ring + (size_t)thread_idz * rings.out_steps * F * 64;
    float* rl_a = relay_a + BX * thread_idz;
    float* rl_b = relay_b + BX * thread_idz;
    int* rl_s = relay_status + BX * thread_idz;
    float* row = sampling->sampleRow(theta_d_shared, shared_idx);
    const bool pair_store = C == 2 && sampling->rows_global_d_ != nullptr;  // block-uniform
    int seen_dyn[DW], seen_cost = 0;
#pragma unroll
    for (int w = 0; w < DW; w++)
      seen_dyn[w] = 0;
    int status_guess = 0;  // the status this wave expects its next pair to start from
    for (int t = 2 * cost_id; t < num_timesteps; t += 2 * NC)
    {
      const int hi = min(t + 2, num_timesteps);
#pragma unroll
      for (int w = 0; w < DW; w++)
        pipeWait(counters + 4 * (thread_idz * DW + w), hi, seen_dyn[w]);
      float yb[2][O], ub[2][C], fbb[2][C];
#pragma unroll
      for (int q = 0; q < 2; q++)
      {
        const int tt = min(t + q, num_timesteps - 1);
        const float* slot = my_out + (size_t)(tt & out_mask) * F * 64 + lane;
#pragma unroll
        for (int i = 0; i < O; i++)
          yb[q][i] = slot[i * 64];
#pragma unroll
        for (int i = 0; i < C; i++)
        {
          ub[q][i] = slot[(O + i) * 64];
          fbb[q][i] = slot[(O + C + i) * 64];
        }
        // the feedback-filled, clamped control replaces the sample (rmppi_kernels.cu:780-781)
        if (!(pair_store && t + 1 < num_timesteps) && t + q < num_timesteps)
        {
#pragma unroll
          for (int i = 0; i < C; i++)
            row[(t + q) * C + i] = ub[q][i];
        }
      }
      if constexpr (C == 2)
      {
        // rows in HBM start on 128-byte lines (rowStrideGlobal) and t is even: the pair's four floats are ONE aligned 16-byte
        // store per lane — a quarter of the store instructions, and lines that fill front to back
        if (pair_store && t + 1 < num_timesteps)
        {
          typedef float pair_f32x4 __attribute__((ext_vector_type(4)));
          *reinterpret_cast<pair_f32x4*>(row + t * C) = pair_f32x4{ ub[0][0], ub[0][1], ub[1][0], ub[1][1] };
        }
      }
      float da[2] = { 0.0f, 0.0f }, db[2] = { 0.0f, 0.0f };
      auto evaluate = [&](int status) {
        if constexpr (COST_VIEW)
          costs_w = kernargObject<COST_T>(kernargBase(), COST_OFFSET);
#pragma unroll
        for (int q = 0; q < 2; q++)
        {
          if (t + q < num_timesteps)
          {
            const float curr_cost = costs_w->computeRunningCost(yb[q], ub[q], t + q, theta_c_shared, &status);
            const float lr = sampling->template computeLikelihoodRatioCost<true>(ub[q], theta_d_shared, global_idx, t + q, thread_idz,
                                                                  args.lambda, args.alpha);
            if (is_nominal)
            {
              da[q] = curr_cost;
              db[q] = lr;
            }
            else
            {
              da[q] = curr_cost + lr;
              db[q] = curr_cost +
                      sampling->template computeFeedbackCost<true>(fbb[q], theta_d_shared, t + q, thread_idz, args.lambda, args.alpha);
            }
          }
        }
        return status;
      };
      int status_out = evaluate(status_guess);
      if (t > 0)
      {
        pipeWait(cost_prog, t, seen_cost);
        acc_a = rl_a[lane];
        acc_b = rl_b[lane];
        crash_status = rl_s[lane];
      }
      if (__builtin_amdgcn_ballot_w64(crash_status != status_guess) != 0ull)  // wave-uniform: a status changed in between
        status_out = evaluate(crash_status);
      acc_a += da[0];
      acc_b += db[0];
      if (t + 1 < num_timesteps)
      {
        acc_a += da[1];
        acc_b += db[1];
      }
      crash_status = status_out;
      status_guess = status_out;
      rl_a[lane] = acc_a;
      rl_b[lane] = acc_b;
      rl_s[lane] = crash_status;
      pipePublishLds(cost_prog, hi, lane);
    }
  }
  __syncthreads();

  /* ---- the two accumulators -> trajectory costs (rmppi_kernels.cu:832-865); cost wave 0 of each system publishes ---- */
  const bool writer = !is_dyn && !is_sampler && cost_id == 0;
  if (writer)
  {
    acc_a = relay_a[shared_idx];
    acc_b = relay_b[shared_idx];
    const float* slot =
        out_ring + ((size_t)thread_idz * rings.out_steps + ((num_timesteps - 1) & out_mask)) * F * 64 + lane;
#pragma unroll
    for (int i = 0; i < O; i++)
      y[i] = slot[i * 64];
  }
  const float terminal = costs->terminalCost(y, theta_c_shared);
  acc_a += terminal;
  if (!is_nominal)
    acc_b += terminal;
  acc_a /= (float)num_timesteps;
  acc_b /= (float)num_timesteps;
  if (writer)
  {
    acc_a_s[shared_idx] = acc_a;
    acc_b_s[shared_idx] = acc_b;
  }
  __syncthreads();
  float traj_cost = acc_a;
  if (is_nominal)
  {
    const float tracking = acc_b_s[BX * (1 - RMPPI_NOMINAL_IDX) + thread_idx];  // the real system's B of this rollout
    traj_cost = 0.5f * acc_a + 0.5f * fmaxf(fminf(tracking, rargs.value_function_threshold), acc_a);
    traj_cost += acc_b;
  }
  blockSoftminEpilogueCost<SAMPLING_T, C, BX, BZ, NTHREADS>(sampling, args, traj_cost, writer, valid, global_idx, shared_idx,
                                                             thread_idz, tid_x, block_idx, nrows, theta_d_shared, cost_s, w_s);
}

/* =====================================================================================================================
 * Candidate evaluation (reference: rmppi_kernels.cu:231-356, initEvalKernel) as role waves.  9 x 32 rollouts are five blocks:
 * the fused kernel is ONE wave per 16 rollouts doing draw, step and cost in turn — a chain of T x ~1100 instructions,
 * 382 us for AutoRally-NN at T = 150 whatever the machine's width.  Here a block of 64 evaluation rollouts runs like
 * rolloutPipelineRepKernel: REP dynamics waves, two sampler waves (sampleAt: sample `rollout % samples_per_candidate` at the
 * time index shifted by the candidate's stride), two cost waves evaluating ahead of their relay.  The whole shaped row of a
 * rollout (T x C floats, 64 rollouts: 77 KB at T = 150) sits in LDS — the samplers run ahead as far as they like — next to
 * the output ring.  Same plugin calls in the same order per rollout: the candidate costs are the fused kernel's bits.
 * ===================================================================================================================== */
constexpr int INIT_EVAL_PIPE_SAMPLERS = 2, INIT_EVAL_PIPE_COSTS = 2;

template <class DYN_T, class COST_T>
__host__ inline size_t initEvalPipelineSharedBytes(const DYN_T& dyn, const COST_T& cost, int num_timesteps, int ring)
{
  constexpr int C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  size_t n = calcClassSharedMemSize(&dyn, 64) + calcClassSharedMemSize(&cost, 64);
  n += sizeof(float) * (size_t)math::nearest_multiple_4(num_timesteps * C) * 64;  // shaped samples [t][c][rollout]
  n += sizeof(float) * (size_t)ring * O * 64;                                      // output ring [slot][i][rollout]
  n += sizeof(float) * 2 * 64;                                                     // relay: running cost, status
  n += sizeof(int) * 4 * (replicated_lanes<DYN_T>::value + INIT_EVAL_PIPE_SAMPLERS + 1);
  return n;
}
template <class DYN_T, class COST_T>
__host__ inline int initEvalPipelineRing(const DYN_T& dyn, const COST_T& cost, int num_timesteps, size_t max_lds)
{
  for (int ring = 32; ring >= 4; ring >>= 1)
    if (initEvalPipelineSharedBytes(dyn, cost, num_timesteps, ring) <= max_lds)
      return ring;
  return 0;
}

template <class DYN_T, class COST_T, class SAMPLING_T>
__global__ void __launch_bounds__(64 * (replicated_lanes<DYN_T>::value + INIT_EVAL_PIPE_SAMPLERS + INIT_EVAL_PIPE_COSTS))
    initEvalPipelineKernel(DYN_T dynamics_obj, COST_T costs_obj, SAMPLING_T sampling_obj, const InitEvalArgs args,
                           const int ring_steps)
{
  constexpr int BX = 64;
  constexpr int REP = replicated_lanes<DYN_T>::value;
  static_assert(REP >= 1 && 64 % REP == 0, "whole waves of replica groups");
  constexpr int DW = BX * REP / 64;
  constexpr int NS = INIT_EVAL_PIPE_SAMPLERS, NC = INIT_EVAL_PIPE_COSTS;
  constexpr int NTHREADS = 64 * (DW + NS + NC);
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

  const int tid_x = (int)__builtin_amdgcn_workitem_id_x();
  const int wave = __builtin_amdgcn_readfirstlane(tid_x >> 6);
  const int lane = tid_x & 63;
  const bool is_dyn = wave < DW;
  const bool is_sampler = !is_dyn && wave < DW + NS;
  const int helper_id = is_dyn ? 0 : (is_sampler ? wave - DW : wave - DW - NS);
  const int thread_idx = is_dyn ? wave * PER_WAVE + (lane % PER_WAVE) : lane;
  const int global_idx = BX * (int)blockIdx.x + thread_idx;
  const bool valid = global_idx < args.num_eval_rollouts;
  const int gi = valid ? global_idx : 0;
  const int candidate_idx = gi / args.samples_per_candidate;
  const int candidate_sample_idx = gi % args.samples_per_candidate;
  const int num_timesteps = args.num_timesteps;
  const int ring_mask = ring_steps - 1;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s_shared = reinterpret_cast<float*>(smem_raw);
  float* theta_c_shared = theta_s_shared + calcClassSharedMemSize(dynamics, BX) / (int)sizeof(float);
  float* samples = theta_c_shared + calcClassSharedMemSize(costs, BX) / (int)sizeof(float);  // [t][c][rollout]
  float* ring = samples + (size_t)math::nearest_multiple_4(num_timesteps * C) * 64;           // [slot][i][rollout]
  float* relay_cost = ring + (size_t)ring_steps * O * 64;
  int* relay_status = reinterpret_cast<int*>(relay_cost + 64);
  lds_counter_t counters = (lds_counter_t)(relay_status + 64);
  // counters + 4 w: steps dynamics wave w has put into the ring; + 4 (DW + s): sampler s; + 4 (DW + NS): the cost waves
  lds_counter_t cost_prog = counters + 4 * (DW + NS);

  float x[S], x_next[S], xdot[S], u[C], y[O];
  int crash_status = 0;
#pragma unroll
  for (int i = 0; i < S; i++)
  {
    x[i] = args.states_d[candidate_idx * S + i];
    x_next[i] = 0.0f;
    xdot[i] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < C; i++)
    u[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < O; i++)
    y[i] = 0.0f;
  const int stride = args.strides_d[candidate_idx];
  if (tid_x < DW + NS + 1)
    counters[4 * tid_x] = 0;
  __syncthreads();
  dynamics->initializeDynamics(x, u, y, theta_s_shared, 0.0f, args.dt);
  costs->initializeCosts(y, u, theta_c_shared, 0.0f, args.dt);
  __syncthreads();

  float running_cost = 0.0f;
  constexpr int TRIP = 2;  // steps per sampler trip and per cost-wave turn

  if (is_sampler)
  {
    lds_counter_t smp_prog = counters + 4 * (DW + helper_id);
    typename SAMPLING_T::QuadCache quad_cache;
    for (int t = TRIP * helper_id; t < num_timesteps; t += TRIP * NS)
    {
#pragma unroll
      for (int q = 0; q < TRIP; q++)
      {
        if (t + q < num_timesteps)
        {
          const int candidate_t = min(t + q + stride, num_timesteps - 1);
          sampling->sampleAt(candidate_sample_idx, candidate_t, 0, u, &quad_cache);
#pragma unroll
          for (int i = 0; i < C; i++)
            samples[((t + q) * C + i) * 64 + lane] = u[i];
        }
      }
      pipePublishLds(smp_prog, min(t + TRIP, num_timesteps), lane);
    }
  }
  else if (is_dyn)
  {
    lds_counter_t my_prog = counters + 4 * wave;
    const int rep_lane = lane / PER_WAVE;
    int seen_smp[NS], seen_cost = 0;
#pragma unroll
    for (int q = 0; q < NS; q++)
      seen_smp[q] = 0;
    auto dyn_step = [&](float* xc, float* xn, int t) {
#pragma unroll
      for (int i = 0; i < C; i++)
        u[i] = samples[(t * C + i) * 64 + thread_idx];
      dynamics->enforceConstraints(xc, u);
      dynamics->step(xc, xn, xdot, u, y, theta_s_shared, t, args.dt);
      // the clamped control goes back next to the outputs: the cost waves read both (every lane stores: stripedStore)
      float rec[O];
#pragma unroll
      for (int i = 0; i < O; i++)
        rec[i] = y[i];
      stripedStore<REP>(ring + (size_t)(t & ring_mask) * O * 64 + thread_idx, rec, rep_lane);
      float uc[C];
#pragma unroll
      for (int i = 0; i < C; i++)
        uc[i] = u[i];
      stripedStore<REP>(samples + (size_t)t * C * 64 + thread_idx, uc, rep_lane);
    };
    int t = 0;
    for (; t + 1 < num_timesteps; t += 2)
    {
      const int owner = (t / TRIP) % NS;
#pragma unroll
      for (int q = 0; q < NS; q++)
        if (owner == q)
          pipeWait(counters + 4 * (DW + q), t + 2, seen_smp[q]);
      pipeWait(cost_prog, t + 2 - ring_steps, seen_cost);
      dyn_step(x, x_next, t);
      dyn_step(x_next, x, t + 1);
      pipePublishLds(my_prog, t + 2, lane);
    }
    if (t < num_timesteps)
    {
      const int owner = (t / TRIP) % NS;
#pragma unroll
      for (int q = 0; q < NS; q++)
        if (owner == q)
          pipeWait(counters + 4 * (DW + q), num_timesteps, seen_smp[q]);
      pipeWait(cost_prog, num_timesteps - ring_steps, seen_cost);
      dyn_step(x, x_next, t);
      pipePublishLds(my_prog, num_timesteps, lane);
    }
  }
  else
  {
    COST_T costs_v = *costs;
    vgprResident(costs_v);
    COST_T* costs_w = &costs_v;
    int seen_dyn[DW], seen_cost = 0;
#pragma unroll
    for (int w = 0; w < DW; w++)
      seen_dyn[w] = 0;
    int status_guess = 0;
    for (int t = 2 * helper_id; t < num_timesteps; t += 2 * NC)
    {
      const int hi = min(t + 2, num_timesteps);
#pragma unroll
      for (int w = 0; w < DW; w++)
        pipeWait(counters + 4 * w, hi, seen_dyn[w]);
      float yb[2][O], ub[2][C];
#pragma unroll
      for (int q = 0; q < 2; q++)
      {
        const int tt = min(t + q, num_timesteps - 1);
        const float* slot = ring + (size_t)(tt & ring_mask) * O * 64 + lane;
#pragma unroll
        for (int i = 0; i < O; i++)
          yb[q][i] = slot[i * 64];
#pragma unroll
        for (int i = 0; i < C; i++)
          ub[q][i] = samples[(tt * C + i) * 64 + lane];
      }
      float cq[2] = { 0.0f, 0.0f };
      auto evaluate = [&](int status) {
#pragma unroll
        for (int q = 0; q < 2; q++)
        {
          // the likelihood-ratio term is taken at the UNSHIFTED time and the GLOBAL index (rmppi_kernels.cu:337-339)
          if (t + q < num_timesteps)
            cq[q] = costs_w->computeRunningCost(yb[q], ub[q], t + q, theta_c_shared, &status) +
                    sampling->computeLikelihoodRatioCost(ub[q], nullptr, global_idx, t + q, 0, args.lambda, args.alpha);
        }
        return status;
      };
      int status_out = evaluate(status_guess);
      if (t > 0)
      {
        pipeWait(cost_prog, t, seen_cost);
        running_cost = relay_cost[lane];
        crash_status = relay_status[lane];
      }
      if (__builtin_amdgcn_ballot_w64(crash_status != status_guess) != 0ull)
        status_out = evaluate(crash_status);
      running_cost += cq[0];
      if (t + 1 < num_timesteps)
        running_cost += cq[1];
      crash_status = status_out;
      status_guess = status_out;
      relay_cost[lane] = running_cost;
      relay_status[lane] = crash_status;
      pipePublishLds(cost_prog, hi, lane);
    }
  }
  __syncthreads();
  // computeAndSaveCost (mppi_common.cu:843-853) with running / T passed in; cost wave 0 publishes
  if (!is_dyn && !is_sampler && helper_id == 0)
  {
    running_cost = relay_cost[lane];
    const float* slot = ring + (size_t)((num_timesteps - 1) & ring_mask) * O * 64 + lane;
#pragma unroll
    for (int i = 0; i < O; i++)
      y[i] = slot[i * 64];
    if (valid)
      args.trajectory_costs_d[global_idx] =
          running_cost / (float)num_timesteps + costs->terminalCost(y, theta_c_shared) / (float)num_timesteps;
  }
}

}  // namespace kernels
}  // namespace mppi
#endif
