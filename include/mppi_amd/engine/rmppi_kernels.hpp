/**
 * rmppi_kernels.hpp — the two kernels of Robust MPPI for gfx950.
 *
 * Replaces (reference: include/mppi/core/rmppi_kernels.cu):
 *   initEvalKernel                      :231-356   free-energy evaluation of the candidate nominal states
 *   rolloutRMPPIKernel                  :666-866   nominal + real system rollout with tracking feedback on the real one
 *   (and their split dynamics / cost variants :30-229, :358-664 — the reference's default — whose per-rollout
 *    semantics these kernels follow)
 *   multiCostArrayReduction             :1135-1209 not needed: one lane per rollout and system
 *
 * Structure of rolloutRMPPIKernel: block = (BX rollouts, 1, 2 systems), z = 0 the NOMINAL system and z = 1 the REAL one
 * (the reference's NOMINAL_STATE_IDX = 0 convention, robust_mppi_controller.cu:637-640).  One lane per (rollout, system),
 * state in registers, like rolloutKernel with BY == 1.  The only coupling between the two systems of a rollout is the
 * tracking feedback u_real += K_t (x_real - x_nom): the nominal lane publishes its state in a double-buffered LDS slot
 * and ONE block barrier per step orders the exchange (the reference synchronises the block five times per step).
 * Both systems see the same eps (use_same_noise_for_all_distributions, gaussian.cu:376-389) and — in RMPPI — the same
 * mean, the nominal control (robust_mppi_controller.cu:655-656).
 *
 * Cost bookkeeping per rollout, exactly the reference's two accumulators (rmppi_kernels.cu:797-812, 832-860):
 *   nominal:  A = sum cost            B = sum likelihood-ratio cost
 *   real:     A = sum cost + LR       B = sum cost + feedback cost(u_fb)
 *   A += terminal (both); real: B += terminal; all / T;
 *   S_real = A_real;   S_nom = 0.5 A_nom + 0.5 max(min(B_real, value_function_threshold), A_nom) + B_nom
 * Reference quirk NOT replicated: the combined kernel's fix-up reads the nominal accumulators of rollout 0 of the block
 * for every rollout (`running_cost_shared[blockDim.x * blockDim.y * NOMINAL_STATE_IDX]` without `+ threadIdx.x`,
 * rmppi_kernels.cu:852-859, a data race); the split cost kernel (:651-660, the reference's default path) does it per
 * rollout, and so does this kernel.
 *
 * The epilogue (block softmin records per system) is shared with rolloutKernel.
 */
#ifndef MPPI_AMD_RMPPI_KERNELS_HPP_
#define MPPI_AMD_RMPPI_KERNELS_HPP_

#include "rollout_kernel.hpp"
#include "kernarg_view.hpp"

namespace mppi
{
namespace kernels
{
constexpr int RMPPI_NOMINAL_IDX = 0;

struct InitEvalArgs
{
  float dt;
  int num_timesteps;
  int num_eval_rollouts;      ///< candidates * samples_per_candidate
  int samples_per_candidate;
  float lambda, alpha;
  const int* strides_d;       ///< [candidates]
  const float* states_d;      ///< [candidates][S]
  float* trajectory_costs_d;  ///< [num_eval_rollouts]
};

template <class DYN_T, class COST_T, class SAMPLING_T>
__host__ inline size_t initEvalSharedBytes(const DYN_T& dyn, const COST_T& cost, int bx)
{
  return calcClassSharedMemSize(&dyn, bx) + calcClassSharedMemSize(&cost, bx);
}

/** lane -> (rollout of the block, replica) for dynamics with REPLICATED_LANES > 1 (MFMA forward: a wave carries 64 / REP
 *  rollouts, each REP times), identity otherwise — the mapping of rolloutKernel */
template <int REP>
__device__ inline void replicaMapping(const int tid_x, int& thread_idx, int& rep_lane)
{
  thread_idx = tid_x;
  rep_lane = 0;
  if (REP > 1)
  {
    constexpr int PER_WAVE = 64 / REP;
    const int l = tid_x & 63;
    thread_idx = (tid_x >> 6) * PER_WAVE + (l % PER_WAVE);
    rep_lane = l / PER_WAVE;
  }
}

/** reference: rmppi_kernels.cu:231-356.  block = (BX * REP, 1, 1); rollout = candidate * samples_per_candidate + sample */
template <class DYN_T, class COST_T, class SAMPLING_T, int BX>
__global__ void __launch_bounds__(BX* replicated_lanes<DYN_T>::value)
    initEvalKernel(DYN_T dynamics_obj, COST_T costs_obj, SAMPLING_T sampling_obj, const InitEvalArgs args)
{
  constexpr int REP = replicated_lanes<DYN_T>::value;
  static_assert(REP == 1 || (64 % REP == 0 && (BX * REP) % 64 == 0), "replicated lanes need whole waves");
  __builtin_assume(__builtin_amdgcn_workgroup_size_x() == BX * REP);
  __builtin_assume(__builtin_amdgcn_workgroup_size_y() == 1);
  __builtin_assume(__builtin_amdgcn_workgroup_size_z() == 1);
  __builtin_assume(__builtin_amdgcn_workitem_id_x() < BX * REP);
  __builtin_assume(__builtin_amdgcn_workitem_id_y() == 0);
  __builtin_assume(__builtin_amdgcn_workitem_id_z() == 0);
  DYN_T* dynamics = &dynamics_obj;
  COST_T* costs = &costs_obj;
  SAMPLING_T* sampling = &sampling_obj;
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  int thread_idx, rep_lane;
  replicaMapping<REP>((int)__builtin_amdgcn_workitem_id_x(), thread_idx, rep_lane);
  const int global_idx = BX * (int)blockIdx.x + thread_idx;
  const bool valid = global_idx < args.num_eval_rollouts;
  const int gi = valid ? global_idx : 0;
  const int candidate_idx = gi / args.samples_per_candidate;
  const int candidate_sample_idx = gi % args.samples_per_candidate;
  const int num_timesteps = args.num_timesteps;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s_shared = reinterpret_cast<float*>(smem_raw);
  float* theta_c_shared = theta_s_shared + calcClassSharedMemSize(dynamics, BX) / (int)sizeof(float);

  float xa[S], xb[S], xdot[S], u[C], y[O];
  int crash_status = 0;
#pragma unroll
  for (int i = 0; i < S; i++)
  {
    xa[i] = args.states_d[candidate_idx * S + i];
    xb[i] = 0.0f;
    xdot[i] = 0.0f;  // the reference leaves x_dot_shared uninitialised here; every model overwrites it in step()
  }
#pragma unroll
  for (int i = 0; i < C; i++)
    u[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < O; i++)
    y[i] = 0.0f;
  const int stride = args.strides_d[candidate_idx];
  __syncthreads();
  dynamics->initializeDynamics(xa, u, y, theta_s_shared, 0.0f, args.dt);
  costs->initializeCosts(y, u, theta_c_shared, 0.0f, args.dt);
  __syncthreads();

  float running_cost = 0.0f;
  float* x = xa;
  float* x_next = xb;
  // the rollout walks ONE row of samples: a Philox quad is drawn when the walk enters it, not once per element
  // (two-control models: one draw per two steps instead of two per step; none while candidate_t rests at T - 1)
  typename SAMPLING_T::QuadCache quad_cache;
  for (int t = 0; t < num_timesteps; t++)
  {
    const int candidate_t = min(t + stride, num_timesteps - 1);
    sampling->sampleAt(candidate_sample_idx, candidate_t, 0, u, &quad_cache);
    dynamics->enforceConstraints(x, u);
    dynamics->step(x, x_next, xdot, u, y, theta_s_shared, t, args.dt);
    // the likelihood-ratio term is taken at the UNSHIFTED time and the GLOBAL index (rmppi_kernels.cu:337-339)
    running_cost += costs->computeRunningCost(y, u, t, theta_c_shared, &crash_status) +
                    sampling->computeLikelihoodRatioCost(u, nullptr, global_idx, t, 0, args.lambda, args.alpha);
    float* tmp = x;
    x = x_next;
    x_next = tmp;
  }
  // computeAndSaveCost (mppi_common.cu:843-853) with running / T passed in
  if (valid && rep_lane == 0)
    args.trajectory_costs_d[global_idx] =
        running_cost / (float)num_timesteps + costs->terminalCost(y, theta_c_shared) / (float)num_timesteps;
}

struct RMPPIArgs
{
  RolloutArgs base;
  float value_function_threshold;
};

template <class DYN_T, class COST_T, class FB_T, class SAMPLING_T>
__host__ inline size_t rmppiSharedBytes(const DYN_T& dyn, const COST_T& cost, const FB_T& fb, const SAMPLING_T& smp, int bx)
{
  const int slots = bx * 2;
  size_t n = rolloutSharedBytes(dyn, cost, smp, bx, 1, 2);
  n += calcClassSharedMemSize(&fb, slots);
  n += sizeof(float) * 2 * math::nearest_multiple_4(bx * DYN_T::STATE_DIM);  // x_nom exchange, double-buffered
  n += sizeof(float) * 2 * math::nearest_multiple_4(slots);                   // the two cost accumulators
  return n;
}

template <class DYN_T, class COST_T, class FB_T, class SAMPLING_T, int BX, bool DRAW_IN_LOOP>
__global__ void __launch_bounds__(BX * 2 * replicated_lanes<DYN_T>::value)
    rolloutRMPPIKernel(DYN_T dynamics_obj, COST_T costs_obj, FB_T fb_obj, SAMPLING_T sampling_obj, const RMPPIArgs rargs)
{
  constexpr int BZ = 2;
  // REP > 1 (MFMA dynamics): REP wave lanes carry private copies of a rollout of one system and cooperate only inside
  // the dynamics plugin; blockDim.x = BX * REP, everything else is evaluated redundantly on the replicas
  constexpr int REP = replicated_lanes<DYN_T>::value;
  static_assert(REP == 1 || (64 % REP == 0 && (BX * REP) % 64 == 0), "replicated lanes need whole waves");
  __builtin_assume(__builtin_amdgcn_workgroup_size_x() == BX * REP);
  __builtin_assume(__builtin_amdgcn_workgroup_size_y() == 1);
  __builtin_assume(__builtin_amdgcn_workgroup_size_z() == BZ);
  __builtin_assume(__builtin_amdgcn_workitem_id_x() < BX * REP);
  __builtin_assume(__builtin_amdgcn_workitem_id_y() == 0);
  __builtin_assume(__builtin_amdgcn_workitem_id_z() < BZ);
  const RolloutArgs& args = rargs.base;
  DYN_T* dynamics = &dynamics_obj;
  COST_T* costs = &costs_obj;
  FB_T* fb_controller = &fb_obj;
  SAMPLING_T* sampling = &sampling_obj;
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  constexpr int SLOTS = BX * BZ;
  constexpr int NTHREADS = BX * REP * BZ;
  const int tid_x = (int)__builtin_amdgcn_workitem_id_x();
  int thread_idx, rep_lane;  // rollout within the block, replica
  replicaMapping<REP>(tid_x, thread_idx, rep_lane);
  const int thread_idz = (int)__builtin_amdgcn_workitem_id_z();
  const int block_idx = (int)blockIdx.x;
  const int global_idx = BX * block_idx + thread_idx;
  const int shared_idx = BX * thread_idz + thread_idx;
  const int distribution_idx = thread_idz;
  sampling->setNoiseStream(distribution_idx);  // Philox stream of this thread's draws (independent-noise option)
  const int tid_flat = tid_x + BX * REP * thread_idz;
  const bool is_nominal = thread_idz == RMPPI_NOMINAL_IDX;
  const int num_timesteps = args.num_timesteps;
  const int num_rollouts = args.num_rollouts;
  const float dt = args.dt;
  const bool valid = global_idx < num_rollouts;
  const int nrows = min(BX, num_rollouts - BX * block_idx);

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s_shared = reinterpret_cast<float*>(smem_raw);
  float* theta_c_shared = theta_s_shared + calcClassSharedMemSize(dynamics, SLOTS) / (int)sizeof(float);
  float* theta_d_lds = theta_c_shared + calcClassSharedMemSize(costs, SLOTS) / (int)sizeof(float);
  float* cost_s = theta_d_lds + calcClassSharedMemSize(sampling, SLOTS) / (int)sizeof(float);
  // the block's sample rows: in LDS, or — horizons whose rows do not fit — in the sampler's HBM buffer (see rolloutKernel)
  float* theta_d_shared = sampling->blockRows(theta_d_lds, block_idx, SLOTS);
  sampling->setStagingBase(theta_d_lds);
  float* w_s = cost_s + math::nearest_multiple_4(SLOTS);
  float* theta_fb = w_s + math::nearest_multiple_4(SLOTS);
  float* xnom_buf = theta_fb + calcClassSharedMemSize(fb_controller, SLOTS) / (int)sizeof(float);  // [2][BX][S]
  float* acc_a_s = xnom_buf + 2 * math::nearest_multiple_4(BX * S);                                   // [SLOTS]
  float* acc_b_s = acc_a_s + math::nearest_multiple_4(SLOTS);

  float xa[S], xb[S], xdot[S], u[C], y[O], fb_control[C];
  int crash_status = 0;
#pragma unroll
  for (int i = 0; i < S; i++)
  {
    xa[i] = args.init_x_d[S * thread_idz + i];
    xb[i] = 0.0f;
    xdot[i] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < C; i++)
    u[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < O; i++)
    y[i] = 0.0f;
  __syncthreads();

  if (REP > 1)
    sampling->setThreadMapping(shared_idx, BX);
  dynamics->initializeDynamics(xa, u, y, theta_s_shared, 0.0f, dt);
  sampling->initializeDistributions(y, 0.0f, dt, theta_d_shared);
  costs->initializeCosts(y, u, theta_c_shared, 0.0f, dt);
  fb_controller->initializeFeedback(xa, u, theta_fb, 0.0f, dt);
  __syncthreads();

  float acc_a = 0.0f, acc_b = 0.0f;
  // one step up to and including the dynamics; leaves the clamped control in u and the feedback term in fb_control
  auto step_core = [&](float* xc, float* xn, int t, const float* eps) {
    if (DRAW_IN_LOOP)
      sampling->shapeControlSample(global_idx, t, distribution_idx, eps, u);
    else
      sampling->readControlSample(global_idx, t, distribution_idx, u, theta_d_shared, 1, 0, y);
    // publish the nominal state of this step, then one barrier orders the exchange (double-buffered: the slot written at
    // step t is next written at step t + 2, after the barrier of step t + 1 which every reader of step t has passed)
    float* slot = xnom_buf + (t & 1) * math::nearest_multiple_4(BX * S) + thread_idx * S;
    if (is_nominal)
    {
#pragma unroll
      for (int i = 0; i < S; i++)
        slot[i] = xc[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < C; i++)
      fb_control[i] = 0.0f;
    if (!is_nominal)
    {
      float x_nom[S];
#pragma unroll
      for (int i = 0; i < S; i++)
        x_nom[i] = slot[i];
      fb_controller->k(xc, x_nom, t, theta_fb, fb_control);
    }
#pragma unroll
    for (int i = 0; i < C; i++)
      u[i] += fb_control[i];
    dynamics->enforceConstraints(xc, u);
    // the feedback-filled, clamped control replaces the sample (rmppi_kernels.cu:780-781)
    // (all replicas of a rollout write — same word, same value: no replica-divergent region in the step loop)
    sampling->writeControlSample(global_idx, t, distribution_idx, u, theta_d_shared, 1, 0, y);
    dynamics->step(xc, xn, xdot, u, y, theta_s_shared, t, dt);
  };
  // what step tt adds to the two accumulators of this lane's system (rmppi_kernels.cu:797-812)
  auto cost_terms = [&](float* ys, float* us, float* fbs, int tt, int* status, float& da, float& db) {
    const float curr_cost = costs->computeRunningCost(ys, us, tt, theta_c_shared, status);
    const float lr =
        sampling->computeLikelihoodRatioCost(us, theta_d_shared, global_idx, tt, distribution_idx, args.lambda, args.alpha);
    if (is_nominal)
    {
      da = curr_cost;
      db = lr;
    }
    else
    {
      da = curr_cost + lr;
      db = curr_cost + sampling->computeFeedbackCost(fbs, theta_d_shared, tt, distribution_idx, args.lambda, args.alpha);
    }
  };
  auto one_step = [&](float* xc, float* xn, int t, const float* eps) {
    step_core(xc, xn, t, eps);
    float da, db;
    cost_terms(y, u, fb_control, t, &crash_status, da, db);
    acc_a += da;
    acc_b += db;
  };
  constexpr int STEPS = (C % 2 == 0) ? 2 : 4;
  constexpr int QUADS = STEPS * C / 4;
  int t = 0;
  if constexpr (REP > 1 && REP % STEPS == 0)
  {
    // Replicated-lane (MFMA) dynamics: the REP replicas of a rollout would evaluate every step's cost terms REP times.
    // Instead the wave keeps the outputs, clamped controls and feedback terms of REP steps and replica r evaluates step
    // t + r; the accumulators then take the REP contributions in step order (cross-lane reads), bit for bit the sum the
    // sequential code forms.  The status word is handled as in rolloutKernel: slices run with the status at the start of
    // the group, and a group in which one of the first REP - 1 slices changes it is redone in order on every lane.
    // (The sample shaping stays per step: the feedback term makes the control depend on the state.)
    constexpr int PER_WAVE = 64 / REP;
    const int col = (tid_x & 63) % PER_WAVE;
    for (; t + REP <= num_timesteps; t += REP)
    {
      float ybuf[REP][O], ubuf[REP][C], fbuf[REP][C];
      auto keep = [&](const int r) {
#pragma unroll
        for (int i = 0; i < O; i++)
          ybuf[r][i] = y[i];
#pragma unroll
        for (int i = 0; i < C; i++)
        {
          ubuf[r][i] = u[i];
          fbuf[r][i] = fb_control[i];
        }
      };
#pragma unroll
      for (int g = 0; g < REP; g += STEPS)
      {
        float zq[4 * QUADS];
        if (DRAW_IN_LOOP)
        {
#pragma unroll
          for (int q = 0; q < QUADS; q++)
            sampling->drawQuad(global_idx, (t + g) * C / 4 + q, &zq[4 * q]);
        }
#pragma unroll
        for (int s2 = 0; s2 < STEPS; s2 += 2)
        {
          step_core(xa, xb, t + g + s2, &zq[s2 * C]);
          keep(g + s2);
          step_core(xb, xa, t + g + s2 + 1, &zq[(s2 + 1) * C]);
          keep(g + s2 + 1);
        }
      }
      float ys[O], us[C], fbs[C];
      auto pick = [&](const int r) {  // this lane's copy of step t + r (r may be per-lane)
#pragma unroll
        for (int i = 0; i < O; i++)
        {
          ys[i] = ybuf[0][i];
#pragma unroll
          for (int q = 1; q < REP; q++)
            ys[i] = (r == q) ? ybuf[q][i] : ys[i];
        }
#pragma unroll
        for (int i = 0; i < C; i++)
        {
          us[i] = ubuf[0][i];
          fbs[i] = fbuf[0][i];
#pragma unroll
          for (int q = 1; q < REP; q++)
          {
            us[i] = (r == q) ? ubuf[q][i] : us[i];
            fbs[i] = (r == q) ? fbuf[q][i] : fbs[i];
          }
        }
      };
      pick(rep_lane);
      int status = crash_status;
      float da, db;
      cost_terms(ys, us, fbs, t + rep_lane, &status, da, db);
      const bool stale = (rep_lane < REP - 1) && (status != crash_status);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(stale) != 0, 0))
      {
#pragma nounroll
        for (int r = 0; r < REP; r++)
        {
          pick(r);
          cost_terms(ys, us, fbs, t + r, &crash_status, da, db);
          acc_a += da;
          acc_b += db;
        }
      }
      else
      {
#pragma unroll
        for (int r = 0; r < REP; r++)
        {
          acc_a += __shfl(da, r * PER_WAVE + col, 64);
          acc_b += __shfl(db, r * PER_WAVE + col, 64);
        }
        crash_status = __shfl(status, (REP - 1) * PER_WAVE + col, 64);
      }
    }
  }
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
      one_step(xa, xb, t + s2, &zq[s2 * C]);
      one_step(xb, xa, t + s2 + 1, &zq[(s2 + 1) * C]);
    }
  }
  if (t < num_timesteps)
  {
    float zq[4 * QUADS];
    if (DRAW_IN_LOOP)
    {
#pragma unroll
      for (int q = 0; q < QUADS; q++)
        sampling->drawQuad(global_idx, t * C / 4 + q, &zq[4 * q]);
    }
    const int rem = num_timesteps - t;  // block-uniform, so the barriers inside one_step stay uniform
    one_step(xa, xb, t, &zq[0 * C]);
    if (STEPS > 2)
    {
      if (rem > 1)
        one_step(xb, xa, t + 1, &zq[1 * C]);
      if (rem > 2)
        one_step(xa, xb, t + 2, &zq[(STEPS > 2 ? 2 : 0) * C]);
    }
  }

  /* ---- the two accumulators -> trajectory costs (rmppi_kernels.cu:832-865) ---- */
  const float terminal = costs->terminalCost(y, theta_c_shared);
  acc_a += terminal;
  if (!is_nominal)
    acc_b += terminal;
  acc_a /= (float)num_timesteps;
  acc_b /= (float)num_timesteps;
  acc_a_s[shared_idx] = acc_a;
  acc_b_s[shared_idx] = acc_b;
  __syncthreads();
  float traj_cost = acc_a;
  if (is_nominal)
  {
    const float tracking = acc_b_s[BX * (1 - RMPPI_NOMINAL_IDX) + thread_idx];  // the real system's B of this rollout
    traj_cost = 0.5f * acc_a + 0.5f * fmaxf(fminf(tracking, rargs.value_function_threshold), acc_a);
    traj_cost += acc_b;
  }
  blockSoftminEpilogueCost<SAMPLING_T, C, BX, BZ, NTHREADS>(sampling, args, traj_cost, rep_lane == 0, valid, global_idx, shared_idx,
                                                             thread_idz, tid_flat, block_idx, nrows, theta_d_shared, cost_s,
                                                             w_s);
}

}  // namespace kernels
}  // namespace mppi
#endif
