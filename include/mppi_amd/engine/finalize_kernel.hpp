/**
 * finalize_kernel.hpp — post-processing of the optimised control sequence, on the device.
 *
 * Replaces the reference's host-side tail of computeControl (controllers/MPPI/mppi_controller.cu:225-231):
 *   smoothControlTrajectoryHelper  controllers/controller.cuh:557-586   5-tap Savitzky-Golay [-3,12,17,12,-3]/35 over
 *                                                                        [hist0, hist1, u_0..u_{T-1}, u_{T-1}, u_{T-1}]
 *   computeStateTrajectoryHelper   controllers/controller.cuh:643-663   Euler re-rollout of u* (T-1 steps)
 *   enforceConstraints on every column of the control                    mppi_controller.cu:227-231
 * The reference runs these through the plugins' Eigen host overloads; here the SAME device plugin code that the rollout
 * kernel calls is reused (one block per system, blockDim = (1, BY, 1)), so a model needs no host implementation and the
 * nominal trajectory is propagated with exactly the arithmetic of the rollouts.
 */
#ifndef MPPI_AMD_FINALIZE_KERNEL_HPP_
#define MPPI_AMD_FINALIZE_KERNEL_HPP_

#include <hip/hip_runtime.h>
#include "mppi_amd/plugin/managed.hpp"
#include "kernarg_view.hpp"
#include "mppi_amd/plugin/math_utils.hpp"
#include "mppi_amd/plugin/parallel_utils.hpp"
#include "rollout_kernel.hpp"
#include "merge_wave.hpp"

namespace mppi
{
namespace kernels
{
struct FinalizeArgs
{
  const float* control_in_d;  ///< [D][T][C]  (the sampler's control means after the last iteration)
  const float* history_d;     ///< [2][C] control history (row 0 older); system z uses history_d + z * history_stride
  int history_stride;         ///< 0: one history shared by the systems that smooth (Vanilla / Tube); 2*C: one each (RMPPI)
  const float* x0_d;          ///< [D][S]
  float* control_out_d;       ///< [D][T][C]
  float* state_out_d;         ///< [D][T][S]
  float* output_out_d;        ///< [D][T][O] or nullptr: the outputs along the same trajectory
                              ///< (computeOutputTrajectoryHelper, controllers/controller.cuh:643-662)
  float dt;
  int num_timesteps;
  int smooth_mask;            ///< bit z set: smooth system z
  int constrain_mask;         ///< bit z set: enforceConstraints on every column of system z's control
  int constrain_mode;         ///< 0: Dynamics::enforceConstraints (mppi_controller.cu:227-231);
                              ///< 1: ColoredMPPI — only control channel 1 is clamped to its range, no deadband
                              ///<    (controllers/ColoredMPPI/colored_mppi_controller.cu:232-237)
  /* Low-latency hand-over: the *_out_d pointers then are host memory mapped into the device, the kernel copies the merge
   * statistics next to them and raises flags the host spins on — system z (one block each): flags_d[2 z] <- seq as soon as
   * its control sequence (and the statistics) are out, i.e. BEFORE the T-step re-rollout of the state trajectory, and
   * flags_d[2 z + 1] <- seq when its trajectories are complete.  nullptr: no flags (results fetched with a copy +
   * synchronise). */
  unsigned* flags_d;
  unsigned seq;
  const float* stats_in_d;    ///< [stats_floats] merge statistics (combineKernel), or nullptr
  float* stats_out_d;
  int stats_floats;
  /** horizons whose control sequence does not fit the LDS twice (T * C beyond ~19 000): [D][(2 T + 4) * C] floats of HBM for
   *  the smoothing buffer and the smoothed sequence; nullptr: both in LDS.  (One lane per rollout and replicated-lane
   *  kernels; the LDS + barrier variant keeps its trajectories in LDS as well and stays limited.) */
  float* scratch_d;
  /* Split hand-over (round 5; Vanilla / Colored and Tube MPPI): the two halves of the pass as two launches on two streams, so that the
   * re-rollout of call N — a lone wave's chain of T steps, most of the kernel — runs BESIDE the rollouts of call N + 1 instead
   * of in front of them.  phases bit 0: the control phase (smoothing, constraints, write-out, statistics, flag 2 z); bit 1: the
   * trajectory phase (re-rollout, flag 2 z + 1).  3: both, one launch (every other caller).  1: before it raises its flag the
   * block leaves in carry_d a copy of the call's input block carry_src_d[carry_floats] (initial state, control history) with
   * the smoothed control sequence in the place of the nominal control (offset carry_mean_off) — everything the trajectory
   * phase and later device-side readers of the inputs need, none of which the host or the next call's launches write — and
   * publishes seq in carry_ready_d[z] (release, device scope; one word per system).  2: the trajectory phase alone, launched on the other stream with NO
   * stream dependency (an event between two streams cost 5-12 us here, and a marker in front of the next call's rollouts): its
   * wave sleeps until carry_ready_d[0] and carry_ready_d[z] hold seq (bounded), then reads control_in_d / x0_d, which point into that carry block;
   * smooth_mask is 0. */
  int phases = 3;
  float* carry_d = nullptr;
  const float* carry_src_d = nullptr;
  int carry_floats = 0;
  int carry_mean_off = 0;
  unsigned* carry_ready_d = nullptr;
};

/** control phase of a split pass (FinalizeArgs::phases == 1): the carry block, written and published before the hand-over flag
 *  goes up (the host may overwrite the inbox carry_src_d the moment it sees the flag).  `stride` lanes of one block share the
 *  copy (one wave in the finalize kernels, sixteen in mergeControlKernel); the barrier in front of the release store has every
 *  wave's stores out. */
__device__ inline void finalizeWriteCarry(const FinalizeArgs& a, const int z, const float* ctrl, const int TC, const int lane,
                                          const int stride)
{
  if (a.phases != 1 || !a.carry_d)
    return;
  const int m0 = a.carry_mean_off + z * TC;
  if (z == 0)
    for (int e = lane; e < a.carry_floats; e += stride)
      if (e < a.carry_mean_off || e >= a.carry_mean_off + (int)gridDim.x * TC)
        a.carry_d[e] = a.carry_src_d[e];
  for (int e = lane; e < TC; e += stride)
    a.carry_d[m0 + e] = ctrl[e];
  __syncthreads();
  if (lane == 0 && a.carry_ready_d)  // one word per system: block z of the trajectory phase waits for words 0 and z
    __hip_atomic_store(a.carry_ready_d + z, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

/** trajectory phase of a split pass (phases == 2): wait for the control phase of the same call (it may not even have started:
 *  the two launches are ordered by nothing else).  false: it did not come within ~2 s — the block leaves without its flag and
 *  the host's wait reports the failure. */
__device__ inline bool finalizeAwaitCarry(const FinalizeArgs& a, const int z)
{
  if (a.phases != 2 || !a.carry_ready_d)
    return true;
  const unsigned long long t0 = wall_clock64();  // 100 MHz
  bool ok = true;
  // block 0 of the control phase wrote the shared part (initial states, history), block z this system's control sequence
  while (__hip_atomic_load(a.carry_ready_d, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != a.seq ||
         __hip_atomic_load(a.carry_ready_d + z, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != a.seq)
  {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > 200000000ull)
    {
      ok = false;
      break;
    }
  }
  return __syncthreads_and(ok) != 0;  // the same answer in every wave of the block
}

/** floats of FinalizeArgs::scratch_d per system */
__host__ __device__ inline size_t finalizeScratchFloats(int num_timesteps, int control_dim)
{
  return (size_t)math::nearest_multiple_4((num_timesteps + 4) * control_dim) + math::nearest_multiple_4(num_timesteps * control_dim);
}

/** all stores of the block are out (barrier), then one lane publishes `seq` at system scope: the host sees the data it guards */
__device__ inline void raiseHostFlag(unsigned* flags_d, const int idx, const unsigned seq, const bool one_lane)
{
  __syncthreads();
  if (flags_d && one_lane)
  {
    __threadfence_system();
    __hip_atomic_store(flags_d + idx, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

/**
 * enforceConstraints on every column of the smoothed control (mppi_controller.cu:227-231; ColoredMPPI: channel 1 only) and
 * write-out.  Works on a COPY (`work`, the smoothing buffer that is free by now): the re-rollout that follows applies the
 * constraints to its own copy of each column, and with a deadband the rule is not idempotent.  NL lanes share the element-
 * wise loops; `column_lane` / `column_stride`: which columns this thread constrains (BY == 1: one column per lane; contract
 * variant: every y lane takes its share of each column inside Dynamics::enforceConstraints).
 */
template <class DYN_T>
__device__ inline void finalizeEmitControl(DYN_T* dynamics, const FinalizeArgs& a, const int z, const float* ctrl, float* work,
                                           float* zero_state, const int elem_lane, const int elem_stride,
                                           const int column_lane, const int column_stride)
{
  constexpr int C = DYN_T::CONTROL_DIM;
  const int T = a.num_timesteps;
  for (int e = elem_lane; e < T * C; e += elem_stride)
    work[e] = ctrl[e];
  __syncthreads();
  if ((a.constrain_mask >> z) & 1)
  {
    if (a.constrain_mode == 1)
    {
      if constexpr (C > 1)
        for (int t = elem_lane; t < T; t += elem_stride)
          work[t * C + 1] = fminf(fmaxf(work[t * C + 1], dynamics->control_rngs_[1].x), dynamics->control_rngs_[1].y);
    }
    else
    {
      for (int t = column_lane; t < T; t += column_stride)
        dynamics->enforceConstraints(zero_state, &work[t * C]);
    }
  }
  __syncthreads();
  for (int e = elem_lane; e < T * C; e += elem_stride)
    a.control_out_d[(size_t)z * T * C + e] = work[e];
  if (a.stats_in_d && a.stats_out_d)
    for (int e = elem_lane; e < a.stats_floats; e += elem_stride)
      a.stats_out_d[e] = a.stats_in_d[e];
  raiseHostFlag(a.flags_d, 2 * z + 0, a.seq, elem_lane == 0);
}

/** by > 1 (LDS + barrier contract): the state and output trajectories are collected in LDS and written out once at the end —
 *  a block barrier waits for the wave's outstanding global stores, so storing every step put a memory round trip on
 *  each of the ~5 barriers of a step */
template <class DYN_T>
__host__ inline size_t finalizeSharedBytes(const DYN_T& dyn, int num_timesteps, int by = 1, bool scratch = false)
{
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  size_t n = calcClassSharedMemSize(&dyn, 1);
  if (!scratch)
    n += sizeof(float) * finalizeScratchFloats(num_timesteps, C);
  n += sizeof(float) * (4 * math::nearest_multiple_4(S) + math::nearest_multiple_4(C) + math::nearest_multiple_4(O));
  if (by > 1)
    n += sizeof(float) * (math::nearest_multiple_4(num_timesteps * S) + math::nearest_multiple_4(num_timesteps * O));
  return n;
}

/** x extent of finalizeKernel's block: with one contract lane (BY == 1) the block is a whole wave — lane 0 carries the
 *  trajectory (a serial chain), all 64 lanes share the element-wise passes around it (history copy, 5-tap smoothing,
 *  per-column constraints, write-out), which as one lane's serial loops were ~40 % of the kernel */
__host__ __device__ constexpr int finalizeBlockX(int by)
{
  return by == 1 ? 64 : 1;
}

/** SCRATCH: smoothing buffer and control sequence in FinalizeArgs::scratch_d (HBM) instead of LDS — a template flag, not a
 *  run-time one: with both possible the pointers become generic and the step loop's LDS reads flat loads (Cartpole T = 100:
 *  29.6 -> 36.0 us for the kernel) */
template <class DYN_T, int BY, bool SCRATCH = false>
__global__ void __launch_bounds__(BY* finalizeBlockX(BY)) finalizeKernel(DYN_T dynamics_obj, const FinalizeArgs a)
{
  constexpr int LX = finalizeBlockX(BY);
  __builtin_assume(__builtin_amdgcn_workgroup_size_x() == LX);
  __builtin_assume(__builtin_amdgcn_workgroup_size_y() == BY);
  __builtin_assume(__builtin_amdgcn_workgroup_size_z() == 1);
  __builtin_assume(__builtin_amdgcn_workitem_id_x() < LX);
  __builtin_assume(__builtin_amdgcn_workitem_id_y() < BY);
  __builtin_assume(__builtin_amdgcn_workitem_id_z() == 0);
  DYN_T* dynamics = &dynamics_obj;
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  const int T = a.num_timesteps;
  const int z = (int)blockIdx.x;
  const int lx = (int)__builtin_amdgcn_workitem_id_x();
  // index / stride of the element-wise loops: the y lanes of the contract variant, the 64 x lanes of the BY == 1 variant
  const int ty = (BY == 1) ? lx : (int)__builtin_amdgcn_workitem_id_y();
  constexpr int NL = BY * LX;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s = reinterpret_cast<float*>(smem_raw);
  float* lds_next = theta_s + calcClassSharedMemSize(dynamics, 1) / (int)sizeof(float);
  float* buf = lds_next;  // [(T+4)][C]
  if constexpr (SCRATCH)
    buf = a.scratch_d + (size_t)z * finalizeScratchFloats(T, C);  // long horizons: smoothing buffer and sequence in HBM
  else
    lds_next += finalizeScratchFloats(T, C);
  float* ctrl = buf + math::nearest_multiple_4((T + 4) * C);  // [T][C]
  float* x = lds_next;
  float* xn = x + math::nearest_multiple_4(S);
  float* xdot = xn + math::nearest_multiple_4(S);
  float* zero_state = xdot + math::nearest_multiple_4(S);
  float* u = zero_state + math::nearest_multiple_4(S);
  float* y = u + math::nearest_multiple_4(C);
  float* x_traj = y + math::nearest_multiple_4(O);             // [T][S], BY > 1 only
  float* y_traj = x_traj + math::nearest_multiple_4(T * S);    // [T][O], BY > 1 only

  const float* uin = a.control_in_d + (size_t)z * T * C;
  if (!finalizeAwaitCarry(a, z))  // (block-uniform)
    return;
  for (int i = ty; i < S; i += NL)
    zero_state[i] = 0.0f;

  if ((a.smooth_mask >> z) & 1)
  {
    for (int i = ty; i < C; i += NL)
    {
      buf[0 * C + i] = a.history_d[z * a.history_stride + 0 * C + i];
      buf[1 * C + i] = a.history_d[z * a.history_stride + 1 * C + i];
      buf[(T + 2) * C + i] = uin[(T - 1) * C + i];
      buf[(T + 3) * C + i] = uin[(T - 1) * C + i];
    }
    for (int e = ty; e < T * C; e += NL)
      buf[2 * C + e] = uin[e];
    __syncthreads();
    // filter_coefficients << -3, 12, 17, 12, -3; filter_coefficients /= 35.0  (controller.cuh:564-566)
    const float c0 = (float)(-3.0 / 35.0), c1 = (float)(12.0 / 35.0), c2 = (float)(17.0 / 35.0);
    for (int e = ty; e < T * C; e += NL)
    {
      float acc = c0 * buf[e];
      acc += c1 * buf[e + C];
      acc += c2 * buf[e + 2 * C];
      acc += c1 * buf[e + 3 * C];
      acc += c0 * buf[e + 4 * C];
      ctrl[e] = acc;
    }
  }
  else
  {
    for (int e = ty; e < T * C; e += NL)
      ctrl[e] = uin[e];
  }
  __syncthreads();  // ctrl is complete
  finalizeWriteCarry(a, z, ctrl, T * C, ty, NL);
  // the constrained control sequence goes out first: the host can act on it while the trajectory below is re-rolled
  if (a.phases & 1)
    finalizeEmitControl(dynamics, a, z, ctrl, buf, zero_state, ty, NL, (BY == 1) ? lx : 0, LX);
  if (!(a.phases & 2))
    return;
  if constexpr (BY == 1)
  {
    if (lx == 0)
    {
    // One lane per rollout: the whole trajectory is one thread's serial chain, so the state lives in registers and there
    // is nothing to synchronise with (the LDS-resident variant below spent ~0.55 us per step on LDS round trips and
    // barriers; this one ~0.2 us — the kernel is most of what mppi_compute_control costs beyond the iteration itself).
    float xr[S], xnr[S], xdr[S], ur[C], yr[O];
#pragma unroll
    for (int i = 0; i < S; i++)
    {
      xr[i] = a.x0_d[(size_t)z * S + i];
      xdr[i] = 0.0f;
      xnr[i] = 0.0f;
      a.state_out_d[((size_t)z * T + 0) * S + i] = xr[i];
    }
#pragma unroll
    for (int i = 0; i < O; i++)
      yr[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < C; i++)
      ur[i] = ctrl[i];
    // computeStateTrajectoryHelper / computeOutputTrajectoryHelper (controller.cuh:643-663)
    dynamics->initializeDynamics(xr, ur, yr, theta_s, 0.0f, a.dt);
    if (a.output_out_d)
    {
#pragma unroll
      for (int i = 0; i < O; i++)
        a.output_out_d[((size_t)z * T + 0) * O + i] = yr[i];
    }
    float un[C];  // the next step's control is fetched a step ahead (it may live in HBM: FinalizeArgs::scratch_d)
#pragma unroll
    for (int i = 0; i < C; i++)
      un[i] = ctrl[i];
    for (int t = 0; t < T - 1; t++)
    {
#pragma unroll
      for (int i = 0; i < C; i++)
      {
        ur[i] = un[i];
        un[i] = ctrl[(t + 1) * C + i];
      }
      dynamics->enforceConstraints(xr, ur);
      dynamics->step(xr, xnr, xdr, ur, yr, theta_s, t, a.dt);
#pragma unroll
      for (int i = 0; i < S; i++)
      {
        a.state_out_d[((size_t)z * T + t + 1) * S + i] = xnr[i];
        xr[i] = xnr[i];
      }
      if (a.output_out_d)
      {
#pragma unroll
        for (int i = 0; i < O; i++)
          a.output_out_d[((size_t)z * T + t + 1) * O + i] = yr[i];
      }
    }
    }
  }
  else
  {
  for (int i = ty; i < S; i += BY)
  {
    x[i] = a.x0_d[(size_t)z * S + i];
    xdot[i] = 0.0f;
    x_traj[i] = x[i];
  }
  for (int i = ty; i < O; i += BY)
    y[i] = 0.0f;
  __syncthreads();
  for (int i = ty; i < C; i += BY)
    u[i] = ctrl[i];
  __syncthreads();

  // computeStateTrajectoryHelper (controller.cuh:643-663)
  dynamics->initializeDynamics(x, u, y, theta_s, 0.0f, a.dt);
  __syncthreads();
  for (int i = ty; i < O; i += BY)
    y_traj[i] = y[i];
  for (int t = 0; t < T - 1; t++)
  {
    for (int i = ty; i < C; i += BY)
      u[i] = ctrl[t * C + i];
    __syncthreads();
    dynamics->enforceConstraints(x, u);
    __syncthreads();
    dynamics->step(x, xn, xdot, u, y, theta_s, t, a.dt);
    __syncthreads();
    for (int i = ty; i < S; i += BY)
      x_traj[(t + 1) * S + i] = xn[i];
    for (int i = ty; i < O; i += BY)
      y_traj[(t + 1) * O + i] = y[i];
    float* tmp = x;
    x = xn;
    xn = tmp;
  }
  __syncthreads();
  for (int e = ty; e < T * S; e += BY)
    a.state_out_d[(size_t)z * T * S + e] = x_traj[e];
  if (a.output_out_d)
    for (int e = ty; e < T * O; e += BY)
      a.output_out_d[(size_t)z * T * O + e] = y_traj[e];
  }
  raiseHostFlag(a.flags_d, 2 * z + 1, a.seq, ty == 0 && lx == 0);
}

/**
 * The same pass for models with replicated-lane (MFMA) dynamics: one wave, laid out as the rollout kernels lay out 16
 * rollouts x REP lanes, all 16 columns carrying the SAME trajectory (an MFMA computes 16 columns whether they are needed
 * or not).  State in registers, no barrier in the step loop — the LDS + barrier network forward of the contract variant
 * made this kernel 700 us for AutoRally (T = 150), i.e. most of mppi_compute_control; the MFMA forward is the one the
 * rollouts use, so the trajectory is bit-identical to what they integrated.  Launch: grid = D, block = (64, 1, 1).
 */
template <class DYN_T>
__host__ inline size_t finalizeRepSharedBytes(const DYN_T& dyn, int num_timesteps, bool scratch = false)
{
  constexpr int C = DYN_T::CONTROL_DIM;
  constexpr int ROLLOUTS = 64 / replicated_lanes<DYN_T>::value;
  return calcClassSharedMemSize(&dyn, ROLLOUTS) + (scratch ? 0 : sizeof(float) * finalizeScratchFloats(num_timesteps, C));
}

template <class DYN_T, bool SCRATCH = false>
__global__ void __launch_bounds__(64) finalizeRepKernel(DYN_T dynamics_obj, const FinalizeArgs a)
{
  constexpr int REP = replicated_lanes<DYN_T>::value;
  static_assert(REP > 1 && 64 % REP == 0, "for replicated-lane dynamics");
  constexpr int ROLLOUTS = 64 / REP;
  __builtin_assume(__builtin_amdgcn_workgroup_size_x() == 64);
  __builtin_assume(__builtin_amdgcn_workgroup_size_y() == 1);
  __builtin_assume(__builtin_amdgcn_workgroup_size_z() == 1);
  __builtin_assume(__builtin_amdgcn_workitem_id_y() == 0);
  __builtin_assume(__builtin_amdgcn_workitem_id_z() == 0);
  DYN_T* dynamics = &dynamics_obj;
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  const int T = a.num_timesteps;
  const int z = (int)blockIdx.x;
  const int lane = (int)__builtin_amdgcn_workitem_id_x();

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s = reinterpret_cast<float*>(smem_raw);
  float* buf = theta_s + calcClassSharedMemSize(dynamics, ROLLOUTS) / (int)sizeof(float);  // [(T+4)][C]
  if constexpr (SCRATCH)
    buf = a.scratch_d + (size_t)z * finalizeScratchFloats(T, C);  // long horizons: smoothing buffer and sequence in HBM
  float* ctrl = buf + math::nearest_multiple_4((T + 4) * C);                                // [T][C]
  const float* uin = a.control_in_d + (size_t)z * T * C;
  if (!finalizeAwaitCarry(a, z))
    return;

  if ((a.smooth_mask >> z) & 1)
  {
    // smoothControlTrajectoryHelper (controller.cuh:557-586): [history(2) | u | last, last] * [-3, 12, 17, 12, -3] / 35
    if (lane < C)
    {
      buf[0 * C + lane] = a.history_d[z * a.history_stride + 0 * C + lane];
      buf[1 * C + lane] = a.history_d[z * a.history_stride + 1 * C + lane];
      buf[(T + 2) * C + lane] = uin[(T - 1) * C + lane];
      buf[(T + 3) * C + lane] = uin[(T - 1) * C + lane];
    }
    for (int e = lane; e < T * C; e += 64)
      buf[2 * C + e] = uin[e];
    __syncthreads();
    const float c0 = (float)(-3.0 / 35.0), c1 = (float)(12.0 / 35.0), c2 = (float)(17.0 / 35.0);
    for (int e = lane; e < T * C; e += 64)
    {
      float acc = c0 * buf[e];
      acc += c1 * buf[e + C];
      acc += c2 * buf[e + 2 * C];
      acc += c1 * buf[e + 3 * C];
      acc += c0 * buf[e + 4 * C];
      ctrl[e] = acc;
    }
  }
  else
  {
    for (int e = lane; e < T * C; e += 64)
      ctrl[e] = uin[e];
  }
  __syncthreads();
  finalizeWriteCarry(a, z, ctrl, T * C, lane, 64);
  // the constrained control sequence goes out first (see finalizeKernel)
  if (a.phases & 1)
  {
    float zero_state[S];
#pragma unroll
    for (int i = 0; i < S; i++)
      zero_state[i] = 0.0f;
    finalizeEmitControl(dynamics, a, z, ctrl, buf, zero_state, lane, 64, lane, 64);
  }
  if (!(a.phases & 2))
    return;

  float x[S], xn[S], xdot[S], u[C], y[O];
#pragma unroll
  for (int i = 0; i < S; i++)
  {
    x[i] = a.x0_d[(size_t)z * S + i];
    xn[i] = 0.0f;
    xdot[i] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < O; i++)
    y[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < C; i++)
    u[i] = ctrl[i];
  const bool writer = lane == 0;
  if (writer)
  {
#pragma unroll
    for (int i = 0; i < S; i++)
      a.state_out_d[((size_t)z * T + 0) * S + i] = x[i];
  }
  // computeStateTrajectoryHelper / computeOutputTrajectoryHelper (controller.cuh:643-663)
  dynamics->initializeDynamics(x, u, y, theta_s, 0.0f, a.dt);
  if (writer && a.output_out_d)
  {
#pragma unroll
    for (int i = 0; i < O; i++)
      a.output_out_d[((size_t)z * T + 0) * O + i] = y[i];
  }
  float un[C];  // the next step's control, fetched a step ahead (it may live in HBM: FinalizeArgs::scratch_d)
#pragma unroll
  for (int i = 0; i < C; i++)
    un[i] = ctrl[i];
  for (int t = 0; t < T - 1; t++)
  {
#pragma unroll
    for (int i = 0; i < C; i++)
    {
      u[i] = un[i];
      un[i] = ctrl[(t + 1) * C + i];
    }
    dynamics->enforceConstraints(x, u);
    dynamics->step(x, xn, xdot, u, y, theta_s, t, a.dt);
    if (writer)
    {
#pragma unroll
      for (int i = 0; i < S; i++)
        a.state_out_d[((size_t)z * T + t + 1) * S + i] = xn[i];
      if (a.output_out_d)
      {
#pragma unroll
        for (int i = 0; i < O; i++)
          a.output_out_d[((size_t)z * T + t + 1) * O + i] = y[i];
      }
    }
#pragma unroll
    for (int i = 0; i < S; i++)
      x[i] = xn[i];
  }
  raiseHostFlag(a.flags_d, 2 * z + 1, a.seq, lane == 0);
}


/**
 * mergeControlKernel — the control phase of a split hand-over (FinalizeArgs::phases == 1) that MERGES the last rollout launch's
 * block records itself: one launch where mppi_compute_control had two on its critical path (combineKernel, then the control
 * phase of finalizeKernel, a dependent launch boundary of ~1.5 us between them and the merge kernel's own ramp in front).
 *
 * What made the merge a grid of 26 one-wave blocks on 26 CUs (csrc/reduce_kernels.hpp: MERGE_WAVES) was the record layout: lane =
 * record with the records 416 B apart is 64 cache lines per load instruction, and one CU's address unit needed ~6 us for
 * 256 records.  The rollout kernels of one-system launches now leave a TRANSPOSED copy of their records
 * (RolloutArgs::records_t_d) for the next launch's sampler waves; read from that copy a load instruction is 1 KB of contiguous
 * memory, and one block of 16 waves pulls the 106 KB of a 256-record launch through one CU in well under a microsecond.
 *
 * Arithmetic: merge_wave.hpp, the functions combineWave calls, with the same lane <-> record assignment — every wave forms rho,
 * the scale factors and eta for itself, wave w then takes the column quads w, w + 16, ...; u*[j] = U[j] / float(eta); the last
 * wave also sums w^2 and writes the statistics.  The same bits as combineKernel + finalizeKernel (tests/test_merge_control.py).  The merged mean
 * also goes to mean_out_d (what later mppi_optimize / getter calls read), the statistics to stats_d.
 * One system (grid = 1 block), T * C a multiple of 4, at most 256 records (the conditions of the streamed merge), control
 * sequence in LDS (no FinalizeArgs::scratch_d).
 * Measured (Cartpole K = 16384, T = 100; profiles/r06_compute_control_merge_control.json): 7.2 us per launch in the kernel trace
 * where combineKernel + control phase were 4.8 + 4.8; control sequence on the host 36.0 us after the call instead of 37.9.
 * Everything else the block reads from global memory (control history, carry inputs, the statistics' sticky mark) is requested
 * in the records' round trip.
 */
struct MergeControlArgs
{
  const float* records_t_d;  ///< the transposed copy of the records (rollout_kernel.hpp: RolloutArgs::records_t_d)
  int num_records;
  float lambda;
  int num_rollouts_total;    ///< K (free-energy normalisation)
  float* mean_out_d;         ///< [T*C]
  float* stats_d;            ///< [STATS_STRIDE]; [6] is read, not written (sticky failure mark)
};

constexpr int MERGE_CONTROL_WAVES = 16;
constexpr int MERGE_CONTROL_STATS = 8;  ///< == STATS_STRIDE (csrc/reduce_kernels.hpp; static_assert there)

template <class DYN_T>
__host__ inline size_t mergeControlSharedBytes(const DYN_T& dyn, int num_timesteps)
{
  constexpr int C = DYN_T::CONTROL_DIM;
  return finalizeSharedBytes(dyn, num_timesteps, 1, false) +
         sizeof(float) * (math::nearest_multiple_4(num_timesteps * C) + MERGE_CONTROL_STATS);
}

template <class DYN_T>
__global__ void __launch_bounds__(64 * MERGE_CONTROL_WAVES) mergeControlKernel(DYN_T dynamics_obj, const FinalizeArgs a,
                                                                                 const MergeControlArgs m)
{
  constexpr int NL = 64 * MERGE_CONTROL_WAVES;
  __builtin_assume(__builtin_amdgcn_workgroup_size_x() == NL);
  __builtin_assume(__builtin_amdgcn_workgroup_size_y() == 1);
  __builtin_assume(__builtin_amdgcn_workgroup_size_z() == 1);
  __builtin_assume(__builtin_amdgcn_workitem_id_x() < NL);
  DYN_T* dynamics = &dynamics_obj;
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  static_assert(MERGE_COLS == 4 && MERGE_LANE_RECORDS == 4, "the transposed copy is laid out in column quads; 256 records");
  const int T = a.num_timesteps;
  const int TC = T * C;
  const int z = 0;
  const int tid = (int)__builtin_amdgcn_workitem_id_x();
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s = reinterpret_cast<float*>(smem_raw);
  float* buf = theta_s + calcClassSharedMemSize(dynamics, 1) / (int)sizeof(float);  // [(T+4)][C]
  float* ctrl = buf + math::nearest_multiple_4((T + 4) * C);                         // [T][C]
  float* zero_state = buf + finalizeScratchFloats(T, C) + 3 * math::nearest_multiple_4(S);  // (finalizeKernel's layout)
  float* mean_s = zero_state + math::nearest_multiple_4(S) + math::nearest_multiple_4(C) + math::nearest_multiple_4(O);
  float* st_s = mean_s + math::nearest_multiple_4(TC);

  /* ---- the merge: every load of the wave in flight before anything is waited for ---- */
  typedef float merge_f4 __attribute__((ext_vector_type(4)));
  typedef float merge_f2 __attribute__((ext_vector_type(2)));
  const int NR = m.num_records;
  const int quads = TC >> 2;
  const float* tails_t = m.records_t_d + (size_t)TC * NR;
  const float* eta2_t = m.records_t_d + (size_t)(TC + 2) * NR;
  constexpr int PRE = 2;  // quads per wave whose loads are issued up front (T * C <= 128: all of them)
  merge_f2 tailq[MERGE_LANE_RECORDS];
  float eta2q[MERGE_LANE_RECORDS];
  merge_f4 vq[PRE][MERGE_LANE_RECORDS];
#pragma unroll
  for (int i = 0; i < MERGE_LANE_RECORDS; i++)
  {
    const int b = lane + 64 * i;
    const int bb = b < NR ? b : 0;
    tailq[i] = *reinterpret_cast<const merge_f2*>(tails_t + 2 * bb);
    eta2q[i] = eta2_t[bb];
  }
#pragma unroll
  for (int p = 0; p < PRE; p++)
  {
    const int q = wave + MERGE_CONTROL_WAVES * p;
#pragma unroll
    for (int i = 0; i < MERGE_LANE_RECORDS; i++)
    {
      const int b = lane + 64 * i;
      const bool ok = b < NR && q < quads;
      vq[p][i] = *reinterpret_cast<const merge_f4*>(m.records_t_d + ((size_t)(ok ? q : 0) * NR + (ok ? b : 0)) * 4);
    }
  }
  // ... and everything else the block reads from global memory, in the same round trip: the control history the smoothing starts
  // from, the sticky failure mark of the statistics block, the call's inputs that go into the carry block (finalizeWriteCarry:
  // every float of carry_src_d outside the control sequence — initial state and history, a handful)
  const bool smooth = (a.smooth_mask >> z) & 1;
  float hist0 = 0.0f, hist1 = 0.0f;
  if (smooth && tid < C)
  {
    hist0 = a.history_d[z * a.history_stride + 0 * C + tid];
    hist1 = a.history_d[z * a.history_stride + 1 * C + tid];
  }
  const float sticky_mark = tid == NL - 64 ? m.stats_d[6] : 0.0f;
  const bool carry = a.phases == 1 && a.carry_d;
  const int carry_others = carry ? a.carry_floats - TC : 0;  // floats of the input block that are not the control sequence
  auto carry_index = [&](const int k) { return k < a.carry_mean_off ? k : k + TC; };
  float carry_v = 0.0f;
  if (tid < carry_others)
    carry_v = a.carry_src_d[carry_index(tid)];
  asm volatile("" ::: "memory");  // (issued here, not sunk to their uses behind the merge)
  for (int i = tid; i < S; i += NL)
    zero_state[i] = 0.0f;
  MergeTails mt;
  {
    float rho_b[MERGE_LANE_RECORDS], eta_b[MERGE_LANE_RECORDS], eta2_b[MERGE_LANE_RECORDS];
#pragma unroll
    for (int i = 0; i < MERGE_LANE_RECORDS; i++)
    {
      const bool ok = lane + 64 * i < NR;  // padding records: rho_b = inf, eta_b = eta2_b = 0 (combineWave's rule)
      rho_b[i] = ok ? tailq[i].x : INFINITY;
      eta_b[i] = ok ? tailq[i].y : 0.0f;
      eta2_b[i] = ok ? eta2q[i] : 0.0f;
    }
    // the statistics are the business of the LAST wave (the one with the fewest column quads): only it sums w^2
    if (wave == MERGE_CONTROL_WAVES - 1)
      mergeTails<true>(rho_b, eta_b, eta2_b, (float)(1.0 / (double)m.lambda), mt);
    else
      mergeTails<false>(rho_b, eta_b, eta2_b, (float)(1.0 / (double)m.lambda), mt);
  }
  auto merge_quad = [&](const int q, const merge_f4 (&raw)[MERGE_LANE_RECORDS]) {
    float v[MERGE_LANE_RECORDS][MERGE_COLS], tot[MERGE_COLS];
#pragma unroll
    for (int i = 0; i < MERGE_LANE_RECORDS; i++)
    {
      const bool ok = lane + 64 * i < NR;
      v[i][0] = ok ? raw[i].x : 0.0f;
      v[i][1] = ok ? raw[i].y : 0.0f;
      v[i][2] = ok ? raw[i].z : 0.0f;
      v[i][3] = ok ? raw[i].w : 0.0f;
    }
    mergeColumns(mt.s, v, tot);
    if (lane < MERGE_COLS)
    {  // lane c takes column c (the sums are wave-uniform), as in combineWave
      float mine = tot[0];
#pragma unroll
      for (int c = 1; c < MERGE_COLS; c++)
        mine = lane == c ? tot[c] : mine;
      const float mu = mine / mt.eta_f;
      mean_s[4 * q + lane] = mu;
      m.mean_out_d[4 * q + lane] = mu;
    }
  };
#pragma unroll
  for (int p = 0; p < PRE; p++)
  {
    const int q = wave + MERGE_CONTROL_WAVES * p;
    if (q < quads)  // (wave-uniform)
      merge_quad(q, vq[p]);
  }
  for (int q = wave + MERGE_CONTROL_WAVES * PRE; q < quads; q += MERGE_CONTROL_WAVES)
  {  // longer sequences: further rounds, a memory round trip each
    merge_f4 raw[MERGE_LANE_RECORDS];
#pragma unroll
    for (int i = 0; i < MERGE_LANE_RECORDS; i++)
    {
      const int b = lane + 64 * i;
      raw[i] = *reinterpret_cast<const merge_f4*>(m.records_t_d + ((size_t)q * NR + (b < NR ? b : 0)) * 4);
    }
    merge_quad(q, raw);
  }
  if (tid == NL - 64)  // lane 0 of the last wave
  {
    mergeStatistics(mt.rho, mt.eta_f, mt.eta2, m.lambda, m.num_rollouts_total, st_s);
    st_s[6] = sticky_mark;
#pragma unroll
    for (int i = 0; i < MERGE_CONTROL_STATS; i++)
      if (i != 6)
        m.stats_d[i] = st_s[i];
  }
  __syncthreads();  // mean_s, st_s complete

  /* ---- the control phase of finalizeKernel on the merged mean ---- */
  const float* uin = mean_s;
  if (smooth)
  {
    if (tid < C)
    {
      buf[0 * C + tid] = hist0;
      buf[1 * C + tid] = hist1;
      buf[(T + 2) * C + tid] = uin[(T - 1) * C + tid];
      buf[(T + 3) * C + tid] = uin[(T - 1) * C + tid];
    }
    for (int e = tid; e < TC; e += NL)
      buf[2 * C + e] = uin[e];
    __syncthreads();
    // filter_coefficients << -3, 12, 17, 12, -3; filter_coefficients /= 35.0  (controller.cuh:564-566)
    const float c0 = (float)(-3.0 / 35.0), c1 = (float)(12.0 / 35.0), c2 = (float)(17.0 / 35.0);
    for (int e = tid; e < TC; e += NL)
    {
      float acc = c0 * buf[e];
      acc += c1 * buf[e + C];
      acc += c2 * buf[e + 2 * C];
      acc += c1 * buf[e + 3 * C];
      acc += c0 * buf[e + 4 * C];
      ctrl[e] = acc;
    }
  }
  else
  {
    for (int e = tid; e < TC; e += NL)
      ctrl[e] = uin[e];
  }
  __syncthreads();  // ctrl is complete
  if (carry)
  {  // finalizeWriteCarry with the input block's floats already in registers
    if (tid < carry_others)
      a.carry_d[carry_index(tid)] = carry_v;
    for (int k = tid + NL; k < carry_others; k += NL)
      a.carry_d[carry_index(k)] = a.carry_src_d[carry_index(k)];
    for (int e = tid; e < TC; e += NL)
      a.carry_d[a.carry_mean_off + e] = ctrl[e];
    __syncthreads();
    if (tid == 0 && a.carry_ready_d)
      __hip_atomic_store(a.carry_ready_d + z, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (a.stats_out_d)
    for (int e = tid; e < MERGE_CONTROL_STATS; e += NL)
      a.stats_out_d[e] = st_s[e];
  FinalizeArgs emit = a;
  emit.stats_in_d = nullptr;  // (written above, from this block's own merge)
  finalizeEmitControl(dynamics, emit, z, ctrl, buf, zero_state, tid, NL, tid, NL);
}
}  // namespace kernels
}  // namespace mppi
#endif
