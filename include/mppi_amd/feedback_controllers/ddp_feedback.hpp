/**
 * ddp_feedback.hpp — device side of the DDP tracking controller used by Robust MPPI: u_fb = K_t (x - x*).
 *
 * reference: include/mppi/feedback_controllers/DDP/ddp.cuh:18-60 (DDPFeedbackState: fb_gain_traj_[T][S][C], i.e. the
 * C x S Eigen gain matrix of every timestep, column-major) and ddp.cu:11-45 (DeviceDDPImpl::k).
 * The gain PRODUCER (the host-side DDP / iLQR solver, include/mppi/ddp/) is outside the hot path (SURVEY.md §8f item
 * 3): gains arrive through the C ABI (mppi_set_feedback_gains) and live in one device buffer owned by the engine.
 *
 * Reference quirk, kept by default: for an even CONTROL_DIM ddp.cu:27-37 ASSIGNS `control_output = gain_row * e` inside
 * the loop over the states instead of accumulating, so only the LAST state's gain row survives
 * (u_fb[j] = K[S-1][j] * (x[S-1] - x*[S-1])); for an odd CONTROL_DIM it accumulates over all states (:39-43).
 * accumulate_all_states_ = true gives the mathematically intended sum over all states for every CONTROL_DIM.
 */
#ifndef MPPI_AMD_DDP_FEEDBACK_HPP_
#define MPPI_AMD_DDP_FEEDBACK_HPP_

#include <hip/hip_runtime.h>
#include "mppi_amd/plugin/managed.hpp"

namespace mppi
{
template <class DYN_T>
class DeviceDDP : public Managed
{
public:
  static const int STATE_DIM = DYN_T::STATE_DIM;
  static const int CONTROL_DIM = DYN_T::CONTROL_DIM;

  const float* fb_gain_traj_d_ = nullptr;  ///< [T][S][C] (device pointer owned by the engine)
  int num_timesteps_ = 0;
  bool accumulate_all_states_ = false;     ///< false: the reference's behaviour (see the header comment)

  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return 0;
  }
  __host__ __device__ int getBlkSharedSizeBytes() const
  {
    return 0;
  }
  /** reference: feedback_controllers/feedback.cuh initializeFeedback — nothing to set up for DDP gains */
  __device__ inline void initializeFeedback(const float* x, const float* u, float* theta_fb, const float t, const float dt)
  {
  }

  /** true when k() depends on the LAST state only (the reference's behaviour for an even CONTROL_DIM, see above): kernels
   *  that hand the nominal state from wave to wave then move one float per step instead of STATE_DIM */
  __device__ inline bool lastStateOnly() const
  {
    return (CONTROL_DIM % 2 == 0) && !accumulate_all_states_;
  }
  /** k() for lastStateOnly(): the assignment of the last state's term is the one that survives ddp.cu:27-37 */
  __device__ inline void kLastState(const float x_act_last, const float x_goal_last, const int t,
                                    float* __restrict__ control_output) const
  {
    const float* fb_gain_t = fb_gain_traj_d_ + (size_t)STATE_DIM * CONTROL_DIM * t + (STATE_DIM - 1) * CONTROL_DIM;
    const float e = x_act_last - x_goal_last;
#pragma unroll
    for (int j = 0; j < CONTROL_DIM; j++)
      control_output[j] = fb_gain_t[j] * e;
  }

  /** reference: ddp.cu:11-45.  control_output must be zero-initialised by the caller (rmppi_kernels.cu:755-758). */
  __device__ inline void k(const float* __restrict__ x_act, const float* __restrict__ x_goal, const int t,
                           float* __restrict__ theta, float* __restrict__ control_output) const
  {
    const float* fb_gain_t = fb_gain_traj_d_ + (size_t)STATE_DIM * CONTROL_DIM * t;
    const bool assign = (CONTROL_DIM % 2 == 0) && !accumulate_all_states_;
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
    {
      const float e = x_act[i] - x_goal[i];
#pragma unroll
      for (int j = 0; j < CONTROL_DIM; j++)
      {
        const float term = fb_gain_t[i * CONTROL_DIM + j] * e;
        control_output[j] = assign ? term : control_output[j] + term;
      }
    }
  }
};
}  // namespace mppi

#include <vector>

/**
 * Host-side holder with the reference's name and constructor (feedback_controllers/DDP/ddp.cuh:62-140:
 * DDPFeedback<DYN_T, NUM_TIMESTEPS>(model, dt)), so that a caller's `new DDPFeedback<Dyn, T>(model, dt)` and the FB_T
 * template argument of the controller classes keep compiling.  The reference's class also OWNS the DDP / iLQR solver that
 * produces the gains (ddp/ddp.h, host Eigen code, out of scope: SURVEY.md §2); here the gains are handed in —
 * setFeedbackGains(K[T][S][C]) — and the controller uploads them (mppi_set_feedback_gains) before its next computeControl.
 */
template <class DYN_T, int NUM_TIMESTEPS>
class DDPFeedback
{
public:
  static const int FB_TIMESTEPS = NUM_TIMESTEPS;
  typedef DYN_T DYN_TYPE;
  DDPFeedback(DYN_T* model, float dt) : model_(model), dt_(dt)
  {
  }
  /** gains [T][S][C]: the C x S gain matrix of every time step, column-major (DDPFeedbackState::fb_gain_traj_) */
  void setFeedbackGains(const std::vector<float>& gains, bool accumulate_all_states = false)
  {
    gains_ = gains;
    accumulate_all_states_ = accumulate_all_states;
    version_++;
  }
  const std::vector<float>& getFeedbackGains() const
  {
    return gains_;
  }
  bool accumulateAllStates() const
  {
    return accumulate_all_states_;
  }
  unsigned version() const
  {
    return version_;
  }
  float getDt() const
  {
    return dt_;
  }
  DYN_T* model_ = nullptr;

private:
  float dt_ = 0.0f;
  std::vector<float> gains_;
  bool accumulate_all_states_ = false;
  unsigned version_ = 0;
};
#endif
