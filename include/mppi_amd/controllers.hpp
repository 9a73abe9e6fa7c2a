/**
 * controllers.hpp — header-only C++ host classes over the C ABI (include/mppi_amd.h), named like the reference's
 * controllers so that a caller written against ACDSLab/MPPI-Generic keeps its call sites:
 *
 *   reference class (include/mppi/controllers/...)                         here (namespace mppi_amd)
 *   VanillaMPPIController   MPPI/mppi_controller.cuh                       VanillaMPPIController
 *   TubeMPPIController      Tube-MPPI/tube_mppi_controller.cuh             TubeMPPIController
 *   ColoredMPPIController   ColoredMPPI/colored_mppi_controller.cuh        ColoredMPPIController
 *   RobustMPPIController    R-MPPI/robust_mppi_controller.cuh              RobustMPPIController
 *
 * The reference's classes are templates over <DYN_T, COST_T, FB_T, MAX_TIMESTEPS, NUM_ROLLOUTS, SAMPLING_T>; the precompiled
 * engine selects the (DYN_T, COST_T, SAMPLING_T) instantiation by NAME ("cartpole", "double_integrator", "autorally_nn", "racer_dubins",
 * "bicycle_slip_lstm"; mppi_list_models()) and takes the sizes at run time.  Trajectories are row-major [T][C] / [T][S]
 * std::vector<float> — byte-compatible with the reference's column-major Eigen C x T / S x T matrices
 * (controllers/controller.cuh:96), so `Eigen::Map<control_trajectory>(u.data())` gives the reference's view.
 * Plain C++11, no HIP/CUDA headers, no Eigen: compile with any host compiler and link libmppi_amd.so.
 * Errors: every failing C call throws mppi_amd::Error carrying the status and mppi_last_error() (the reference exits
 * or throws std::runtime_error, utils/gpu_err_chk.cuh:32-40, MPPI/mppi_controller.cu:64-76).
 */
#ifndef MPPI_AMD_CONTROLLERS_HPP_
#define MPPI_AMD_CONTROLLERS_HPP_

#include <stdexcept>
#include <string>
#include <vector>

#include "mppi_amd.h"
#include "mppi_amd/model_params.h"

namespace mppi_amd
{
class Error : public std::runtime_error
{
public:
  Error(mppi_status s, const std::string& what) : std::runtime_error(what), status(s)
  {
  }
  mppi_status status;
};

/** common part of the reference's Controller<...> (controllers/controller.cuh) */
class Controller
{
public:
  Controller(const std::string& model, int controller_kind, int num_rollouts, int num_timesteps, float dt, int max_iter,
             float lambda, float alpha, unsigned long long seed = 42, int device = 0, int rank = 0, int world_size = 1,
             void* stream = nullptr)
    : num_rollouts_(num_rollouts), num_timesteps_(num_timesteps), dt_(dt)
  {
    mppi_config cfg{};
    cfg.model = model.c_str();
    cfg.controller = controller_kind;
    cfg.num_rollouts = num_rollouts;
    cfg.num_timesteps = num_timesteps;
    cfg.dt = dt;
    cfg.lambda = lambda;
    cfg.alpha = alpha;
    cfg.num_iters = max_iter;
    cfg.seed = seed;
    cfg.noise_source = MPPI_NOISE_PHILOX_FUSED;
    cfg.device = device;
    cfg.stream = stream;
    cfg.rank = rank;
    cfg.world_size = world_size;
    const mppi_status s = mppi_create(&cfg, &h_);
    if (s != MPPI_OK)
      throw Error(s, std::string("mppi_create: ") + mppi_last_error(nullptr));
    check(mppi_get_dims(h_, &state_dim_, &control_dim_, &output_dim_, &num_systems_));
  }
  virtual ~Controller()
  {
    mppi_destroy(h_);
  }
  Controller(const Controller&) = delete;
  Controller& operator=(const Controller&) = delete;

  /* ---- parameters (Dynamics::setParams, Cost::setParams, SamplingDistribution::setParams, control_rngs_) ---- */
  template <class P>
  void setDynamicsParams(const P& p)
  {
    check(mppi_set_dynamics_params(h_, &p, sizeof(P)));
  }
  template <class P>
  void setCostParams(const P& p)
  {
    check(mppi_set_cost_params(h_, &p, sizeof(P)));
  }
  /** std_dev: [C] (or [D][C]); control_cost_coeff: [C] or empty for zeros */
  void setSamplingParams(std::vector<float> std_dev, std::vector<float> control_cost_coeff = {},
                         float pure_noise_trajectories_percentage = 0.01f, float std_dev_decay = 1.0f)
  {
    if ((int)std_dev.size() == control_dim_ && num_systems_ == 2)
      std_dev.insert(std_dev.end(), std_dev.begin(), std_dev.begin() + control_dim_);
    if (control_cost_coeff.empty())
      control_cost_coeff.assign(control_dim_, 0.0f);
    mppi_gaussian_params p{ std_dev.data(), control_cost_coeff.data(), pure_noise_trajectories_percentage, std_dev_decay, 32 };
    check(mppi_set_sampler_params(h_, &p));
  }
  /** lo_hi: {lo0, hi0, lo1, hi1, ...} */
  void setControlRanges(const std::vector<float>& lo_hi)
  {
    check(mppi_set_control_ranges(h_, lo_hi.data()));
  }
  void setControlDeadbands(const std::vector<float>& deadband)
  {
    check(mppi_set_control_deadband(h_, deadband.data()));
  }
  void setModelBlob(const std::string& name, const std::vector<float>& data, const std::vector<int>& dims)
  {
    check(mppi_set_model_blob(h_, name.c_str(), data.data(), data.size(), dims.data(), (int)dims.size()));
  }
  /** LSTMHelper::setHiddenState / setCellState + copyHiddenCellToDevice; produce the values with mppi::LSTMLSTMHelper
   *  (utils/nn_helpers/lstm_lstm_helper.hpp) from the history buffer, as Dynamics::updateFromBuffer does in the reference */
  void setLSTMInitialState(const std::vector<float>& hidden, const std::vector<float>& cell)
  {
    check(mppi_set_lstm_initial_state(h_, hidden.data(), cell.data()));
  }
  /** kind: "dynamics" | "lstm" | "costmap" (the reference's .npz layouts; mppi_load_npz) */
  void loadNpz(const std::string& kind, const std::string& path, const std::string& prefix = "")
  {
    check(mppi_load_npz(h_, kind.c_str(), path.c_str(), prefix.empty() ? nullptr : prefix.c_str()));
  }
  void setLambda(float lambda, float alpha = 0.0f)
  {
    check(mppi_set_lambda_alpha(h_, lambda, alpha));
  }
  void setNumIters(int n)
  {
    check(mppi_set_num_iters(h_, n));
  }
  /** mppi_reduction_mode: MPPI_REDUCTION_REFERENCE_ORDER runs the last stage of every iteration in the reference's own
   *  summation order (bit-equal closed loops against the reference's arithmetic; costs the samples a round trip through HBM) */
  void setReductionMode(int mode)
  {
    check(mppi_set_reduction_mode(h_, mode));
  }
  /** kernel launches since construction: rollout launches, launches of the reduction stage (mppi_get_launch_counts) */
  void getLaunchCounts(unsigned long long& rollout_launches, unsigned long long& merge_launches) const
  {
    check(mppi_get_launch_counts(h_, &rollout_launches, &merge_launches));
  }
  void setSeed(unsigned long long seed)
  {
    check(mppi_set_seed(h_, seed));
  }
  void setSlideControlScale(const std::vector<float>& scale)
  {
    check(mppi_set_slide_control_scale(h_, scale.data()));
  }

  /* ---- control loop ---- */
  void updateImportanceSampler(const std::vector<float>& u)
  {
    check(mppi_set_nominal_control(h_, u.data()));
  }
  void computeControl(const std::vector<float>& state, int optimization_stride = 1)
  {
    check(mppi_compute_control(h_, state.data(), optimization_stride));
  }
  std::vector<float> getControlSeq() const
  {
    std::vector<float> u((size_t)num_timesteps_ * control_dim_);
    check(mppi_get_control_seq(h_, u.data()));
    return u;
  }
  std::vector<float> getTargetStateSeq() const
  {
    std::vector<float> x((size_t)num_timesteps_ * state_dim_);
    check(mppi_get_state_seq(h_, x.data()));
    return x;
  }
  std::vector<float> getTargetOutputSeq() const
  {
    std::vector<float> y((size_t)num_timesteps_ * output_dim_);
    check(mppi_get_output_seq(h_, y.data()));
    return y;
  }
  void slideControlSequence(int steps)
  {
    check(mppi_slide(h_, steps));
  }
  mppi_stats getFreeEnergyStatistics() const
  {
    mppi_stats s{};
    check(mppi_get_stats(h_, &s));
    return s;
  }
  float getBaselineCost() const
  {
    return getFreeEnergyStatistics().real_sys.baseline;
  }
  float getNormalizerCost() const
  {
    return getFreeEnergyStatistics().real_sys.normalizer;
  }
  std::vector<float> getSampledCostSeq() const
  {
    int kl = 0;
    check(mppi_get_local_rollouts(h_, &kl, nullptr));
    std::vector<float> c((size_t)num_systems_ * kl);
    check(mppi_get_costs(h_, c.data()));
    return c;
  }
  /** the caller's simulation step, examples/cartpole_example.cu:76-80: enforceConstraints + step on the model */
  void modelStep(std::vector<float>& x, std::vector<float>& u, float dt, bool enforce_constraints = true)
  {
    check(mppi_model_step(h_, x.data(), u.data(), dt, enforce_constraints ? 1 : 0));
  }
  /** Dynamics::enforceConstraints on one control vector; on the host for plugins that keep the base rule, and without the
   *  handle lock, so the state-callback thread may call it while computeControl runs (mppi_enforce_constraints) */
  void enforceConstraints(const std::vector<float>& state, std::vector<float>& u)
  {
    check(mppi_enforce_constraints(h_, state.empty() ? nullptr : state.data(), u.data()));
  }
  /** controllers/MPPI/mppi_controller.cu:44-143: time the fused and the role-pipelined rollout kernel, keep the faster;
   *  returns the chosen mppi_kernel_variant */
  int chooseAppropriateKernel(int num_evaluations = 10)
  {
    int v = 0;
    check(mppi_choose_kernel(h_, num_evaluations, &v, nullptr, nullptr));
    return v;
  }
  /** device-resident iterations without host round trips (the unit bench.py times) */
  void optimize(int num_iterations, bool synchronize = true)
  {
    check(mppi_optimize(h_, num_iterations, synchronize ? 1 : 0));
  }

  /* ---- what a plant needs between optimisations (controllers/controller.cuh:329-387) ---- */
  /** linear interpolation between the knots around rel_time; traj is [T][dim] */
  std::vector<float> interpolate(double rel_time, const std::vector<float>& traj, int dim) const
  {
    const int lower = (int)(rel_time / dt_);
    const double alpha = (rel_time - lower * (double)dt_) / dt_;
    std::vector<float> out(dim);
    for (int i = 0; i < dim; i++)
      out[i] = (float)((1.0 - alpha) * traj[(size_t)lower * dim + i] + alpha * traj[(size_t)(lower + 1) * dim + i]);
    return out;
  }
  std::vector<float> interpolateControls(double rel_time, const std::vector<float>& c_traj) const
  {
    return interpolate(rel_time, c_traj, control_dim_);
  }
  std::vector<float> interpolateState(const std::vector<float>& s_traj, double rel_time) const
  {
    return interpolate(rel_time, s_traj, state_dim_);
  }
  /** feedback.cuh:216-228 with k = K[t]^T (x - x*); gains is [T][S][C] (empty: feedback disabled) */
  std::vector<float> interpolateFeedback(const std::vector<float>& state, const std::vector<float>& goal_state,
                                         double rel_time, const std::vector<float>& gains) const
  {
    const int lower = (int)(rel_time / dt_);
    const double alpha = (rel_time - lower * (double)dt_) / dt_;
    std::vector<float> out(control_dim_, 0.0f);
    for (int j = 0; j < control_dim_; j++)
    {
      double lo = 0.0, hi = 0.0;
      for (int i = 0; i < state_dim_; i++)
      {
        const double e = (double)state[i] - goal_state[i];
        lo += e * gains[((size_t)lower * state_dim_ + i) * control_dim_ + j];
        hi += e * gains[((size_t)(lower + 1) * state_dim_ + i) * control_dim_ + j];
      }
      out[j] = (float)((1.0 - alpha) * lo + alpha * hi);
    }
    return out;
  }
  /** controller.cuh:329-345: u_ff + u_fb, then Dynamics::enforceConstraints */
  std::vector<float> getCurrentControl(const std::vector<float>& state, double rel_time,
                                       const std::vector<float>& target_nominal_state, const std::vector<float>& c_traj,
                                       const std::vector<float>& gains)
  {
    std::vector<float> u = interpolateControls(rel_time, c_traj);
    if (!gains.empty())
    {
      const std::vector<float> u_fb = interpolateFeedback(state, target_nominal_state, rel_time, gains);
      for (int j = 0; j < control_dim_; j++)
        u[j] += u_fb[j];
    }
    enforceConstraints(state, u);
    return u;
  }
  /** Robust MPPI overrides this; a no-op for the other controllers (controller.cuh:321-327) */
  virtual void updateImportanceSamplingControl(const std::vector<float>& state, int stride)
  {
  }
  /** controller.cuh:617-620 resetControls(): an empty TODO in the reference — the nominal control sequence set through
   *  updateImportanceSampler before the control loop starts survives it, so it does here */
  void resetControls()
  {
  }
  float getDt() const
  {
    return dt_;
  }

  int getStateDim() const
  {
    return state_dim_;
  }
  int getControlDim() const
  {
    return control_dim_;
  }
  int getNumTimesteps() const
  {
    return num_timesteps_;
  }
  mppi_handle handle() const
  {
    return h_;
  }

protected:
  void check(mppi_status s) const
  {
    if (s != MPPI_OK)
      throw Error(s, std::string(mppi_status_string(s)) + ": " + mppi_last_error(h_));
  }
  mppi_handle h_ = nullptr;
  int num_rollouts_, num_timesteps_;
  float dt_;
  int state_dim_ = 0, control_dim_ = 0, output_dim_ = 0, num_systems_ = 1;
};

/** reference: controllers/MPPI/mppi_controller.cuh */
class VanillaMPPIController : public Controller
{
public:
  VanillaMPPIController(const std::string& model, int num_rollouts, int num_timesteps, float dt, int max_iter, float lambda,
                        float alpha, unsigned long long seed = 42, int device = 0)
    : Controller(model, MPPI_CONTROLLER_VANILLA, num_rollouts, num_timesteps, dt, max_iter, lambda, alpha, seed, device)
  {
  }
};

/** reference: controllers/Tube-MPPI/tube_mppi_controller.cuh */
class TubeMPPIController : public Controller
{
public:
  TubeMPPIController(const std::string& model, int num_rollouts, int num_timesteps, float dt, int max_iter, float lambda,
                     float alpha, unsigned long long seed = 42, int device = 0)
    : Controller(model, MPPI_CONTROLLER_TUBE, num_rollouts, num_timesteps, dt, max_iter, lambda, alpha, seed, device)
  {
  }
  void setNominalThreshold(float t)
  {
    check(mppi_set_nominal_threshold(h_, t));
  }
  std::vector<float> getNominalControlSeq() const
  {
    std::vector<float> u((size_t)num_timesteps_ * control_dim_);
    check(mppi_get_nominal_control_seq(h_, u.data()));
    return u;
  }
  std::vector<float> getNominalStateSeq() const
  {
    std::vector<float> x((size_t)num_timesteps_ * state_dim_);
    check(mppi_get_nominal_state_seq(h_, x.data()));
    return x;
  }
};

/** reference: controllers/ColoredMPPI/colored_mppi_controller.cuh with ColoredNoiseDistribution */
class ColoredMPPIController : public Controller
{
public:
  ColoredMPPIController(const std::string& model, int num_rollouts, int num_timesteps, float dt, int max_iter, float lambda,
                        float alpha, unsigned long long seed = 42, int device = 0)
    : Controller(model, MPPI_CONTROLLER_COLORED, num_rollouts, num_timesteps, dt, max_iter, lambda, alpha, seed, device)
  {
  }
  void setColoredNoiseParams(const std::vector<float>& exponents, float offset_decay_rate = 0.97f, float fmin = 0.0f)
  {
    check(mppi_set_colored_noise_params(h_, exponents.data(), offset_decay_rate, fmin));
  }
};

/** reference: controllers/R-MPPI/robust_mppi_controller.cuh (systems: 0 nominal, 1 real; DDP gains supplied by the caller) */
class RobustMPPIController : public Controller
{
public:
  RobustMPPIController(const std::string& model, int num_rollouts, int num_timesteps, float dt, int max_iter, float lambda,
                       float alpha, float value_function_threshold, int num_candidate_nominal_states = 9,
                       int samples_per_candidate = 32, unsigned long long seed = 42, int device = 0)
    : Controller(model, MPPI_CONTROLLER_ROBUST, num_rollouts, num_timesteps, dt, max_iter, lambda, alpha, seed, device)
  {
    check(mppi_set_rmppi_params(h_, value_function_threshold, num_candidate_nominal_states, samples_per_candidate));
    num_candidates_ = num_candidate_nominal_states;
  }
  /** gains[T][S][C] == DDPFeedbackState::fb_gain_traj_ (feedback_controllers/DDP/ddp.cuh:18-60) */
  void setFeedbackGains(const std::vector<float>& gains, bool accumulate_all_states = false)
  {
    check(mppi_set_feedback_gains(h_, gains.data(), accumulate_all_states ? 1 : 0));
  }
  void updateImportanceSamplingControl(const std::vector<float>& state, int stride) override
  {
    check(mppi_update_importance_sampling_control(h_, state.data(), stride));
  }
  std::vector<float> getNominalState() const
  {
    std::vector<float> x(state_dim_);
    check(mppi_get_rmppi_state(h_, x.data(), nullptr, nullptr, nullptr));
    return x;
  }
  int getBestIndex() const
  {
    int b = 0;
    check(mppi_get_rmppi_state(h_, nullptr, &b, nullptr, nullptr));
    return b;
  }
  std::vector<float> getCandidateFreeEnergy() const
  {
    std::vector<float> fe(num_candidates_);
    check(mppi_get_rmppi_state(h_, nullptr, nullptr, nullptr, fe.data()));
    return fe;
  }
  std::vector<float> getNominalControlSeq() const
  {
    std::vector<float> u((size_t)num_timesteps_ * control_dim_);
    check(mppi_get_nominal_control_seq(h_, u.data()));
    return u;
  }

private:
  int num_candidates_ = 9;
};

}  // namespace mppi_amd
#endif
