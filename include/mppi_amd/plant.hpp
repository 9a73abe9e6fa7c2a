/**
 * plant.hpp — BasePlant-style real-time wrapper around a controller: the CALLER of the hot path.
 *
 * Same class name, method names and semantics as the reference's BasePlant<CONTROLLER_T>
 * (include/mppi/core/base_plant.hpp), so a plant written against ACDSLab/MPPI-Generic (a ROS node, a simulator bridge)
 * derives from this one unchanged apart from the trajectory types:
 *   updateState(state, time)        base_plant.hpp:288-320   store the newest state, publish the control for `time`
 *   setSolution(...)                :271-282                 latch the optimised trajectories and their time stamp
 *   updateParameters()              :397-425                 apply parameter updates queued from other threads
 *   runControlIteration(is_alive)   :436-564                 wait for a new state; stride from ROBOT time;
 *                                                            updateImportanceSamplingControl + slideControlSequence;
 *                                                            computeControl; feedback; timing averages
 *   runControlLoop(is_alive)        :566-603                 iterate, pacing on the state time stamps
 * Pure virtual hooks as in the reference (:148-174): pubControl, pubNominalState, pubFreeEnergyStatistics, checkStatus,
 * getCurrentTime, getPoseTime.
 *
 * Differences, on purpose:
 *  - trajectories are row-major std::vector<float> [T][dim] (byte-compatible with the reference's column-major Eigen
 *    matrices, see controllers.hpp);
 *  - a non-finite solution makes computeControl throw mppi_amd::Error(MPPI_ERR_NAN) — the reference calls exit(-1)
 *    (:515-535);
 *  - feedback gains enter from the caller (setFeedbackGains, [T][S][C]); the DDP solver that produces them is host code
 *    outside the hot path.  computeFeedback() is the hook where a derived plant runs its solver;
 *  - the parameter queue holds closures instead of typed DYN/COST/CONTROLLER params (the engine's models are selected by
 *    name at run time).
 * Plain C++11 + <thread>/<mutex>; link libmppi_amd.so.
 */
#ifndef MPPI_AMD_PLANT_HPP_
#define MPPI_AMD_PLANT_HPP_

#include <atomic>
#include <chrono>
#include <cmath>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "mppi_amd/controllers.hpp"

namespace mppi_amd
{
template <class CONTROLLER_T>
class BasePlant
{
public:
  using s_array = std::vector<float>;
  using c_array = std::vector<float>;
  using s_traj = std::vector<float>;  ///< [T][S]
  using c_traj = std::vector<float>;  ///< [T][C]

  BasePlant(std::shared_ptr<CONTROLLER_T> controller, int hz, int optimization_stride)
    : controller_(std::move(controller)), hz_(hz), optimization_stride_(optimization_stride)
  {
    const int S = controller_->getStateDim(), C = controller_->getControlDim(), T = controller_->getNumTimesteps();
    init_state_.assign(S, 0.0f);
    state_.assign(S, 0.0f);
    init_u_.assign(C, 0.0f);
    u_.assign(C, 0.0f);
    state_traj_.assign((size_t)T * S, 0.0f);
    control_traj_.assign((size_t)T * C, 0.0f);
  }
  virtual ~BasePlant() = default;

  /* ---- hooks of the concrete plant ---- */
  virtual void pubControl(const c_array& u) = 0;
  virtual void pubNominalState(const s_array& s) = 0;
  virtual void pubFreeEnergyStatistics(const mppi_stats& fe_stats) = 0;
  virtual int checkStatus() = 0;
  virtual double getCurrentTime() = 0;
  virtual double getPoseTime() = 0;
  virtual double getStateTime()
  {
    return state_time_;
  }
  /** where a derived plant runs its feedback solver and calls setFeedbackGains() */
  virtual void computeFeedback(const s_array& state, const s_traj& state_traj, const c_traj& control_traj)
  {
  }

  /* ---- accessors ---- */
  s_traj getStateTraj()
  {
    std::lock_guard<std::mutex> lck(access_guard_);
    return state_traj_;
  }
  c_traj getControlTraj()
  {
    std::lock_guard<std::mutex> lck(access_guard_);
    return control_traj_;
  }
  virtual s_array getState()
  {
    std::lock_guard<std::mutex> lck(access_guard_);
    return state_;
  }
  virtual void setState(const s_array& state)
  {
    state_ = state;
  }
  virtual void setControl(const c_array& u)
  {
    u_ = u;
  }
  virtual void setDebugMode(bool mode)
  {
    debug_mode_ = mode;
  }
  void setFeedbackGains(const std::vector<float>& gains)
  {
    std::lock_guard<std::mutex> lck(access_guard_);
    feedback_gains_ = gains;
  }
  void resetStateTime()
  {
    last_used_state_update_time_ = -1;
  }
  double getAvgOptimizationTime() const
  {
    return avg_optimize_time_ms_;
  }
  double getAvgLoopTime() const
  {
    return avg_loop_time_ms_;
  }
  int getTargetOptimizationStride() const
  {
    return optimization_stride_;
  }
  int getLastOptimizationStride() const
  {
    return last_optimization_stride_;
  }
  void setTargetOptimizationStride(int v)
  {
    optimization_stride_ = v;
  }
  int getHz() const
  {
    return hz_;
  }
  void setHz(int hz)
  {
    hz_ = hz;
  }
  int getNumIter() const
  {
    return num_iter_;
  }

  /* ---- parameter updates from other threads: applied at the top of the next iteration ---- */
  void queueParameterUpdate(std::function<void(CONTROLLER_T&)> update)
  {
    std::lock_guard<std::mutex> lck(params_guard_);
    pending_updates_.push_back(std::move(update));
  }
  virtual bool updateParameters()
  {
    std::vector<std::function<void(CONTROLLER_T&)>> todo;
    {
      std::lock_guard<std::mutex> lck(params_guard_);
      todo.swap(pending_updates_);
    }
    for (auto& f : todo)
      f(*controller_);
    return !todo.empty();
  }

  virtual void setSolution(const s_traj& state_seq, const c_traj& control_seq, const std::vector<float>& output_seq,
                           double timestamp)
  {
    last_used_state_update_time_ = timestamp;
    std::lock_guard<std::mutex> lck(access_guard_);
    state_traj_ = state_seq;
    control_traj_ = control_seq;
    output_traj_ = output_seq;
    num_iter_++;
  }
  /** outputs y[T][O] along the latched solution (getTargetOutputSeq) */
  std::vector<float> getOutputTraj()
  {
    std::lock_guard<std::mutex> lck(access_guard_);
    return output_traj_;
  }

  virtual void updateState(const s_array& state, double time)
  {
    const double last = last_used_state_update_time_;
    const double time_since_last_opt = time - last;
    {
      std::lock_guard<std::mutex> lck(access_guard_);
      state_ = state;
      state_time_ = time;
    }
    if (last < 0)
      return;  // not optimised yet: nothing to publish
    const double dt = controller_->getDt();
    const int T = controller_->getNumTimesteps();
    // the reference tests time < last + dt*T; the knot above the interpolation interval must exist as well
    const bool within = time >= last && time < last + dt * T && (int)(time_since_last_opt / dt) + 1 < T;
    if (time_since_last_opt > 0 && within)
    {
      s_array target;
      c_array u;
      {
        std::lock_guard<std::mutex> lck(access_guard_);
        target = controller_->interpolateState(state_traj_, time_since_last_opt);
        u = controller_->getCurrentControl(state, time_since_last_opt, target, control_traj_, feedback_gains_);
      }
      pubControl(u);
      if (debug_mode_)
        pubNominalState(target);
    }
  }

  void runControlIteration(std::atomic<bool>* is_alive)
  {
    using clock = std::chrono::steady_clock;
    auto ms_since = [](clock::time_point t0) {
      return std::chrono::duration<double, std::milli>(clock::now() - t0).count();
    };
    const clock::time_point loop_start = clock::now();
    if (!is_alive->load())
      return;
    double state_time = getStateTime();
    const double last = last_used_state_update_time_;
    while (last == state_time && is_alive->load())
    {  // wait for a state newer than the one last optimised from
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      state_time = getStateTime();
    }
    if (!is_alive->load())
      return;
    updateParameters();
    s_array state;
    {
      std::lock_guard<std::mutex> lck(access_guard_);
      state = state_;
      state_time = state_time_;
    }
    float sum = 0.0f;
    for (float v : state)
      sum += v;
    if (!std::isfinite(sum))
      return;
    const int status = checkStatus();
    // robot time decides how far the previous solution is slid
    if (last == -1)
      last_optimization_stride_ = 0;
    else
      last_optimization_stride_ =
          std::max((int)std::lround((state_time - last) / controller_->getDt()), optimization_stride_);
    if (last_optimization_stride_ > 0 && last_optimization_stride_ < controller_->getNumTimesteps())
    {
      controller_->updateImportanceSamplingControl(state, last_optimization_stride_);
      controller_->slideControlSequence(last_optimization_stride_);
    }
    const clock::time_point opt_start = clock::now();
    controller_->computeControl(state, last_optimization_stride_);  // throws Error(MPPI_ERR_NAN) on a non-finite solution
    const mppi_stats fe_stats = controller_->getFreeEnergyStatistics();
    const c_traj control_traj = controller_->getControlSeq();
    const s_traj state_traj = controller_->getTargetStateSeq();
    const std::vector<float> output_traj = controller_->getTargetOutputSeq();
    optimization_duration_ = ms_since(opt_start);
    const clock::time_point fb_start = clock::now();
    computeFeedback(state, state_traj, control_traj);
    feedback_duration_ = ms_since(fb_start);
    setSolution(state_traj, control_traj, output_traj, state_time);
    status_ = status;
    pubFreeEnergyStatistics(fe_stats);
    const double prev = (num_iter_ - 1.0) / num_iter_;
    avg_optimize_time_ms_ = prev * avg_optimize_time_ms_ + optimization_duration_ / num_iter_;
    avg_feedback_time_ms_ = prev * avg_feedback_time_ms_ + feedback_duration_ / num_iter_;
    optimize_loop_duration_ = ms_since(loop_start);
    avg_loop_time_ms_ = prev * avg_loop_time_ms_ + optimize_loop_duration_ / num_iter_;
  }

  void runControlLoop(std::atomic<bool>* is_alive)
  {
    state_ = init_state_;
    u_ = init_u_;
    controller_->resetControls();
    while (is_alive->load())
    {
      runControlIteration(is_alive);
      const double wait_until_state_time = last_used_state_update_time_ + (1.0 / hz_) * optimization_stride_;
      const auto sleep_start = std::chrono::steady_clock::now();
      while (is_alive->load() && wait_until_state_time > getStateTime())
      {
        updateParameters();
        std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
      sleep_duration_ =
          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sleep_start).count();
      if (num_iter_ > 0)
      {
        const double prev = (num_iter_ - 1.0) / num_iter_;
        avg_sleep_time_ms_ = prev * avg_sleep_time_ms_ + sleep_duration_ / num_iter_;
      }
    }
  }

protected:
  std::shared_ptr<CONTROLLER_T> controller_;
  std::mutex access_guard_, params_guard_;
  int hz_ = 10;
  bool debug_mode_ = false;
  int optimization_stride_ = 1;
  int last_optimization_stride_ = 0;
  s_array init_state_, state_;
  c_array init_u_, u_;
  s_traj state_traj_;
  c_traj control_traj_;
  std::vector<float> output_traj_;
  std::vector<float> feedback_gains_;
  std::vector<std::function<void(CONTROLLER_T&)>> pending_updates_;
  double state_time_ = -1;
  double last_used_state_update_time_ = -1;
  // wall-clock bookkeeping, milliseconds (base_plant.hpp:102-109)
  double optimize_loop_duration_ = 0, optimization_duration_ = 0, feedback_duration_ = 0, sleep_duration_ = 0;
  double avg_loop_time_ms_ = 0, avg_optimize_time_ms_ = 0, avg_feedback_time_ms_ = 0, avg_sleep_time_ms_ = 0;
  int num_iter_ = 0;
  int status_ = 1;
};

/**
 * A plant whose robot is the engine's own model integrated in simulated time (mppi_model_step): the state estimator
 * ticks every 1/hz seconds of robot time and the published control is held over the tick — the loop of the reference's
 * examples (examples/cartpole_example.cu:63-85) expressed through the plant interface.
 */
template <class CONTROLLER_T>
class SimulatedPlant : public BasePlant<CONTROLLER_T>
{
public:
  using Base = BasePlant<CONTROLLER_T>;
  SimulatedPlant(std::shared_ptr<CONTROLLER_T> controller, int hz, int optimization_stride,
                 const std::vector<float>& init_state)
    : Base(std::move(controller), hz, optimization_stride), sim_state_(init_state)
  {
    this->init_state_ = init_state;
    current_control_ = this->init_u_;
  }
  void pubControl(const std::vector<float>& u) override
  {
    current_control_ = u;
    num_published_++;
  }
  void pubNominalState(const std::vector<float>& s) override
  {
  }
  void pubFreeEnergyStatistics(const mppi_stats& s) override
  {
    last_free_energy_ = s.real_sys.free_energy_mean;
  }
  int checkStatus() override
  {
    return 0;
  }
  double getCurrentTime() override
  {
    return sim_time_;
  }
  double getPoseTime() override
  {
    return sim_time_;
  }
  void stepSimulation()
  {
    const double tick = 1.0 / this->hz_;
    std::vector<float> u = current_control_;
    this->controller_->modelStep(sim_state_, u, (float)tick, true);
    sim_time_ += tick;
    this->updateState(sim_state_, sim_time_);
  }
  /** single-threaded closed loop: optimise whenever optimization_stride ticks of robot time have passed */
  const std::vector<float>& runSimulation(int num_ticks)
  {
    std::atomic<bool> alive(true);
    this->updateState(sim_state_, sim_time_);
    this->runControlIteration(&alive);
    for (int i = 0; i < num_ticks; i++)
    {
      stepSimulation();
      const double due = this->last_used_state_update_time_ + (1.0 / this->hz_) * this->optimization_stride_;
      if (this->getStateTime() >= due - 1e-9)
        this->runControlIteration(&alive);
    }
    return sim_state_;
  }
  int numPublished() const
  {
    return num_published_;
  }
  float lastFreeEnergy() const
  {
    return last_free_energy_;
  }

protected:
  std::vector<float> sim_state_, current_control_;
  double sim_time_ = 0.0;
  int num_published_ = 0;
  float last_free_energy_ = 0.0f;
};
}  // namespace mppi_amd

#endif
