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
 *   checkRequiresBuffer / updateFromBuffer(getSmoothedBuffer(t))   :266, :477-482   the per-cycle hook through which a model with
 *                                   an LSTM in its rollouts gets its initial (hidden, cell) from the recent history — BufferedPlant
 *                                   below (core/buffered_plant.hpp, core/buffer.hpp) keeps that history and runs the initialiser
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
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
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
  /** time series of named values, one sample every buffer_dt over the last buffer_tau seconds, oldest first
   *  (base_plant.hpp:31 buffer_trajectory = std::map<std::string, Eigen::VectorXf>) */
  using buffer_trajectory = std::map<std::string, std::vector<float>>;
  /** base_plant.hpp:266: a plant without a buffer returns an empty one */
  virtual buffer_trajectory getSmoothedBuffer(double time)
  {
    return buffer_trajectory();
  }
  /** Dynamics::checkRequiresBuffer() of the reference's model (dynamics.cuh:366-369): does the model want its history every cycle? */
  virtual bool checkRequiresBuffer()
  {
    return false;
  }
  /** Dynamics::updateFromBuffer (dynamics.cuh:371-374; racer_dubins_elevation_lstm_steering.cu:216-233): false = keys missing */
  virtual bool updateFromBuffer(const buffer_trajectory& buffer)
  {
    return false;
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
    if (checkRequiresBuffer())
    {  // base_plant.hpp:477-482 — before the optimisation: the rollouts of this cycle start from the history's (h0, c0)
      std::lock_guard<std::mutex> lck(params_guard_);
      updateFromBuffer(getSmoothedBuffer(state_time));
    }
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
 * BufferedPlant — a plant that keeps the recent history of named signals and, every control cycle, turns it into the initial
 * recurrent state of the model's prediction LSTM.
 *
 * reference: core/buffered_plant.hpp (the plant), core/buffer.hpp (time-stamped lists, getInterpState :180-207,
 * getSmoothedBuffer :209-250: tau / dt + 1 samples ending at the newest state's time, linear interpolation, empty while the
 * history is shorter than tau, cleanBuffers :252-264), and the consumer
 * dynamics/racer_dubins/racer_dubins_elevation_lstm_steering.cu:216-233 (updateFromBuffer: rows STEER_ANGLE * 0.2,
 * STEER_ANGLE_RATE * 0.2, CAN_STEER_CMD of the initialiser's input, LSTMLSTMHelper::initializeLSTM).
 * Here the model lives behind the C ABI, so the model-side half of that hook is a description the plant owns
 * (LSTMBufferInit: which buffer keys, scaled by what, feed which initialiser network) and its effect is
 * mppi_lstm_lstm_initialize (host, as in the reference) + controller.setLSTMInitialState (mppi_set_lstm_initial_state).
 */
struct LSTMBufferInit
{
  std::vector<std::string> keys;  ///< one buffer key per input of the initialiser LSTM, in input order
  std::vector<float> scales;      ///< factor applied to that key's samples (the steering model: 0.2, 0.2, 1)
  int init_input_dim = 0, init_hidden_dim = 0;
  std::vector<int> init_output_layers;  ///< {init_hidden_dim + init_input_dim, ..., 2 * hidden_dim}
  std::vector<float> init_lstm_blob, init_output_blob;  ///< the device helpers' blob layouts (lstm_blob_from_npz "init_")
  int hidden_dim = 0;  ///< of the prediction LSTM inside the rollouts
  int init_len = 0;    ///< samples the initialiser reads (the newest init_len of the buffer)
};

template <class CONTROLLER_T>
class BufferedPlant : public BasePlant<CONTROLLER_T>
{
public:
  using Base = BasePlant<CONTROLLER_T>;
  using buffer_trajectory = typename Base::buffer_trajectory;
  BufferedPlant(std::shared_ptr<CONTROLLER_T> controller, int hz, int optimization_stride)
    : Base(std::move(controller), hz, optimization_stride)
  {
  }

  /** buffer.hpp updateExtraValue / updateControls / updateOdometry: every signal is a named, time-stamped list here */
  void updateExtraValue(const std::string& name, float value, double time)
  {
    std::lock_guard<std::mutex> lck(buffer_guard_);
    std::deque<Sample>& q = lists_[name];
    if (!q.empty() && time < q.back().time)
      return;  // insertIntoBuffer drops samples older than the newest (buffer.hpp:120-133)
    q.push_back(Sample{ time, value });
  }
  void updateValues(const std::vector<std::string>& names, const std::vector<float>& values, double time)
  {
    for (size_t i = 0; i < names.size() && i < values.size(); i++)
      updateExtraValue(names[i], values[i], time);
  }
  /** linear interpolation of every list at `time`, clamped to the list's ends */
  std::map<std::string, float> getInterpState(double time)
  {
    std::lock_guard<std::mutex> lck(buffer_guard_);
    std::map<std::string, float> out;
    for (const auto& kv : lists_)
      if (!kv.second.empty())
        out[kv.first] = interp(kv.second, time);
    return out;
  }
  buffer_trajectory getSmoothedBuffer(double latest_time) override
  {
    buffer_trajectory result;
    {
      std::lock_guard<std::mutex> lck(buffer_guard_);
      if (lists_.empty())
        return result;
      for (const auto& kv : lists_)  // not enough history yet: empty, the model keeps its previous initial state
        if (kv.second.empty() || kv.second.back().time - kv.second.front().time < buffer_tau_ - 1e-9)
          return result;
    }
    const int steps = (int)(buffer_tau_ / buffer_dt_ + 1e-9) + 1;
    for (int t = 0; t < steps; t++)
    {
      const double query = latest_time - (steps - 1 - t) * buffer_dt_;
      for (const auto& kv : getInterpState(query))
      {
        std::vector<float>& row = result[kv.first];
        row.resize(steps);
        row[t] = kv.second;
      }
    }
    return result;
  }
  void cleanBuffers(double time)
  {
    std::lock_guard<std::mutex> lck(buffer_guard_);
    for (auto& kv : lists_)
      while (kv.second.size() > 1 && kv.second.front().time < time - buffer_time_horizon_)
        kv.second.pop_front();
  }
  void clearBuffers()
  {
    std::lock_guard<std::mutex> lck(buffer_guard_);
    lists_.clear();
  }
  void setBufferParams(double time_horizon, double tau, double dt)
  {
    buffer_time_horizon_ = time_horizon;
    buffer_tau_ = tau;
    buffer_dt_ = dt;
  }

  /** the model-side description of updateFromBuffer; with it the plant "requires the buffer" */
  void setLSTMBufferInit(const LSTMBufferInit& init)
  {
    lstm_init_ = init;
    has_lstm_init_ = true;
  }
  bool checkRequiresBuffer() override
  {
    return has_lstm_init_;
  }
  bool updateFromBuffer(const buffer_trajectory& buffer) override
  {
    if (!has_lstm_init_)
      return false;
    const LSTMBufferInit& li = lstm_init_;
    size_t cols = 0;
    for (const std::string& k : li.keys)
    {  // checkIfKeysInBuffer: a missing key leaves the model as it is
      auto it = buffer.find(k);
      if (it == buffer.end() || (int)it->second.size() < li.init_len)
        return false;
      cols = it->second.size();
    }
    std::vector<float> samples(cols * li.init_input_dim);  // [cols][init_input_dim], oldest first
    for (int i = 0; i < li.init_input_dim; i++)
    {
      const std::vector<float>& row = buffer.at(li.keys[i]);
      for (size_t c = 0; c < cols; c++)
        samples[c * li.init_input_dim + i] = row[c] * li.scales[i];
    }
    std::vector<float> hc(2 * (size_t)li.hidden_dim);
    const mppi_status st = mppi_lstm_lstm_initialize(li.init_input_dim, li.init_hidden_dim, li.init_output_layers.data(),
                                                     (int)li.init_output_layers.size(), li.init_lstm_blob.data(),
                                                     li.init_output_blob.data(), li.hidden_dim, li.init_len, samples.data(),
                                                     (int)cols, hc.data());
    if (st != MPPI_OK)
      throw Error(st, "BufferedPlant::updateFromBuffer: mppi_lstm_lstm_initialize failed");
    last_hidden_.assign(hc.begin(), hc.begin() + li.hidden_dim);
    last_cell_.assign(hc.begin() + li.hidden_dim, hc.end());
    this->controller_->setLSTMInitialState(last_hidden_, last_cell_);
    num_buffer_updates_++;
    return true;
  }
  int numBufferUpdates() const
  {
    return num_buffer_updates_;
  }
  const std::vector<float>& lastHidden() const
  {
    return last_hidden_;
  }
  const std::vector<float>& lastCell() const
  {
    return last_cell_;
  }

protected:
  struct Sample
  {
    double time;
    float value;
  };
  static float interp(const std::deque<Sample>& q, double time)
  {
    if (time <= q.front().time)
      return q.front().value;
    if (time >= q.back().time)
      return q.back().value;
    size_t hi = 1;
    while (q[hi].time < time)
      hi++;
    const Sample &a = q[hi - 1], &b = q[hi];
    const double w = b.time > a.time ? (time - a.time) / (b.time - a.time) : 0.0;
    return (float)((1.0 - w) * a.value + w * b.value);
  }
  std::mutex buffer_guard_;
  std::map<std::string, std::deque<Sample>> lists_;
  double buffer_time_horizon_ = 2.0;  ///< how long values are kept           (buffered_plant.hpp:81-83)
  double buffer_tau_ = 1.0;           ///< how far back the well-sampled history reaches
  double buffer_dt_ = 0.02;           ///< spacing of the well-sampled history
  LSTMBufferInit lstm_init_;
  bool has_lstm_init_ = false;
  std::vector<float> last_hidden_, last_cell_;
  int num_buffer_updates_ = 0;
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
