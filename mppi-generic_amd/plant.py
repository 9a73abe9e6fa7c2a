"""plant.py — the caller of the hot path: a BasePlant-style real-time wrapper around a controller.

Mirrors the reference's `BasePlant<CONTROLLER_T>` (include/mppi/core/base_plant.hpp), same method names and semantics:
    updateState(state, time)            :288-320   store the newest state; publish the interpolated control for `time`
    setSolution(...)                    :271-282   latch the newest optimised trajectories + their time stamp
    updateParameters()                  :397-425   apply parameter updates queued from other threads
    runControlIteration(is_alive)       :436-564   wait for a new state, derive the optimisation stride from ROBOT time,
                                                   updateImportanceSamplingControl + slideControlSequence, computeControl,
                                                   NaN checks, feedback, timing averages
    runControlLoop(is_alive)            :566-603   iterate, pacing on the state time stamps
and of the controller-side helpers it calls (include/mppi/controllers/controller.cuh):
    interpolateControls / interpolateState     :363-387   linear in time between neighbouring knots
    getCurrentControl                          :329-345   u_ff + u_fb, then Dynamics::enforceConstraints
    interpolateFeedback_  (feedback.cuh:216-228)          (1-a) k(x, x*, lo) + a k(x, x*, hi),  k = K[t]^T (x - x*)

Two notions of time, as in the reference: wall clock (how long an optimisation takes; the *_duration_ / avg_* fields, in ms)
and robot time (the state estimator's stamps; they alone decide the stride).  The pure-virtual hooks of the reference are
methods to override here (pubControl, pubNominalState, pubFreeEnergyStatistics, checkStatus, getCurrentTime, getPoseTime).
The reference exits the process on a non-finite solution (:515-535); the engine reports MPPI_ERR_NAN instead and the plant
re-raises it.

`SimulatedPlant` is the concrete plant used by the examples and tests: it integrates the engine's own model
(mppi_model_step) in simulated time, the way the reference's examples drive their controllers
(examples/cartpole_example.cu:63-85, examples/double_integrator_CORL2020.cu).
"""
import threading
import time as _time

import numpy as np


def interpolateControls(rel_time, c_traj, dt):
    """controller.cuh:363-378; c_traj is [T][C]"""
    lower = int(rel_time / dt)
    alpha = (rel_time - lower * dt) / dt
    return ((1.0 - alpha) * c_traj[lower] + alpha * c_traj[lower + 1]).astype(np.float32)


def interpolateState(s_traj, rel_time, dt):
    """controller.cuh:380-387 with the default Dynamics::interpolateState (linear)"""
    lower = int(rel_time / dt)
    alpha = (rel_time - lower * dt) / dt
    return ((1.0 - alpha) * s_traj[lower] + alpha * s_traj[lower + 1]).astype(np.float32)


def interpolateFeedback(state, goal_state, rel_time, gains, dt):
    """feedback.cuh:216-228 with the DDP feedback k = K[t]^T (x - x*); gains is [T][S][C]"""
    lower = int(rel_time / dt)
    alpha = (rel_time - lower * dt) / dt
    e = np.asarray(state, np.float32) - np.asarray(goal_state, np.float32)
    return ((1.0 - alpha) * (e @ gains[lower]) + alpha * (e @ gains[lower + 1])).astype(np.float32)


class BasePlant:
    def __init__(self, controller, hz, optimization_stride):
        self.controller_ = controller
        self.hz_ = int(hz)
        self.visualization_hz_ = 5
        self.debug_mode_ = False
        self.optimization_stride_ = int(optimization_stride)
        self.last_optimization_stride_ = 0
        S, C = controller.STATE_DIM, controller.CONTROL_DIM
        self.init_state_ = np.zeros(S, np.float32)
        self.init_u_ = np.zeros(C, np.float32)
        self.state_ = np.zeros(S, np.float32)
        self.u_ = np.zeros(C, np.float32)
        self.state_time_ = -1.0
        self.last_used_state_update_time_ = -1.0
        self.output_traj_ = None  # [T][O], latched with the solution when the controller offers getTargetOutputSeq()
        self.state_traj_ = np.zeros((controller.num_timesteps, S), np.float32)
        self.control_traj_ = np.zeros((controller.num_timesteps, C), np.float32)
        self.feedback_gains_ = None  # [T][S][C] or None (feedback disabled)
        self.num_iter_ = 0
        self.status_ = 1
        # timing, milliseconds (base_plant.hpp:102-109)
        self.optimize_loop_duration_ = 0.0
        self.optimization_duration_ = 0.0
        self.feedback_duration_ = 0.0
        self.sleep_duration_ = 0.0
        self.avg_loop_time_ms_ = 0.0
        self.avg_optimize_time_ms_ = 0.0
        self.avg_feedback_time_ms_ = 0.0
        self.avg_sleep_time_ms_ = 0.0
        self.access_guard_ = threading.Lock()
        self.params_guard_ = threading.Lock()
        self._pending = {}  # "dynamics" / "cost" / "controller" -> value

    # ---- hooks of the concrete plant (pure virtual in the reference, :148-174) ----
    def pubControl(self, u):
        raise NotImplementedError

    def pubNominalState(self, s):
        pass

    def pubFreeEnergyStatistics(self, stats):
        pass

    def checkStatus(self):
        return 0

    def getCurrentTime(self):
        raise NotImplementedError

    def getPoseTime(self):
        return self.state_time_

    def getStateTime(self):
        return self.state_time_

    # ---- accessors (:186-264) ----
    def getStateTraj(self):
        return self.state_traj_

    def getControlTraj(self):
        return self.control_traj_

    def getState(self):
        with self.access_guard_:
            return self.state_.copy()

    def setState(self, state):
        self.state_ = np.asarray(state, np.float32).copy()

    def setControl(self, u):
        self.u_ = np.asarray(u, np.float32).copy()

    def setDebugMode(self, mode):
        self.debug_mode_ = bool(mode)

    def resetStateTime(self):
        self.last_used_state_update_time_ = -1.0

    def getAvgOptimizationTime(self):
        return self.avg_optimize_time_ms_

    def getTargetOptimizationStride(self):
        return self.optimization_stride_

    def getLastOptimizationStride(self):
        return self.last_optimization_stride_

    def setTargetOptimizationStride(self, v):
        self.optimization_stride_ = int(v)

    def getHz(self):
        return self.hz_

    def setHz(self, hz):
        self.hz_ = int(hz)

    def setFeedbackGains(self, gains):
        """enables the feedback term of getCurrentControl; gains[T][S][C] as produced by the caller's DDP solver"""
        with self.access_guard_:
            self.feedback_gains_ = None if gains is None else np.asarray(gains, np.float32).copy()

    # ---- parameter updates from other threads (:322-368, :397-425) ----
    def setDynamicsParams(self, p):
        with self.params_guard_:
            self._pending["dynamics"] = p

    def setCostParams(self, p):
        with self.params_guard_:
            self._pending["cost"] = p

    def setControllerParams(self, lambda_=None, num_iters=None):
        with self.params_guard_:
            self._pending["controller"] = (lambda_, num_iters)

    def hasNewDynamicsParams(self):
        return "dynamics" in self._pending

    def hasNewCostParams(self):
        return "cost" in self._pending

    def hasNewControllerParams(self):
        return "controller" in self._pending

    def updateParameters(self):
        with self.params_guard_:
            pending, self._pending = self._pending, {}
        if "cost" in pending:
            self.controller_.setCostParams(pending["cost"])
        if "dynamics" in pending:
            self.controller_.setDynamicsParams(pending["dynamics"])
        if "controller" in pending:
            lam, iters = pending["controller"]
            if lam is not None:
                self.controller_.setLambda(lam)
            if iters is not None:
                self.controller_.setNumIters(iters)
        return bool(pending)

    # ---- the solution latch and the control publisher ----
    def setSolution(self, state_seq, control_seq, timestamp, output_seq=None):
        self.last_used_state_update_time_ = timestamp
        with self.access_guard_:
            self.state_traj_ = state_seq
            self.control_traj_ = control_seq
            self.output_traj_ = output_seq
            self.num_iter_ += 1

    def getCurrentControl(self, state, rel_time, target_nominal_state):
        """controller.cuh:329-345"""
        dt = self.controller_.dt
        u = interpolateControls(rel_time, self.control_traj_, dt)
        if self.feedback_gains_ is not None:
            u = u + interpolateFeedback(state, target_nominal_state, rel_time, self.feedback_gains_, dt)
        # host-side for the base rule and lock-free: this runs on the state-callback thread while computeControl may be in
        # flight on the control-loop thread (mppi_enforce_constraints)
        return self.controller_.enforceConstraints(state, u)

    def updateState(self, state, time):
        """base_plant.hpp:288-320"""
        last = self.last_used_state_update_time_
        time_since_last_opt = time - last
        with self.access_guard_:
            self.state_ = np.asarray(state, np.float32).copy()
            self.state_time_ = time
        if last < 0:
            return  # not optimised yet: nothing to publish
        dt, T = self.controller_.dt, self.controller_.num_timesteps
        # the reference tests time < last + dt*T; the knot above the interpolation interval must exist as well
        within = time >= last and time < last + dt * T and int(time_since_last_opt / dt) + 1 < T
        if time_since_last_opt > 0 and within:
            with self.access_guard_:
                target = interpolateState(self.state_traj_, time_since_last_opt, dt)
                u = self.getCurrentControl(self.state_, time_since_last_opt, target)
            self.pubControl(u)
            if self.debug_mode_:
                self.pubNominalState(target)

    # ---- the loop ----
    def runControlIteration(self, is_alive):
        """base_plant.hpp:436-564; `is_alive` is a threading.Event (set = keep running)"""
        loop_start = _time.monotonic()
        if not is_alive.is_set():
            return
        state_time = self.getStateTime()
        last = self.last_used_state_update_time_
        while last == state_time and is_alive.is_set():  # wait for a state newer than the one last optimised from
            _time.sleep(50e-6)
            state_time = self.getStateTime()
        if not is_alive.is_set():
            return
        self.updateParameters()
        with self.access_guard_:
            state = self.state_.copy()
            state_time = self.state_time_
        if not np.isfinite(state.sum()):
            return
        if self.checkRequiresBuffer():  # base_plant.hpp:477-482: this cycle's rollouts start from the history's (h0, c0)
            with self.params_guard_ if hasattr(self, "params_guard_") else self.access_guard_:
                self.updateFromBuffer(self.getSmoothedBuffer(state_time))
        status = self.checkStatus()
        # robot time decides how far the previous solution is slid
        ctl = self.controller_
        if last == -1:
            self.last_optimization_stride_ = 0
        else:
            self.last_optimization_stride_ = max(int(round((state_time - last) / ctl.dt)), self.optimization_stride_)
        stride = self.last_optimization_stride_
        if 0 < stride < ctl.num_timesteps:
            if hasattr(ctl, "updateImportanceSamplingControl"):
                ctl.updateImportanceSamplingControl(state, stride)
            ctl.slideControlSequence(stride)
        opt_start = _time.monotonic()
        ctl.computeControl(state, stride)  # raises MPPIError(MPPI_ERR_NAN) on a non-finite solution (:515-535)
        stats = ctl.getStats()
        control_traj = ctl.getControlSeq()
        state_traj = ctl.getTargetStateSeq()
        self.optimization_duration_ = (_time.monotonic() - opt_start) * 1e3
        fb_start = _time.monotonic()
        self.computeFeedback(state, state_traj, control_traj)
        self.feedback_duration_ = (_time.monotonic() - fb_start) * 1e3
        output_traj = ctl.getTargetOutputSeq() if hasattr(ctl, "getTargetOutputSeq") else None
        self.setSolution(state_traj, control_traj, state_time, output_traj)
        self.status_ = status
        self.pubFreeEnergyStatistics(stats)
        n = float(self.num_iter_)
        prev = (n - 1.0) / n
        self.avg_optimize_time_ms_ = prev * self.avg_optimize_time_ms_ + self.optimization_duration_ / n
        self.avg_feedback_time_ms_ = prev * self.avg_feedback_time_ms_ + self.feedback_duration_ / n
        self.optimize_loop_duration_ = (_time.monotonic() - loop_start) * 1e3
        self.avg_loop_time_ms_ = prev * self.avg_loop_time_ms_ + self.optimize_loop_duration_ / n

    # ---- the history hook (base_plant.hpp:266, :477-482; Dynamics::checkRequiresBuffer / updateFromBuffer) ----
    def getSmoothedBuffer(self, time):
        return {}

    def checkRequiresBuffer(self):
        return False

    def updateFromBuffer(self, buffer):
        return False

    def computeFeedback(self, state, state_traj, control_traj):
        """hook: the reference runs its DDP solver here (controller.cuh computeFeedback); gains enter this framework from
        the caller (setFeedbackGains), so the default does nothing"""

    def runControlLoop(self, is_alive):
        """base_plant.hpp:566-603"""
        self.state_ = self.init_state_.copy()
        self.u_ = self.init_u_.copy()
        # controller_->resetControls() is an empty TODO in the reference (controller.cuh:617-620): an initial control
        # trajectory set through updateImportanceSampler before the loop starts is kept
        while is_alive.is_set():
            self.runControlIteration(is_alive)
            wait_until_state_time = self.last_used_state_update_time_ + (1.0 / self.hz_) * self.optimization_stride_
            sleep_start = _time.monotonic()
            while is_alive.is_set() and wait_until_state_time > self.getStateTime():
                self.updateParameters()
                _time.sleep(50e-6)
            self.sleep_duration_ = (_time.monotonic() - sleep_start) * 1e3
            if self.num_iter_ > 0:
                prev = (self.num_iter_ - 1.0) / self.num_iter_
                self.avg_sleep_time_ms_ = prev * self.avg_sleep_time_ms_ + self.sleep_duration_ / self.num_iter_


class BufferedPlantMixin:
    """core/buffered_plant.hpp + core/buffer.hpp of the reference (C++: mppi_amd::BufferedPlant in include/mppi_amd/plant.hpp):
    named, time-stamped signal lists; getSmoothedBuffer(t) = tau / dt + 1 linearly interpolated samples ending at t (empty
    while the history is shorter than tau); and the model-side half of Dynamics::updateFromBuffer for a model with an LSTM in
    its rollouts (racer_dubins_elevation_lstm_steering.cu:216-233: rows STEER_ANGLE * 0.2, STEER_ANGLE_RATE * 0.2,
    CAN_STEER_CMD -> LSTMLSTMHelper::initializeLSTM) as a description the plant owns: setLSTMBufferInit(helper, keys, scales).
    Mix in FRONT of a BasePlant subclass."""

    def _buffer_init(self):
        if not hasattr(self, "lists_"):
            self.lists_ = {}
            self.buffer_time_horizon_, self.buffer_tau_, self.buffer_dt_ = 2.0, 1.0, 0.02
            self.lstm_helper_ = None
            self.num_buffer_updates_ = 0
            self.last_hidden_cell_ = None

    def updateExtraValue(self, name, value, time):
        self._buffer_init()
        q = self.lists_.setdefault(name, [])
        if q and time < q[-1][0]:
            return
        q.append((float(time), float(value)))

    def getInterpState(self, time):
        self._buffer_init()
        out = {}
        for k, q in self.lists_.items():
            if q:
                ts = np.array([a for a, _ in q])
                vs = np.array([b for _, b in q])
                out[k] = np.float32(np.interp(time, ts, vs))
        return out

    def getSmoothedBuffer(self, latest_time):
        self._buffer_init()
        if not self.lists_ or any((not q) or q[-1][0] - q[0][0] < self.buffer_tau_ - 1e-9 for q in self.lists_.values()):
            return {}
        steps = int(self.buffer_tau_ / self.buffer_dt_ + 1e-9) + 1
        times = latest_time - (steps - 1 - np.arange(steps)) * self.buffer_dt_
        out = {}
        for k, q in self.lists_.items():
            ts = np.array([a for a, _ in q])
            vs = np.array([b for _, b in q])
            out[k] = np.interp(times, ts, vs).astype(np.float32)
        return out

    def cleanBuffers(self, time):
        self._buffer_init()
        for k, q in self.lists_.items():
            while len(q) > 1 and q[0][0] < time - self.buffer_time_horizon_:
                q.pop(0)

    def clearBuffers(self):
        self.lists_ = {}

    def setLSTMBufferInit(self, helper, keys, scales):
        """helper: controllers.LSTMLSTMHelper with its initialiser parameters set; keys / scales: one buffer key and factor per
        input of the initialiser LSTM"""
        self._buffer_init()
        assert len(keys) == len(scales) == helper.init_input_dim
        self.lstm_helper_, self.lstm_keys_, self.lstm_scales_ = helper, list(keys), [float(v) for v in scales]

    def checkRequiresBuffer(self):
        self._buffer_init()
        return self.lstm_helper_ is not None

    def updateFromBuffer(self, buffer):
        self._buffer_init()
        h = self.lstm_helper_
        if h is None or any(k not in buffer or len(buffer[k]) < h.init_len for k in self.lstm_keys_):
            return False  # checkIfKeysInBuffer: the model keeps its previous initial state
        rows = np.stack([np.asarray(buffer[k], np.float32) * np.float32(sc) for k, sc in zip(self.lstm_keys_, self.lstm_scales_)])
        hidden, cell = h.initializeLSTM(rows)
        self.controller_.setLSTMInitialState(hidden, cell)
        self.last_hidden_cell_ = (hidden, cell)
        self.num_buffer_updates_ += 1
        return True


class SimulatedPlant(BasePlant):
    """A plant whose robot is the engine's own model integrated in simulated time (mppi_model_step): the state estimator
    ticks every 1/hz seconds of ROBOT time, the published control is held over the tick."""

    def __init__(self, controller, hz, optimization_stride, init_state=None, init_control=None):
        super().__init__(controller, hz, optimization_stride)
        if init_state is not None:
            self.init_state_ = np.asarray(init_state, np.float32).copy()
        if init_control is not None:
            self.init_u_ = np.asarray(init_control, np.float32).copy()
        self.sim_state_ = self.init_state_.copy()
        self.sim_time_ = 0.0
        self.current_control_ = self.init_u_.copy()
        self.published_controls_ = []
        self.free_energy_history_ = []

    def pubControl(self, u):
        self.current_control_ = np.asarray(u, np.float32).copy()
        self.published_controls_.append((self.state_time_, self.current_control_.copy()))

    def pubFreeEnergyStatistics(self, stats):
        self.free_energy_history_.append(stats.real_sys.free_energy_mean)

    def getCurrentTime(self):
        return self.sim_time_

    def stepSimulation(self):
        """one estimator tick: integrate the model with the held control, then report the new state"""
        tick = 1.0 / self.hz_
        x, _ = self.controller_.modelStep(self.sim_state_, self.current_control_, dt=tick, enforce_constraints=True)
        self.sim_state_ = x
        self.sim_time_ += tick
        self.updateState(self.sim_state_, self.sim_time_)

    def runSimulation(self, num_ticks):
        """single-threaded closed loop: optimise whenever optimization_stride ticks of robot time have passed"""
        alive = threading.Event()
        alive.set()
        self.updateState(self.sim_state_, self.sim_time_)
        self.runControlIteration(alive)
        for _ in range(num_ticks):
            self.stepSimulation()
            due = self.last_used_state_update_time_ + (1.0 / self.hz_) * self.optimization_stride_
            if self.getStateTime() >= due - 1e-9:
                self.runControlIteration(alive)
        return self.sim_state_
