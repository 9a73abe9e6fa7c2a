"""BasePlant-style wrapper (mppi-generic_amd/plant.py; reference: include/mppi/core/base_plant.hpp, tested upstream by
tests/plant/base_plant_test.cu with a mock controller).  The CPU tests use a mock controller too — the plant is host
logic; the GPU test closes the loop on the real engine."""
import threading
import types

import numpy as np
import pytest

import mppi_generic_amd as m
from common import cartpole_cfg, make_engine


class MockController:
    """records what the plant asks for; same surface the plant uses on a real controller"""
    STATE_DIM, CONTROL_DIM = 2, 1

    def __init__(self, T=10, dt=0.1):
        self.num_timesteps, self.dt = T, dt
        self.calls = []
        self.lam = None

    def slideControlSequence(self, steps):
        self.calls.append(("slide", steps))

    def updateImportanceSamplingControl(self, state, stride):
        self.calls.append(("is", stride))

    def computeControl(self, state, stride):
        self.calls.append(("compute", stride, np.array(state)))

    def getStats(self):
        return types.SimpleNamespace(real_sys=types.SimpleNamespace(free_energy_mean=1.5))

    def getControlSeq(self):
        return np.arange(self.num_timesteps, dtype=np.float32).reshape(-1, 1)  # u[t] = t

    def getTargetStateSeq(self):
        t = np.arange(self.num_timesteps, dtype=np.float32)
        return np.stack([t, -2 * t], 1)

    def modelStep(self, x, u, dt=None, enforce_constraints=True):
        return np.asarray(x, np.float32), np.clip(np.asarray(u, np.float32), -5.0, 5.0)

    def enforceConstraints(self, state, u):
        return np.clip(np.asarray(u, np.float32), -5.0, 5.0)

    def setLambda(self, lam):
        self.lam = lam

    def setNumIters(self, n):
        self.iters = n

    def updateImportanceSampler(self, u):
        self.calls.append(("reset",))


class RecordingPlant(m.BasePlant):
    def __init__(self, ctl, hz=10, stride=1):
        super().__init__(ctl, hz, stride)
        self.pub, self.nominal, self.fe = [], [], []

    def pubControl(self, u):
        self.pub.append(np.array(u))

    def pubNominalState(self, s):
        self.nominal.append(np.array(s))

    def pubFreeEnergyStatistics(self, stats):
        self.fe.append(stats.real_sys.free_energy_mean)

    def getCurrentTime(self):
        return self.state_time_


def alive():
    e = threading.Event()
    e.set()
    return e


def test_interpolation_helpers():
    c = np.arange(10, dtype=np.float32).reshape(-1, 1)
    assert np.allclose(m.interpolateControls(0.25, c, 0.1), [2.5])
    s = np.stack([np.arange(10.0), -np.arange(10.0)], 1).astype(np.float32)
    assert np.allclose(m.interpolateState(s, 0.31, 0.1), [3.1, -3.1], atol=1e-5)
    gains = np.zeros((10, 2, 1), np.float32)
    gains[:, 0, 0] = np.arange(10)
    # k(t) = K[t]^T (x - x*) = t * 0.5, interpolated half-way between knots 2 and 3
    assert np.allclose(m.interpolateFeedback([1.5, 0.0], [1.0, 0.0], 0.25, gains, 0.1), [1.25])


def test_first_iteration_has_stride_zero_and_no_slide():
    ctl = MockController()
    p = RecordingPlant(ctl)
    p.updateState(np.array([1.0, 2.0]), 0.0)
    assert p.pub == []  # nothing optimised yet: nothing published (base_plant.hpp:299-303)
    p.runControlIteration(alive())
    assert [c[0] for c in ctl.calls] == ["compute"] and ctl.calls[0][1] == 0
    assert p.getLastOptimizationStride() == 0 and p.num_iter_ == 1 and p.fe == [1.5]
    assert p.last_used_state_update_time_ == 0.0


def test_stride_follows_robot_time_and_slides():
    ctl = MockController(T=10, dt=0.1)
    p = RecordingPlant(ctl, stride=1)
    p.updateState(np.zeros(2), 0.0)
    p.runControlIteration(alive())
    ctl.calls.clear()
    p.updateState(np.zeros(2), 0.31)  # 3.1 dt of robot time later
    p.runControlIteration(alive())
    assert ctl.calls[0] == ("is", 3) and ctl.calls[1] == ("slide", 3) and ctl.calls[2][:2] == ("compute", 3)
    # a stride >= T is not slid (base_plant.hpp:494)
    ctl.calls.clear()
    p.updateState(np.zeros(2), 5.0)
    p.runControlIteration(alive())
    assert [c[0] for c in ctl.calls] == ["compute"] and ctl.calls[0][1] == 47
    # the target stride is a lower bound
    p.setTargetOptimizationStride(4)
    ctl.calls.clear()
    p.updateState(np.zeros(2), 5.1)
    p.runControlIteration(alive())
    assert ctl.calls[1] == ("slide", 4)


def test_update_state_publishes_interpolated_constrained_control():
    ctl = MockController(T=10, dt=0.1)
    p = RecordingPlant(ctl)
    p.setDebugMode(True)
    p.updateState(np.zeros(2), 1.0)
    p.runControlIteration(alive())
    p.updateState(np.zeros(2), 1.25)
    assert np.allclose(p.pub[-1], [2.5]) and np.allclose(p.nominal[-1], [2.5, -5.0], atol=1e-5)
    p.updateState(np.zeros(2), 1.0 + 0.72)
    assert np.allclose(p.pub[-1], [5.0])  # 7.2 clamped by enforceConstraints
    n = len(p.pub)
    p.updateState(np.zeros(2), 1.0 + 1.0)  # outside the optimised horizon: nothing published
    p.updateState(np.zeros(2), 0.5)        # older than the solution
    assert len(p.pub) == n
    # feedback term
    gains = np.zeros((10, 2, 1), np.float32)
    gains[:, 0, 0] = 1.0
    p.setFeedbackGains(gains)
    p.updateState(np.array([3.0, 0.0], np.float32), 1.2)  # nominal x0 at 0.2 is 2.0 -> u_fb = 1
    assert np.allclose(p.pub[-1], [3.0])


def test_nan_state_skips_iteration_and_parameter_queue():
    ctl = MockController()
    p = RecordingPlant(ctl)
    p.updateState(np.array([np.nan, 0.0]), 0.0)
    p.runControlIteration(alive())
    assert ctl.calls == [] and p.num_iter_ == 0
    p.setControllerParams(lambda_=0.7, num_iters=3)
    assert p.hasNewControllerParams()
    assert p.updateParameters() and ctl.lam == 0.7 and ctl.iters == 3 and not p.hasNewControllerParams()
    assert not p.updateParameters()


def test_control_loop_thread_paces_on_state_time():
    ctl = MockController(T=10, dt=0.1)
    p = RecordingPlant(ctl, hz=10, stride=1)
    e = alive()
    th = threading.Thread(target=p.runControlLoop, args=(e,))
    th.start()
    try:
        import time
        for i in range(4):
            p.updateState(np.zeros(2), 0.1 * i)
            deadline = time.monotonic() + 5.0
            while p.num_iter_ < i + 1 and time.monotonic() < deadline:
                time.sleep(1e-3)
            assert p.num_iter_ == i + 1
    finally:
        e.clear()
        th.join(5.0)
    assert not th.is_alive()
    strides = [c[1] for c in ctl.calls if c[0] == "compute"]
    assert strides == [0, 1, 1, 1] and p.avg_loop_time_ms_ > 0


@pytest.mark.gpu
def test_simulated_plant_drives_cartpole_to_goal(gpu):
    cfg = cartpole_cfg(K=2048, T=100)
    eng = make_engine(cfg)
    plant = m.SimulatedPlant(eng, hz=50, optimization_stride=1, init_state=cfg["x0"])
    x = plant.runSimulation(600)
    assert plant.num_iter_ == 601 and plant.getLastOptimizationStride() == 1
    assert abs(x[0] - 20.0) < 3.0  # the cart reaches the goal position (examples/cartpole_example.cu goal [20,0,pi,0])
    assert plant.avg_optimize_time_ms_ > 0 and len(plant.published_controls_) == 600
    assert all(abs(u[0]) <= 5.0 + 1e-6 for _, u in plant.published_controls_)
    # stride 2: optimise every second tick, slide by 2
    eng2 = make_engine(cfg)
    plant2 = m.SimulatedPlant(eng2, hz=50, optimization_stride=2, init_state=cfg["x0"])
    plant2.runSimulation(100)
    assert plant2.num_iter_ == 51 and plant2.getLastOptimizationStride() == 2


# ---------------------------------------------------------------------------------------------------------------------
# BufferedPlant: the per-cycle history hook (core/base_plant.hpp:477-482, core/buffered_plant.hpp, core/buffer.hpp:180-264)
# ---------------------------------------------------------------------------------------------------------------------
class BufferedRecordingPlant(m.BufferedPlantMixin, RecordingPlant):
    pass


def test_buffered_plant_history_and_lstm_hook():
    """irregular samples -> tau / dt + 1 interpolated ones ending at the state's time; empty while the history is shorter than
    tau; the initialiser's known answer (lstm_lstm_helper_test.cu:161-180: all ones -> 101) reaches the controller BEFORE the
    optimisation of the same cycle; a missing key leaves the model alone"""
    ctl = MockController()
    sets = []
    ctl.setLSTMInitialState = lambda h, c: sets.append((len(ctl.calls), np.array(h), np.array(c)))
    plant = BufferedRecordingPlant(ctl)
    assert not plant.checkRequiresBuffer()
    for t in (0.0, 0.03, 0.11, 0.2, 0.45, 0.46, 0.8, 0.95):
        plant.updateExtraValue("RAMP", 2.0 * t, t)
    assert plant.getSmoothedBuffer(0.95) == {}
    plant.updateExtraValue("RAMP", 2.4, 1.2)
    plant.updateExtraValue("RAMP", 0.0, 0.5)  # older than the newest sample: dropped
    buf = plant.getSmoothedBuffer(1.2)
    assert buf["RAMP"].shape == (51,)
    assert np.allclose(buf["RAMP"], 2.0 * (1.2 - (50 - np.arange(51)) * 0.02), atol=1e-5)
    plant.cleanBuffers(2.9)
    assert plant.getInterpState(0.0)["RAMP"] == np.float32(1.9)
    plant.clearBuffers()

    helper = m.LSTMLSTMHelper(8, 60, [68, 100, 20], 8, 10, [18, 2], 6)
    helper.setInitParams(np.ones_like(helper.init_lstm), np.ones_like(helper.init_output))
    keys = ["K%d" % i for i in range(8)]
    plant.setLSTMBufferInit(helper, keys, [0.5, 0.5] + [1.0] * 6)
    assert plant.checkRequiresBuffer()
    alive = threading.Event()
    alive.set()
    plant.updateState(np.zeros(2, np.float32), 0.0)
    plant.runControlIteration(alive)  # no history yet: the optimisation runs, the model keeps its state
    assert [c[0] for c in ctl.calls].count("compute") == 1 and not sets
    for k in range(61):
        for i, key in enumerate(keys):
            plant.updateExtraValue(key, 2.0 if i < 2 else 1.0, 0.02 * k)
    plant.updateState(np.zeros(2, np.float32), 1.2)
    plant.runControlIteration(alive)
    assert len(sets) == 1 and plant.num_buffer_updates_ == 1
    at, hidden, cell = sets[0]
    assert np.all(hidden == 101.0) and np.all(cell == 101.0) and hidden.shape == (10,)
    assert ctl.calls[at][0] in ("is", "slide", "compute") and [c[0] for c in ctl.calls[:at]].count("compute") == 1  # before compute #2
    plant.setLSTMBufferInit(helper, ["NOT_THERE"] + keys[1:], [1.0] * 8)
    plant.updateState(np.zeros(2, np.float32), 1.3)
    plant.runControlIteration(alive)
    assert len(sets) == 1


def test_cpp_buffered_plant_probe(lib):
    """the same on the C++ class (include/mppi_amd/plant.hpp: mppi_amd::BufferedPlant) with a stub controller: g++ only, no device"""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "examples", "_build", "buffered_plant_probe")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    lib_dir = os.path.dirname(m.library_path())
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(repo, "include"),
                        os.path.join(repo, "tests", "probes", "buffered_plant_probe.cpp"), "-L" + lib_dir, "-lmppi_amd",
                        "-Wl,-rpath," + lib_dir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "BUFFERED PLANT OK" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_buffered_plant_closed_loop_lstm_steering(gpu):
    """racer_dubins_elevation_lstm_steering in a plant-driven closed loop: every cycle the plant turns the last second of
    (STEER_ANGLE, STEER_ANGLE_RATE, CAN_STEER_CMD) into (h0, c0) with the initialiser LSTM and hands it to the engine before
    computeControl (racer_dubins_elevation_lstm_steering.cu:216-233 through base_plant.hpp:477-482).  Checked: the hook fires on
    every cycle once a second of history exists; what it set is initializeLSTM of exactly that history; and the engine's
    rollouts really start from it — an oracle whose blob carries the same (h0, c0) gives the same trajectory costs, 0 ulp."""
    from common import host_noise, make_oracle, ulp_diff
    from test_racer_dubins_lstm_steering import H, LSTM_PARAMS, OUT_LAYERS, S_STEER, S_STEER_RATE, S_VEL, steering_cfg
    cfg = steering_cfg(K=1024, T=64)
    eng = make_engine(cfg)
    rng = np.random.default_rng(4)
    helper = m.LSTMLSTMHelper(3, 20, [23, 100, 8], 4, 4, OUT_LAYERS, 11)  # the shape of the reference's tests (:26-32)
    helper.setInitParams(rng.uniform(-0.2, 0.2, helper.init_lstm.size), rng.uniform(-0.2, 0.2, helper.init_output.size))

    class Plant(m.BufferedPlantMixin, m.SimulatedPlant):
        def stepSimulation(self):
            super().stepSimulation()
            x, u = self.sim_state_, self.current_control_
            for key, v in (("STEER_ANGLE", x[S_STEER]), ("STEER_ANGLE_RATE", x[S_STEER_RATE]), ("CAN_STEER_CMD", u[1])):
                self.updateExtraValue(key, v, self.sim_time_)
            self.cleanBuffers(self.sim_time_)

    plant = Plant(eng, int(round(1.0 / cfg["dt"])), 1, init_state=cfg["x0"])
    plant.setLSTMBufferInit(helper, ["STEER_ANGLE", "STEER_ANGLE_RATE", "CAN_STEER_CMD"], [0.2, 0.2, 1.0])
    x = plant.runSimulation(80)
    assert np.isfinite(x).all() and x[S_VEL] > 1.5
    # hz ticks fill the one second of history the smoothed buffer reaches back; from then on EVERY cycle initialises the LSTM
    assert abs(plant.num_buffer_updates_ - (80 - plant.hz_)) <= 1, (plant.num_buffer_updates_, plant.hz_)
    buf = plant.getSmoothedBuffer(plant.last_used_state_update_time_)
    rows = np.stack([buf["STEER_ANGLE"] * np.float32(0.2), buf["STEER_ANGLE_RATE"] * np.float32(0.2), buf["CAN_STEER_CMD"]])
    hidden, cell = helper.initializeLSTM(rows)
    assert np.array_equal(hidden, plant.last_hidden_cell_[0]) and np.array_equal(cell, plant.last_hidden_cell_[1])
    assert np.abs(hidden).max() > 1e-3  # a non-trivial initial state
    # the engine's rollouts start from it
    eps = host_noise(1, cfg["K"], cfg["T"], 2)
    eng.injectNoise(eps)
    mean = eng.getControlSeq()
    eng.updateImportanceSampler(mean)
    eng.computeControl(x, 1)
    cfg["blobs"]["lstm_weights"] = cfg["blobs"]["lstm_weights"].copy()
    cfg["blobs"]["lstm_weights"][LSTM_PARAMS:LSTM_PARAMS + H] = hidden
    cfg["blobs"]["lstm_weights"][LSTM_PARAMS + H:LSTM_PARAMS + 2 * H] = cell
    orc = make_oracle(cfg)
    orc.set_nominal_control(mean)
    orc.vanilla_compute_control(x, 1, eps)
    assert int(ulp_diff(eng.getSampledCostSeq(), orc.costs()).max()) == 0
    # (the first two steps of the smoothed sequence see the control history of the loop, which the fresh oracle does not have)
    assert np.abs(eng.getControlSeq()[2:] - orc.control()[2:]).max() <= 1e-5
    eng.close()
