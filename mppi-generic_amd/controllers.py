"""Host-side mirror of the reference's controller classes over the C ABI.

Method names follow the reference (controllers/controller.cuh, controllers/MPPI/mppi_controller.cuh,
controllers/Tube-MPPI/tube_mppi_controller.cuh): computeControl, getControlSeq, getTargetStateSeq,
slideControlSequence, updateImportanceSampler, getBaselineCost, getNormalizerCost, ...
All numerics happen in libmppi_amd.so; arrays are numpy float32, row-major ([T][C], [T][S], [K][T][C]).
"""
import ctypes as C

import numpy as np

from .capi import MppiConfig, MppiGaussianParams, MppiStats, load_library

MPPI_CONTROLLER_VANILLA = 0
MPPI_CONTROLLER_TUBE = 1
MPPI_CONTROLLER_ROBUST = 2
MPPI_CONTROLLER_COLORED = 3
MPPI_NOISE_PHILOX_FUSED = 0
MPPI_NOISE_INJECTED = 1
MPPI_NOISE_ROCRAND_HOST = 2
MPPI_KERNEL_AUTO, MPPI_KERNEL_FUSED, MPPI_KERNEL_PIPELINE = 0, 1, 2
MPPI_REDUCTION_FUSED, MPPI_REDUCTION_REFERENCE_ORDER, MPPI_REDUCTION_REFERENCE_ORDER_FMA = 0, 1, 2


class MPPIError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("mppi status %d: %s" % (status, message))
        self.status = status


# ---- POD parameter blocks (include/mppi_amd/model_params.h) -------------------------------------------------------
class CartpoleDynamicsParams(C.Structure):
    _fields_ = [("cart_mass", C.c_float), ("pole_mass", C.c_float), ("pole_length", C.c_float)]

    def __init__(self, cart_mass=1.0, pole_mass=1.0, pole_length=1.0):
        super().__init__(cart_mass, pole_mass, pole_length)


class CartpoleQuadraticCostParams(C.Structure):
    _fields_ = [
        ("control_cost_coeff", C.c_float * 1), ("discount", C.c_float),
        ("cart_position_coeff", C.c_float), ("cart_velocity_coeff", C.c_float), ("pole_angle_coeff", C.c_float),
        ("pole_angular_velocity_coeff", C.c_float), ("terminal_cost_coeff", C.c_float),
        ("desired_terminal_state", C.c_float * 4),
    ]

    def __init__(self):
        super().__init__()
        self.control_cost_coeff[0] = 10.0
        self.discount = 1.0
        self.cart_position_coeff = 1000
        self.cart_velocity_coeff = 100
        self.pole_angle_coeff = 2000
        self.pole_angular_velocity_coeff = 100
        self.terminal_cost_coeff = 0
        self.desired_terminal_state[:] = [0.0, 0.0, np.float32(np.pi), 0.0]


class DoubleIntegratorParams(C.Structure):
    _fields_ = [("system_noise", C.c_float)]

    def __init__(self, system_noise=1.0):
        super().__init__(system_noise)


class DoubleIntegratorCircleCostParams(C.Structure):
    _fields_ = [
        ("control_cost_coeff", C.c_float * 2), ("discount", C.c_float), ("velocity_cost", C.c_float),
        ("crash_cost", C.c_float), ("velocity_desired", C.c_float), ("inner_path_radius2", C.c_float),
        ("outer_path_radius2", C.c_float), ("angular_momentum_desired", C.c_float),
    ]

    def __init__(self):
        super().__init__()
        self.control_cost_coeff[:] = [0.01, 0.01]
        self.discount = 1.0
        self.velocity_cost = 1
        self.crash_cost = 1000
        self.velocity_desired = 2
        self.inner_path_radius2 = 1.875 * 1.875
        self.outer_path_radius2 = 2.125 * 2.125
        self.angular_momentum_desired = 4


class RacerDubinsParams(C.Structure):
    """mppi_racer_dubins_params (reference: dynamics/racer_dubins/racer_dubins.cuh:67-87)"""
    _fields_ = [("c_t", C.c_float * 3), ("c_b", C.c_float * 3), ("c_v", C.c_float * 3), ("c_0", C.c_float),
                ("steering_constant", C.c_float), ("steer_command_angle_scale", C.c_float), ("steer_angle_scale", C.c_float),
                ("max_steer_angle", C.c_float), ("max_steer_rate", C.c_float), ("steer_accel_constant", C.c_float),
                ("steer_accel_drag_constant", C.c_float), ("brake_delay_constant", C.c_float),
                ("brake_delay_constant_neg", C.c_float), ("max_brake_rate_neg", C.c_float), ("max_brake_rate_pos", C.c_float),
                ("wheel_base", C.c_float), ("low_min_throttle", C.c_float), ("gravity", C.c_float), ("gear_sign", C.c_int)]

    def __init__(self):
        super().__init__()
        self.c_t[:] = [1.3, 2.6, 3.9]
        self.c_b[:] = [2.5, 3.5, 4.5]
        self.c_v[:] = [3.7, 4.7, 5.7]
        self.c_0 = 4.9
        self.steering_constant = 0.6
        self.steer_command_angle_scale = 5
        self.steer_angle_scale = -9.1
        self.max_steer_angle = 0.5
        self.max_steer_rate = 5
        self.steer_accel_constant = 12.1
        self.steer_accel_drag_constant = 1.0
        self.brake_delay_constant = 6.6
        self.brake_delay_constant_neg = 8.2
        self.max_brake_rate_neg = 0.9
        self.max_brake_rate_pos = 0.33
        self.wheel_base = 0.3
        self.low_min_throttle = 0.13
        self.gravity = -9.81
        self.gear_sign = 1


class RacerDubinsElevationParams(C.Structure):
    """mppi_racer_dubins_elevation_params (reference: dynamics/racer_dubins/racer_dubins_elevation.cuh:16-60)"""
    _fields_ = [("base", RacerDubinsParams), ("clamp_ax", C.c_float), ("K_x", C.c_float), ("K_y", C.c_float),
                ("K_yaw", C.c_float), ("K_vel_x", C.c_float), ("Q_x_acc", C.c_float), ("Q_x_v", C.c_float * 3),
                ("Q_y_f", C.c_float), ("Q_omega_v", C.c_float), ("Q_omega_steering", C.c_float)]

    def __init__(self):
        super().__init__()
        RacerDubinsParams.__init__(self.base)
        self.clamp_ax = 5.5
        self.K_x = self.K_y = self.K_yaw = self.K_vel_x = 1.0
        self.Q_x_acc = 1.0
        self.Q_x_v[:] = [41.74219, -0.8187027, -2.2131343]
        self.Q_y_f = 0.1
        self.Q_omega_v = 0.001
        self.Q_omega_steering = 0.0


class RacerDubinsSuspensionParams(C.Structure):
    """mppi_racer_dubins_suspension_params (reference: dynamics/racer_dubins/racer_dubins_elevation_suspension_lstm.cuh:17-66)"""
    _fields_ = [("elevation", RacerDubinsElevationParams), ("spring_k", C.c_float), ("drag_c", C.c_float),
                ("mass", C.c_float), ("I_xx", C.c_float), ("I_yy", C.c_float), ("wheel_radius", C.c_float),
                ("c_g", C.c_float * 3)]

    def __init__(self):
        super().__init__()
        RacerDubinsElevationParams.__init__(self.elevation)
        self.spring_k, self.drag_c, self.mass = 14000.0, 1000.0, 1447.0
        self.I_xx = np.float32(1.0) / np.float32(12) * np.float32(1447.0) * np.float32(2) * np.float32(2.25)
        self.I_yy = np.float32(1.0) / np.float32(12) * np.float32(1447.0) * np.float32(11.25)
        self.wheel_radius = 0.32
        self.c_g[:] = [2.981 * 0.5, 0.0, 0.0]

    @property
    def base(self):
        return self.elevation.base


class RacerDubinsUncertaintyParams(C.Structure):
    """mppi_racer_dubins_uncertainty_params (reference: dynamics/racer_dubins/racer_dubins_elevation_lstm_unc.cuh:5-48)"""
    _fields_ = [("suspension", RacerDubinsSuspensionParams), ("unc_scale", C.c_float * 7),
                ("pos_quad_brake_c", C.c_float * 3), ("neg_quad_brake_c", C.c_float * 3), ("use_static_settling", C.c_int)]

    def __init__(self):
        super().__init__()
        RacerDubinsSuspensionParams.__init__(self.suspension)
        self.unc_scale[:] = [1.0] * 7
        self.pos_quad_brake_c[:] = [2.0, 0.5, 0.3]
        self.neg_quad_brake_c[:] = [5.84, 0.15, 1.7]
        self.use_static_settling = 1

    @property
    def base(self):
        return self.suspension.elevation.base

    @property
    def elevation(self):
        return self.suspension.elevation


class QuadraticCostParams28(C.Structure):
    """mppi_quadratic_cost_params_28 (reference: QuadraticCostTrajectoryParams<RacerDubins, 1>,
    cost_functions/quadratic_cost/quadratic_cost.cuh:11-63)"""
    _fields_ = [("control_cost_coeff", C.c_float * 2), ("discount", C.c_float), ("s_goal", C.c_float * 28),
                ("s_coeffs", C.c_float * 28), ("current_time", C.c_int)]

    def __init__(self):
        super().__init__()
        self.discount = 1.0
        self.s_coeffs[:] = [1.0] * 28


class ARStandardCostParams(C.Structure):
    _fields_ = [
        ("control_cost_coeff", C.c_float * 2), ("discount", C.c_float), ("desired_speed", C.c_float),
        ("speed_coeff", C.c_float), ("track_coeff", C.c_float), ("max_slip_ang", C.c_float),
        ("slip_coeff", C.c_float), ("track_slop", C.c_float), ("crash_coeff", C.c_float),
        ("boundary_threshold", C.c_float), ("grid_res", C.c_int), ("r_c1", C.c_float * 3), ("r_c2", C.c_float * 3),
        ("trs", C.c_float * 3),
    ]

    def __init__(self):
        super().__init__()
        self.control_cost_coeff[:] = [0.0, 0.0]
        self.discount = 1.0
        self.desired_speed = 6.0
        self.speed_coeff = 4.25
        self.track_coeff = 200.0
        self.max_slip_ang = 1.25
        self.slip_coeff = 10.0
        self.track_slop = 0
        self.crash_coeff = 10000
        self.boundary_threshold = 0.65
        self.grid_res = 10
        self.r_c1[:] = [1, 0, 0]
        self.r_c2[:] = [0, 1, 0]
        self.trs[:] = [0, 0, 1]

    def setTransformFromBounds(self, x_min, x_max, y_min, y_max):
        """the transform ARStandardCost::loadTrackData builds (ar_standard_cost.cu:132-137)"""
        self.r_c1[:] = [1.0 / (x_max - x_min), 0, 0]
        self.r_c2[:] = [0, 1.0 / (y_max - y_min), 0]
        self.trs[:] = [-x_min / (x_max - x_min), -y_min / (y_max - y_min), 1]


def fnn_blob_from_npz_dict(d, prefix="dynamics_"):
    """flat FNN parameter blob [W1|b1|W2|b2|...] from the reference's .npz key layout (dynamics_W{i}, dynamics_b{i},
    float64; FNNHelper::loadParams, utils/nn_helpers/fnn_helper.cu:96-174)"""
    parts, i = [], 1
    while prefix + "W%d" % i in d:
        parts.append(np.asarray(d[prefix + "W%d" % i], np.float64).reshape(-1))
        parts.append(np.asarray(d[prefix + "b%d" % i], np.float64).reshape(-1))
        i += 1
    return np.concatenate(parts).astype(np.float32)


def lstm_blob_from_npz_dict(d, prefix=""):
    """LSTM parameter blob [W_im W_fm W_om W_cm | W_ii W_fi W_oi W_ci | b_i b_f b_o b_c | h0 | c0] from the reference's
    .npz key layout ({prefix}lstm/weight_hh_l0, weight_ih_l0, bias_hh_l0, bias_ih_l0, float64, PyTorch gate order
    i, f, g, o; LSTMHelper::loadParams, utils/nn_helpers/lstm_helper.cu:514-578: gates re-ordered to i, f, o, c and the
    two bias vectors summed).  Optional keys {prefix}lstm/h0, c0 give the initial state (default zeros).
    Returns (lstm_blob, output_fnn_blob); the output network uses the keys {prefix}output/dynamics_W{i}, _b{i}."""
    if prefix and not prefix.endswith("/"):
        prefix += "/"
    if "model/" + prefix + "lstm/weight_hh_l0" in d:
        prefix = "model/" + prefix
    whh = np.asarray(d[prefix + "lstm/weight_hh_l0"], np.float64)
    wih = np.asarray(d[prefix + "lstm/weight_ih_l0"], np.float64)
    b = np.asarray(d[prefix + "lstm/bias_hh_l0"], np.float64) + np.asarray(d[prefix + "lstm/bias_ih_l0"], np.float64)
    H = b.size // 4
    order = [0, 1, 3, 2]  # blob gate g <- torch gate order[g]  (i, f, o, c  <-  i, f, g(c), o)
    parts = [whh[g * H:(g + 1) * H].reshape(-1) for g in order]
    parts += [wih[g * H:(g + 1) * H].reshape(-1) for g in order]
    parts += [b[g * H:(g + 1) * H] for g in order]
    parts.append(np.asarray(d.get(prefix + "lstm/h0", np.zeros(H)), np.float64).reshape(-1))
    parts.append(np.asarray(d.get(prefix + "lstm/c0", np.zeros(H)), np.float64).reshape(-1))
    return np.concatenate(parts).astype(np.float32), fnn_blob_from_npz_dict(d, prefix + "output/dynamics_")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class MPPIController:
    """Common part of the reference's Controller<DYN, COST, FB, SAMPLING, MAX_TIMESTEPS, NUM_ROLLOUTS>."""

    KIND = MPPI_CONTROLLER_VANILLA

    def __init__(self, model, num_rollouts, num_timesteps, dt, lambda_, alpha=0.0, num_iters=1, seed=42,
                 noise_source=MPPI_NOISE_PHILOX_FUSED, block_x=0, block_y=0, device=0, stream=None, rank=0,
                 world_size=1, save_samples=False, kernel_variant=0, force_exchange=False):
        self._lib = load_library()
        self._h = C.c_void_p()
        self._model = model.encode()
        cfg = MppiConfig(self._model, self.KIND, num_rollouts, num_timesteps, dt, lambda_, alpha, num_iters, seed,
                         noise_source, block_x, block_y, device, stream, rank, world_size, int(save_samples), kernel_variant,
                         int(force_exchange))
        st = self._lib.mppi_create(C.byref(cfg), C.byref(self._h))
        if st != 0:
            self._h = C.c_void_p()
            raise MPPIError(st, (self._lib.mppi_last_error(None) or b"").decode())
        s, c, o, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.mppi_get_dims(self._h, C.byref(s), C.byref(c), C.byref(o), C.byref(d)))
        self.STATE_DIM, self.CONTROL_DIM, self.OUTPUT_DIM, self.num_systems = s.value, c.value, o.value, d.value
        kl, ko = C.c_int(), C.c_int()
        self._check(self._lib.mppi_get_local_rollouts(self._h, C.byref(kl), C.byref(ko)))
        self.num_rollouts_local, self.rollout_offset = kl.value, ko.value
        self.num_rollouts, self.num_timesteps = num_rollouts, num_timesteps
        self.dt, self.lambda_, self.alpha = dt, lambda_, alpha
        self._init_fast_calls()

    def _init_fast_calls(self):
        """The three calls a control loop makes every cycle (computeControl, getControlSeq, slide) with raw-pointer signatures and
        persistent staging arrays: a numpy.ctypeslib.ndpointer argument costs ~3.3 us of marshalling PER CALL on this class of
        host (measured: 3.31 us against 0.21 us for an integer address) — 7-8 us of a 45 us closed-loop cycle were this wrapper,
        not the engine.  Same library entry points, same values; everything else keeps the checked ndpointer signatures."""
        def raw(name, *argtypes):
            addr = C.cast(getattr(self._lib, name), C.c_void_p).value
            return C.CFUNCTYPE(C.c_int, *argtypes)(addr)
        self._fast_compute = raw("mppi_compute_control", C.c_void_p, C.c_void_p, C.c_int)
        self._fast_get_control = raw("mppi_get_control_seq", C.c_void_p, C.c_void_p)
        self._fast_slide = raw("mppi_slide", C.c_void_p, C.c_int)
        self._hv = self._h.value
        self._x_stage = np.zeros(2 * self.STATE_DIM, np.float32)  # [S], or [D][S] where a caller hands both systems' states
        self._x_ptr = self._x_stage.ctypes.data
        self._u_stage = np.empty((self.num_timesteps, self.CONTROL_DIM), np.float32)
        self._u_ptr = self._u_stage.ctypes.data

    def launchCounts(self):
        """(rollout launches, reduction-stage launches) of this handle since creation (mppi_get_launch_counts)"""
        r, g = C.c_ulonglong(), C.c_ulonglong()
        self._check(self._lib.mppi_get_launch_counts(self._h, C.byref(r), C.byref(g)))
        return r.value, g.value

    # -- plumbing --
    def _check(self, st):
        if st != 0:
            raise MPPIError(st, (self._lib.mppi_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.mppi_destroy(self._h)
            self._h = C.c_void_p()
            self._hv = None  # (a call on a closed controller reaches the library with a null handle: MPPI_ERR_INVALID_ARG)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- parameters --
    def setDynamicsParams(self, pod):
        self._check(self._lib.mppi_set_dynamics_params(self._h, C.byref(pod), C.sizeof(pod)))

    def setCostParams(self, pod):
        self._check(self._lib.mppi_set_cost_params(self._h, C.byref(pod), C.sizeof(pod)))

    def setSamplingParams(self, std_dev, control_cost_coeff=None, pure_noise_trajectories_percentage=0.01,
                          std_dev_decay=1.0, sum_strides=32):
        sd = _f32(std_dev).reshape(-1)
        if sd.size == self.CONTROL_DIM and self.num_systems == 2:
            sd = np.concatenate([sd, sd])
        cc = _f32(np.zeros(self.CONTROL_DIM) if control_cost_coeff is None else control_cost_coeff).reshape(-1)
        p = MppiGaussianParams(sd.ctypes.data_as(C.POINTER(C.c_float)), cc.ctypes.data_as(C.POINTER(C.c_float)),
                               pure_noise_trajectories_percentage, std_dev_decay, sum_strides)
        self._check(self._lib.mppi_set_sampler_params(self._h, C.byref(p)))

    def setTimeSpecificStdDev(self, std_dev):
        """std_dev [D][T][C] (or [T][C] for one distribution), None switches back to one sigma per control"""
        if std_dev is None:
            self._check(self._lib.mppi_set_time_specific_std_dev(self._h, None))
            return
        a = _f32(std_dev).reshape(-1)
        assert a.size == self.num_systems * self.num_timesteps * self.CONTROL_DIM, a.size
        self._check(self._lib.mppi_set_time_specific_std_dev(self._h, a.ctypes.data))

    def setModelBlob(self, name, array):
        """bulk model data (NN weights, costmap) — see mppi_set_model_blob"""
        a = _f32(array)
        dims = (C.c_int * a.ndim)(*a.shape)
        self._check(self._lib.mppi_set_model_blob(self._h, name.encode(), a.reshape(-1), a.size, dims, a.ndim))

    def setLSTMInitialState(self, hidden, cell):
        """new initial hidden / cell state of the rollouts' LSTM (LSTMHelper::setHiddenState / setCellState +
        copyHiddenCellToDevice, utils/nn_helpers/lstm_helper.cu:476-500)"""
        self._check(self._lib.mppi_set_lstm_initial_state(self._h, _f32(hidden).reshape(-1), _f32(cell).reshape(-1)))

    def loadNpz(self, kind, path, prefix=None):
        """model data straight from the reference's .npz layout (kind: "dynamics" | "lstm" | "costmap"); mppi_load_npz"""
        self._check(self._lib.mppi_load_npz(self._h, kind.encode(), str(path).encode(),
                                            None if prefix is None else prefix.encode()))

    def setControlRanges(self, lo_hi):
        self._check(self._lib.mppi_set_control_ranges(self._h, _f32(lo_hi).reshape(-1)))

    def setControlDeadbands(self, db):
        self._check(self._lib.mppi_set_control_deadband(self._h, _f32(db).reshape(-1)))

    def setLambda(self, lambda_):
        self.lambda_ = lambda_
        self._check(self._lib.mppi_set_lambda_alpha(self._h, self.lambda_, self.alpha))

    def setNumIters(self, n):
        self._check(self._lib.mppi_set_num_iters(self._h, n))

    def setReductionMode(self, mode):
        """MPPI_REDUCTION_FUSED (default) | MPPI_REDUCTION_REFERENCE_ORDER | MPPI_REDUCTION_REFERENCE_ORDER_FMA:
        mppi_set_reduction_mode — the last stage of an iteration in the reference's own arithmetic order"""
        self._check(self._lib.mppi_set_reduction_mode(self._h, int(mode)))

    def setSlideControlScale(self, scale):
        self._check(self._lib.mppi_set_slide_control_scale(self._h, _f32(scale).reshape(-1)))

    def setSeed(self, seed):
        self._check(self._lib.mppi_set_seed(self._h, seed))

    # -- control loop --
    def updateImportanceSampler(self, u):
        u = _f32(u)
        assert u.shape == (self.num_timesteps, self.CONTROL_DIM)
        self._check(self._lib.mppi_set_nominal_control(self._h, u))

    def injectNoise(self, eps):
        """eps[n_iters][K_local][T][C] (or [K_local][T][C]); None switches back to the generator."""
        if eps is None:
            self._check(self._lib.mppi_inject_noise(self._h, None, 0))
            return
        eps = _f32(eps)
        if eps.ndim == len(self._noise_shape()):
            eps = eps[None]
        assert eps.shape[1:] == self._noise_shape(), (eps.shape, self._noise_shape())
        self._check(self._lib.mppi_inject_noise(self._h, eps.ctypes.data, eps.shape[0]))

    def _noise_shape(self):
        base = (self.num_rollouts_local, self.num_timesteps, self.CONTROL_DIM)
        return ((self.num_systems,) + base) if getattr(self, "_independent_noise", False) else base

    def setIndependentNoise(self, independent=True):
        """use_same_noise_for_all_distributions = not independent (sampling_distribution.cuh:20); injected noise then is
        eps[n_iters][D][K_local][T][C]"""
        self._check(self._lib.mppi_set_independent_noise(self._h, int(independent)))
        self._independent_noise = bool(independent)

    def sampleNoise(self, optimization_stride=1):
        """raw eps[K_local][T][C] of the next iteration (generator tests; see mppi_sample_noise)"""
        eps = np.empty((self.num_rollouts_local, self.num_timesteps, self.CONTROL_DIM), np.float32)
        self._check(self._lib.mppi_sample_noise(self._h, optimization_stride, eps))
        return eps

    def computeControl(self, state, optimization_stride=1):
        if type(state) is not np.ndarray:
            state = np.asarray(state, np.float32)
        n = state.size
        if n > self._x_stage.size:  # (not a state of this model: let the checked path say so)
            self._check(self._lib.mppi_compute_control(self._h, _f32(state).reshape(-1), optimization_stride))
            return
        self._x_stage[:n] = state if state.ndim == 1 else state.reshape(-1)
        st = self._fast_compute(self._hv, self._x_ptr, optimization_stride)
        if st != 0:
            self._check(st)

    def getControlSeq(self):
        st = self._fast_get_control(self._hv, self._u_ptr)
        if st != 0:
            self._check(st)
        return self._u_stage.copy()

    def getTargetStateSeq(self):
        x = np.empty((self.num_timesteps, self.STATE_DIM), np.float32)
        self._check(self._lib.mppi_get_state_seq(self._h, x))
        return x

    def getTargetOutputSeq(self):
        y = np.empty((self.num_timesteps, self.OUTPUT_DIM), np.float32)
        self._check(self._lib.mppi_get_output_seq(self._h, y))
        return y

    def slideControlSequence(self, steps):
        st = self._fast_slide(self._hv, steps)
        if st != 0:
            self._check(st)

    def getSampledCostSeq(self):
        costs = np.empty((self.num_systems, self.num_rollouts_local), np.float32)
        self._check(self._lib.mppi_get_costs(self._h, costs))
        return costs

    def getSampledControls(self):
        v = np.empty((self.num_systems, self.num_rollouts_local, self.num_timesteps, self.CONTROL_DIM), np.float32)
        self._check(self._lib.mppi_get_sampled_controls(self._h, v))
        return v

    def getStats(self):
        st = MppiStats()
        self._check(self._lib.mppi_get_stats(self._h, C.byref(st)))
        return st

    def getBaselineCost(self, system=0):
        st = self.getStats()
        return (st.real_sys if system == 0 else st.nominal_sys).baseline

    def getNormalizerCost(self, system=0):
        st = self.getStats()
        return (st.real_sys if system == 0 else st.nominal_sys).normalizer

    # -- kernel-level / device-resident --
    def rolloutCosts(self, x0, optimization_stride=1):
        self._check(self._lib.mppi_rollout_costs(self._h, _f32(x0).reshape(-1), optimization_stride))
        return self.getSampledCostSeq()

    def uploadState(self, x0):
        self._check(self._lib.mppi_upload_state(self._h, _f32(x0).reshape(-1)))

    def getOptimalControlSeq(self):
        """raw u* of the last iteration ([D][T][C]); reference: setHostOptimalControlSequence"""
        u = np.empty((self.num_systems, self.num_timesteps, self.CONTROL_DIM), np.float32)
        self._check(self._lib.mppi_get_optimal_control(self._h, u))
        return u

    def optimize(self, num_iterations, synchronize=True):
        self._check(self._lib.mppi_optimize(self._h, num_iterations, int(synchronize)))

    def timeIterations(self, n):
        a, b = C.c_float(), C.c_float()
        self._check(self._lib.mppi_time_iterations(self._h, n, C.byref(a), C.byref(b)))
        return a.value, b.value

    def synchronize(self):
        self._check(self._lib.mppi_synchronize(self._h))

    def chooseAppropriateKernel(self, num_evaluations=10):
        """times the fused and the role-pipelined rollout kernel and keeps the faster (mppi_controller.cu:44-143);
        returns (variant, fused_ms, pipeline_ms) with variant 1 = fused, 2 = pipeline"""
        v, a, b = C.c_int(), C.c_float(), C.c_float()
        self._check(self._lib.mppi_choose_kernel(self._h, num_evaluations, C.byref(v), C.byref(a), C.byref(b)))
        return v.value, a.value, b.value

    def enforceConstraints(self, state, u):
        """Dynamics::enforceConstraints on one control vector (host-side and lock-free for the base rule)"""
        u = _f32(u).reshape(-1).copy()
        sp = None if state is None else _f32(state).reshape(-1).ctypes.data
        self._check(self._lib.mppi_enforce_constraints(self._h, sp, u))
        return u

    def modelStep(self, x, u, dt=None, enforce_constraints=True):
        x = _f32(x).reshape(-1).copy()
        u = _f32(u).reshape(-1).copy()
        self._check(self._lib.mppi_model_step(self._h, x, u, self.dt if dt is None else dt, int(enforce_constraints)))
        return x, u

    # -- multi-GPU --
    def exchangeBuffers(self):
        s, r, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._check(self._lib.mppi_get_exchange_buffers(self._h, C.byref(s), C.byref(r), C.byref(n)))
        return s.value, r.value, n.value

    def readSendRecord(self):
        n = self.exchangeBuffers()[2]
        out = np.empty(n, np.float32)
        self._check(self._lib.mppi_read_send_record(self._h, out))
        return out

    def writeRecvRecords(self, records):
        r = _f32(records).reshape(-1)
        self._check(self._lib.mppi_write_recv_records(self._h, r))

    def iterationLocal(self):
        self._check(self._lib.mppi_iteration_local(self._h))

    def iterationMerge(self):
        self._check(self._lib.mppi_iteration_merge(self._h))

    def p2pMailboxHandle(self):
        """64-byte hipIpcMemHandle_t of this rank's mailbox (to be exchanged between the processes)"""
        buf = C.create_string_buffer(64)
        n = C.c_size_t()
        self._check(self._lib.mppi_p2p_mailbox_handle(self._h, buf, 64, C.byref(n)))
        return bytes(buf.raw[:n.value])

    def p2pReset(self):
        """ends this rank's mailbox session (mppi_p2p_reset): every rank calls it before a NEW session exports / connects"""
        self._check(self._lib.mppi_p2p_reset(self._h))

    def p2pConnect(self, handles):
        """handles: list of world byte strings (rank order) from p2pMailboxHandle of every rank"""
        blob = b"".join(h.ljust(64, b"\0") for h in handles)
        self._check(self._lib.mppi_p2p_connect(self._h, blob, 64))

    @staticmethod
    def p2pConnectLocal(controllers):
        """all ranks of one problem living in this process: wire their mailboxes together"""
        arr = (C.c_void_p * len(controllers))(*[c._h.value for c in controllers])
        for c in controllers:
            c._check(c._lib.mppi_p2p_connect_local(c._h, arr))

    def commInitRccl(self, unique_id_bytes):
        buf = C.create_string_buffer(bytes(unique_id_bytes), 128)
        self._check(self._lib.mppi_comm_init_rccl(self._h, buf, 128))


class VanillaMPPIController(MPPIController):
    """reference: controllers/MPPI/mppi_controller.cuh — VanillaMPPIController"""
    KIND = MPPI_CONTROLLER_VANILLA


class ColoredMPPIController(MPPIController):
    """reference: controllers/ColoredMPPI/colored_mppi_controller.cuh — ColoredMPPIController with
    ColoredNoiseDistribution (sampling_distributions/colored_noise/colored_noise.cuh)"""
    KIND = MPPI_CONTROLLER_COLORED

    def setColoredNoiseParams(self, exponents, offset_decay_rate=0.97, fmin=0.0):
        self._check(self._lib.mppi_set_colored_noise_params(self._h, _f32(exponents).reshape(-1), offset_decay_rate, fmin))

    def setColoredMPPIParams(self, gamma=0.0, r_exp=0.0, state_leash_dist=None, leash_active=False, leash_jump=1):
        """setGamma / setRExp (Tsallis weights when both are non-zero) and the state leash (setStateLeashLength,
        setLeashActive; colored_mppi_controller.cuh:95-193)"""
        p = None if state_leash_dist is None else _f32(state_leash_dist).reshape(-1).ctypes.data
        self._check(self._lib.mppi_set_colored_mppi_params(self._h, gamma, r_exp, p, int(leash_active), leash_jump))

    def _noise_shape(self):
        """spectrum noise, the reference's samples_in_freq_complex_d_ layout: [K][C][T+1][2]"""
        return (self.num_rollouts_local, self.CONTROL_DIM, self.num_timesteps + 1, 2)


class RobustMPPIController(MPPIController):
    """reference: controllers/R-MPPI/robust_mppi_controller.cuh — RobustMPPIController.
    Systems: 0 = nominal, 1 = real.  The DDP gain producer is the caller's (setFeedbackGains)."""
    KIND = MPPI_CONTROLLER_ROBUST

    def setRMPPIParams(self, value_function_threshold=1000.0, num_candidates=9, samples_per_candidate=32):
        self.num_candidates = num_candidates
        self._check(self._lib.mppi_set_rmppi_params(self._h, value_function_threshold, num_candidates,
                                                    samples_per_candidate))

    def setFeedbackGains(self, gains, accumulate_all_states=False):
        """gains[T][S][C] (DDPFeedbackState::fb_gain_traj_)"""
        g = _f32(gains)
        assert g.shape == (self.num_timesteps, self.STATE_DIM, self.CONTROL_DIM), g.shape
        self._check(self._lib.mppi_set_feedback_gains(self._h, g.reshape(-1), int(accumulate_all_states)))

    def updateImportanceSamplingControl(self, state, stride):
        self._check(self._lib.mppi_update_importance_sampling_control(self._h, _f32(state).reshape(-1), stride))

    def getRMPPIState(self):
        """(nominal_state[S], best_index, nominal_stride, candidate_free_energy[num_candidates])"""
        ns = np.zeros(self.STATE_DIM, np.float32)
        fe = np.zeros(getattr(self, "num_candidates", 9), np.float32)
        bi, st = C.c_int(), C.c_int()
        self._check(self._lib.mppi_get_rmppi_state(self._h, ns.ctypes.data, C.byref(bi), C.byref(st), fe.ctypes.data))
        return ns, bi.value, st.value, fe

    def getNominalControlSeq(self):
        u = np.empty((self.num_timesteps, self.CONTROL_DIM), np.float32)
        self._check(self._lib.mppi_get_nominal_control_seq(self._h, u))
        return u

    def getNominalStateSeq(self):
        x = np.empty((self.num_timesteps, self.STATE_DIM), np.float32)
        self._check(self._lib.mppi_get_nominal_state_seq(self._h, x))
        return x


class TubeMPPIController(MPPIController):
    """reference: controllers/Tube-MPPI/tube_mppi_controller.cuh — TubeMPPIController"""
    KIND = MPPI_CONTROLLER_TUBE

    def setNominalThreshold(self, t):
        self._check(self._lib.mppi_set_nominal_threshold(self._h, t))

    def getNominalControlSeq(self):
        u = np.empty((self.num_timesteps, self.CONTROL_DIM), np.float32)
        self._check(self._lib.mppi_get_nominal_control_seq(self._h, u))
        return u

    def getNominalStateSeq(self):
        x = np.empty((self.num_timesteps, self.STATE_DIM), np.float32)
        self._check(self._lib.mppi_get_nominal_state_seq(self._h, x))
        return x


# ---- kernel-level operators -----------------------------------------------------------------------------------------
def _op_check(lib, st):
    if st != 0:
        raise MPPIError(st, (lib.mppi_last_error(None) or b"").decode())


def npz_read_array(path, key):
    """one array of an .npz through the library's own reader (host only; returns float64 in C order)"""
    lib = load_library()
    n, nd = C.c_size_t(), C.c_int()
    dims = (C.c_int * 8)()
    _op_check(lib, lib.mppi_npz_read_array(str(path).encode(), key.encode(), None, 0, C.byref(n), dims, C.byref(nd)))
    out = np.empty(n.value, np.float64)
    _op_check(lib, lib.mppi_npz_read_array(str(path).encode(), key.encode(), out.ctypes.data, n.value, C.byref(n), dims,
                                           C.byref(nd)))
    return out.reshape([dims[i] for i in range(nd.value)])


def texture2d_query(data, points, frame, params=None, device=0):
    """TwoDTextureHelper lookups on the device (mppi_texture2d_query): data[h][w] or [h][w][channels], points[n][3],
    frame 0 texture coordinate / 1 map pose / 2 world pose"""
    from .capi import MppiTexture2dParams
    lib = load_library()
    d = _f32(data)
    if d.ndim == 2:
        d = d[:, :, None]
    h, w, ch = d.shape
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros((pts.shape[0], ch), np.float32)
    p = params if params is not None else MppiTexture2dParams()
    _op_check(lib, lib.mppi_texture2d_query(d.reshape(-1), w, h, ch, C.byref(p), pts.reshape(-1), pts.shape[0], frame,
                                            out.reshape(-1), device))
    return out


class LSTMLSTMHelper:
    """Host side of the reference's LSTMLSTMHelper (utils/nn_helpers/lstm_lstm_helper.cuh; C++ in
    include/mppi_amd/utils/nn_helpers/lstm_lstm_helper.hpp): the initialiser LSTM that turns the recent history buffer into
    the initial (hidden, cell) of the prediction LSTM inside the rollouts.  initializeLSTM() returns them; hand them to
    controller.setLSTMInitialState()."""

    def __init__(self, init_input_dim, init_hidden_dim, init_output_layers, input_dim, hidden_dim, output_layers, init_len):
        assert init_output_layers[0] == init_input_dim + init_hidden_dim and init_output_layers[-1] == 2 * hidden_dim
        self.init_input_dim, self.init_hidden_dim = init_input_dim, init_hidden_dim
        self.init_output_layers = list(init_output_layers)
        self.input_dim, self.hidden_dim, self.output_layers, self.init_len = input_dim, hidden_dim, list(output_layers), init_len
        H, I = init_hidden_dim, init_input_dim
        self.init_lstm = np.zeros(4 * H * H + 4 * H * I + 6 * H, np.float32)
        n = sum(self.init_output_layers[i + 1] * (self.init_output_layers[i] + 1) for i in range(len(init_output_layers) - 1))
        self.init_output = np.zeros(n, np.float32)

    def setInitParams(self, lstm_blob, output_blob):
        """initialiser parameters in the device helpers' blob layouts (lstm_blob_from_npz_dict(d, "init_") builds them)"""
        lstm_blob, output_blob = _f32(lstm_blob).reshape(-1), _f32(output_blob).reshape(-1)
        assert lstm_blob.size == self.init_lstm.size and output_blob.size == self.init_output.size
        self.init_lstm, self.init_output = lstm_blob.copy(), output_blob.copy()

    def initializeLSTM(self, buffer):
        """buffer[init_input_dim][cols] as the reference's Eigen matrix (one column per time step); returns (hidden, cell)"""
        buffer = _f32(buffer)
        if buffer.ndim != 2 or buffer.shape[0] != self.init_input_dim or buffer.shape[1] < self.init_len:
            raise ValueError("history buffer must be [init_input_dim][>= init_len]")
        lib = load_library()
        cols = buffer.shape[1]
        samples = np.ascontiguousarray(buffer.T).reshape(-1)
        out = np.zeros(2 * self.hidden_dim, np.float32)
        layers = (C.c_int * len(self.init_output_layers))(*self.init_output_layers)
        _op_check(lib, lib.mppi_lstm_lstm_initialize(self.init_input_dim, self.init_hidden_dim, layers,
                                                     len(self.init_output_layers), self.init_lstm, self.init_output,
                                                     self.hidden_dim, self.init_len, samples, cols, out))
        return out[:self.hidden_dim].copy(), out[self.hidden_dim:].copy()


def det_eval(func, x, device=0):
    lib = load_library()
    x = _f32(x).reshape(-1)
    y = np.empty_like(x)
    _op_check(lib, lib.mppi_det_eval(func, x, y, x.size, device))
    return y


def launch_boundary_us(device=0, n=2000):
    """average time per launch of n dependent trivial kernels (mppi_measure_launch_boundary)"""
    lib = load_library()
    out = C.c_float()
    _op_check(lib, lib.mppi_measure_launch_boundary(device, n, C.byref(out)))
    return float(out.value)


def issue_interval_ns(device=0):
    """ns per dependent 4-byte VALU instruction of a wave alone on its SIMD (mppi_measure_issue_interval)"""
    lib = load_library()
    out = C.c_float()
    _op_check(lib, lib.mppi_measure_issue_interval(device, C.byref(out)))
    return float(out.value)


def philox_normal(seed, generation, K, T, Cdim, k_begin=0, k_end=None, device=0):
    lib = load_library()
    k_end = K if k_end is None else k_end
    out = np.empty((k_end - k_begin, T, Cdim), np.float32)
    _op_check(lib, lib.mppi_philox_normal(seed, generation, K, T, Cdim, k_begin, k_end, out, device))
    return out


def norm_exp(costs, lambda_inv, baseline, device=0):
    lib = load_library()
    w = _f32(costs).reshape(-1).copy()
    _op_check(lib, lib.mppi_norm_exp(w, w.size, lambda_inv, baseline, device))
    return w


def compute_weights(costs, lambda_inv, device=0):
    lib = load_library()
    w = _f32(costs).reshape(-1).copy()
    out = np.zeros(2, np.float32)
    _op_check(lib, lib.mppi_compute_weights(w, w.size, lambda_inv, out, device))
    return w, float(out[0]), float(out[1])


def compute_weights_reference_order(costs, lambda_, device=0):
    """weights, stats8 = {rho, eta, free energy mean / variance / modified variance, sum w^2, 0, 0} in the reference's order"""
    lib = load_library()
    w = _f32(costs).reshape(-1).copy()
    st = np.zeros(8, np.float32)
    _op_check(lib, lib.mppi_compute_weights_reference_order(w, w.size, lambda_, st, device))
    return w, st


def weighted_reduction_reference_order(weights, v, normalizer, sum_stride=32, fma=False, device=0):
    lib = load_library()
    v = _f32(v)
    K, T, Cd = v.shape
    u = np.empty((T, Cd), np.float32)
    _op_check(lib, lib.mppi_weighted_reduction_reference_order(_f32(weights).reshape(-1), v, normalizer, K, T, Cd,
                                                               sum_stride, 1 if fma else 0, u, device))
    return u


def weighted_reduction(weights, v, normalizer, device=0):
    lib = load_library()
    v = _f32(v)
    K, T, Cd = v.shape
    u = np.empty((T, Cd), np.float32)
    _op_check(lib, lib.mppi_weighted_reduction(_f32(weights).reshape(-1), v, normalizer, K, T, Cd, u, device))
    return u
