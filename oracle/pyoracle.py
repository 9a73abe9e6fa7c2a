"""ctypes loader for the CPU oracle (oracle/_build/libmppi_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmppi_oracle.so")
if os.environ.get("MPPI_ORACLE_LIB"):  # study builds only (oracle/Makefile target libm): a measuring stick, not a checker
    LIB = os.environ["MPPI_ORACLE_LIB"]
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_lib = None


def build(force=False):
    if os.environ.get("MPPI_ORACLE_LIB"):
        return LIB
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".cpp", ".hpp"))]
    srcs += [os.path.join(HERE, "..", "include", "mppi_amd", f) for f in ("det_math.h", "model_params.h")]
    srcs.append(os.path.join(HERE, "Makefile"))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        r = subprocess.run(["make", "-C", HERE, "-B" if force else "-s"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        L.oracle_set_dynamics_params.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_set_cost_params.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_set_blob.argtypes = [C.c_void_p, C.c_char_p, _f32p, C.c_size_t, C.POINTER(C.c_int), C.c_int]
        L.oracle_fnn_forward.argtypes = [C.POINTER(C.c_int), C.c_int, _f32p, _f32p, _f32p]
        L.oracle_fnn_forward2.argtypes = [C.POINTER(C.c_int), C.c_int, _f32p, _f32p, _f32p, C.c_int]
        L.oracle_lstm_forward2.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, _f32p, _f32p, _f32p, C.c_int,
                                           _f32p, C.c_int]
        L.oracle_set_split_output_sum.argtypes = [C.c_void_p, C.c_int]
        L.oracle_lstm_forward.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, _f32p, _f32p, _f32p, C.c_int,
                                          _f32p]
        L.oracle_state_deriv.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
        L.oracle_texture2d_query.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, _f32p, _f32p, _f32p,
                                             _f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.oracle_update_state.argtypes = [C.c_void_p, _f32p, _f32p, C.c_float, _f32p]
        L.oracle_state_cost.restype = C.c_float
        L.oracle_state_cost.argtypes = [C.c_void_p, _f32p, C.c_int, C.POINTER(C.c_int)]
        L.oracle_ar_cost_term.restype = C.c_float
        L.oracle_ar_cost_term.argtypes = [C.c_void_p, C.c_int, _f32p, C.POINTER(C.c_int)]
        L.oracle_ar_coor_transform.argtypes = [C.c_void_p, C.c_float, C.c_float, _f32p]
        L.oracle_set_colored_mppi_params.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int]
        L.oracle_set_control_ranges.argtypes = [C.c_void_p, _f32p]
        L.oracle_set_control_deadband.argtypes = [C.c_void_p, _f32p]
        L.oracle_set_sampler.argtypes = [C.c_void_p, _f32p, _f32p, C.c_float, C.c_float, C.c_int]
        L.oracle_set_independent_noise.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_time_specific_std_dev.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_set_controller_params.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        L.oracle_set_gaussian_controls.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.oracle_rollout_costs.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, C.c_int]
        L.oracle_best_index.restype = C.c_int
        L.oracle_best_index.argtypes = [_f32p, C.c_int]
        L.oracle_baseline.restype = C.c_float
        L.oracle_baseline.argtypes = [_f32p, C.c_int]
        L.oracle_norm_exp.argtypes = [_f32p, C.c_int, C.c_float, C.c_float]
        L.oracle_normalizer.restype = C.c_float
        L.oracle_normalizer.argtypes = [_f32p, C.c_int]
        L.oracle_free_energy.argtypes = [_f32p, C.c_int, C.c_float, C.c_float, _f32p]
        L.oracle_weighted_reduction.argtypes = [_f32p, _f32p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.oracle_smooth.argtypes = [_f32p, _f32p, C.c_int, C.c_int]
        L.oracle_slide.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p]
        L.oracle_save_history.argtypes = [C.c_int, _f32p, _f32p, C.c_int]
        L.oracle_state_trajectory.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
        L.oracle_output_trajectory.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p]
        L.oracle_model_step.argtypes = [C.c_void_p, _f32p, _f32p, C.c_float]
        L.oracle_enforce_leash.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p]
        L.oracle_model_step_full.argtypes = [C.c_void_p, _f32p, _f32p, C.c_float, _f32p, _f32p, _f32p]
        L.oracle_set_nominal_control.argtypes = [C.c_void_p, _f32p]
        L.oracle_iterate.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.oracle_vanilla_compute_control.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p]
        L.oracle_tube_compute_control.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p]
        L.oracle_vanilla_slide.argtypes = [C.c_void_p, C.c_int]
        L.oracle_tube_slide.argtypes = [C.c_void_p, C.c_int]
        for n in ("control", "nominal_control", "state_traj", "nominal_state_traj", "costs", "weights", "samples",
                  "stats"):
            getattr(L, "oracle_get_" + n).argtypes = [C.c_void_p, _f32p]
        L.oracle_colored_noise.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_float, C.c_float, C.c_int,
                                           _f32p, _f32p]
        L.oracle_colored_weights.argtypes = [C.c_int, C.c_int, _f32p, C.c_float, _f32p, _f32p]
        L.oracle_philox_spectrum.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.oracle_colored_compute_control.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, _f32p, C.c_float, C.c_float]
        _i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        L.oracle_rmppi_create.restype = C.c_void_p
        L.oracle_rmppi_create.argtypes = [C.c_void_p]
        L.oracle_rmppi_destroy.argtypes = [C.c_void_p]
        L.oracle_rmppi_set_params.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
        L.oracle_rmppi_set_gains.argtypes = [C.c_void_p, _f32p, C.c_int]
        L.oracle_rmppi_feedback.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, _f32p]
        L.oracle_rmppi_line_search.argtypes = [C.c_void_p, C.c_int, _f32p, _i32p]
        L.oracle_rmppi_candidates.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p]
        L.oracle_rmppi_best_index.restype = C.c_int
        L.oracle_rmppi_best_index.argtypes = [C.c_void_p, _f32p, _f32p]
        L.oracle_rmppi_rollout_costs.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p]
        L.oracle_rmppi_update_importance_sampling.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p]
        L.oracle_rmppi_compute_control.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p]
        L.oracle_rmppi_get_state.argtypes = [C.c_void_p, _f32p, _i32p, _f32p, C.c_void_p]
        L.oracle_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        L.oracle_philox_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, _f32p]
        L.oracle_det_eval.argtypes = [C.c_int, _f32p, _f32p, C.c_int]
        L.oracle_time_iterations.restype = C.c_double
        L.oracle_time_iterations.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_int]
        L.oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """One (model, K, T, D) oracle instance: kernel-level pieces and the controller loops."""

    def __init__(self, model, K, T, D=1, dt=0.01, lambda_=1.0, alpha=0.0, num_iters=1):
        self.L = lib()
        self.h = self.L.oracle_create(model.encode(), K, T, D, dt, lambda_, alpha, num_iters)
        if not self.h:
            raise ValueError("oracle: unknown model " + model)
        s, c, o = C.c_int(), C.c_int(), C.c_int()
        self.L.oracle_dims(self.h, C.byref(s), C.byref(c), C.byref(o))
        self.S, self.C, self.O = s.value, c.value, o.value
        self.K, self.T, self.D = K, T, D
        self.dt, self.lambda_, self.alpha, self.num_iters = dt, lambda_, alpha, num_iters

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_destroy(self.h)
            self.h = None

    def set_dynamics_params(self, pod):
        assert self.L.oracle_set_dynamics_params(self.h, C.byref(pod), C.sizeof(pod)) == 0

    def set_cost_params(self, pod):
        assert self.L.oracle_set_cost_params(self.h, C.byref(pod), C.sizeof(pod)) == 0

    def set_split_output_sum(self, on):
        """AutoRally-NN / bicycle-LSTM models: the output layer's summation order (default: split, what the engine evaluates)"""
        assert self.L.oracle_set_split_output_sum(self.h, int(on)) == 0

    def set_blob(self, name, array):
        a = _f32(array)
        dims = (C.c_int * a.ndim)(*a.shape)
        assert self.L.oracle_set_blob(self.h, name.encode(), a.reshape(-1), a.size, dims, a.ndim) == 0, name

    def state_deriv(self, x, u):
        out = np.zeros(self.S, np.float32)
        self.L.oracle_state_deriv(self.h, _f32(x).reshape(-1), _f32(u).reshape(-1), out)
        return out

    def update_state(self, x, xdot, dt):
        """Dynamics::updateState with a given derivative"""
        xn = np.zeros(self.S, np.float32)
        self.L.oracle_update_state(self.h, _f32(x).reshape(-1), _f32(xdot).reshape(-1), dt, xn)
        return xn

    def state_cost(self, y, t=0, crash=0):
        cr = C.c_int(crash)
        v = self.L.oracle_state_cost(self.h, _f32(y).reshape(-1), t, C.byref(cr))
        return float(v), cr.value

    def ar_cost_term(self, which, s, crash=0):
        """one term of ARStandardCost: which in ("speed", "stabilizing", "track", "crash") -> (value, crash flag)"""
        cr = C.c_int(crash)
        v = self.L.oracle_ar_cost_term(self.h, ("speed", "stabilizing", "track", "crash").index(which),
                                       _f32(s).reshape(-1), C.byref(cr))
        return float(v), cr.value

    def ar_coor_transform(self, x, y):
        out = np.empty(3, np.float32)
        assert self.L.oracle_ar_coor_transform(self.h, x, y, out) == 0
        return out

    def set_colored_mppi_params(self, gamma=0.0, r_exp=0.0, leash_dist=None, leash_active=False, leash_jump=1):
        p = None if leash_dist is None else _f32(leash_dist).reshape(-1).ctypes.data
        self.L.oracle_set_colored_mppi_params(self.h, gamma, r_exp, p, int(leash_active), leash_jump)

    def set_control_ranges(self, lo_hi):
        self.L.oracle_set_control_ranges(self.h, _f32(lo_hi).reshape(-1))

    def set_control_deadband(self, db):
        self.L.oracle_set_control_deadband(self.h, _f32(db).reshape(-1))

    def set_sampler(self, std_dev, control_cost_coeff=None, pure_noise_pct=0.01, std_dev_decay=1.0, sum_strides=32):
        sd = _f32(std_dev).reshape(-1)
        if sd.size == self.C and self.D == 2:
            sd = np.concatenate([sd, sd])
        cc = _f32(np.zeros(self.C) if control_cost_coeff is None else control_cost_coeff).reshape(-1)
        self.L.oracle_set_sampler(self.h, sd, cc, pure_noise_pct, std_dev_decay, sum_strides)

    def set_independent_noise(self, independent=True):
        """eps then is [D][K][T][C] (use_same_noise_for_all_distributions = false, gaussian.cu:378-394)"""
        self.L.oracle_set_independent_noise(self.h, int(independent))

    def set_time_specific_std_dev(self, std_dev):
        """std_dev [D][T][C] or None (gaussian.cuh:64-95 GaussianTimeVaryingStdDevParams)"""
        p = None if std_dev is None else _f32(std_dev).reshape(-1).ctypes.data
        self.L.oracle_set_time_specific_std_dev(self.h, p)

    def set_controller_params(self, nominal_threshold=20.0, slide_scale=None):
        p = None if slide_scale is None else _f32(slide_scale).ctypes.data
        self.L.oracle_set_controller_params(self.h, nominal_threshold, p)

    # kernel-level
    def set_gaussian_controls(self, mean, eps, stride=1, iteration=0):
        v = np.empty((self.D, self.K, self.T, self.C), np.float32)
        self.L.oracle_set_gaussian_controls(self.h, _f32(mean).reshape(-1), _f32(eps).reshape(-1), stride, iteration, v)
        return v

    def rollout_costs(self, x0, mean, v, threads=1):
        v = _f32(v).copy()
        costs = np.empty((self.D, self.K), np.float32)
        self.L.oracle_rollout_costs(self.h, _f32(x0).reshape(-1), _f32(mean).reshape(-1), v, costs, threads)
        return costs, v

    def iterate(self, x0, mean, eps, stride=1, iteration=0):
        u = np.empty((self.D, self.T, self.C), np.float32)
        self.L.oracle_iterate(self.h, _f32(x0).reshape(-1), _f32(mean).reshape(-1), _f32(eps).reshape(-1), stride,
                              iteration, u)
        return u

    def state_trajectory(self, x0, u):
        out = np.empty((self.T, self.S), np.float32)
        self.L.oracle_state_trajectory(self.h, _f32(x0).reshape(-1), _f32(u).reshape(-1), out)
        return out

    def output_trajectory(self, x0, u):
        """computeOutputTrajectoryHelper: (state [T][S], output [T][O])"""
        xs = np.empty((self.T, self.S), np.float32)
        ys = np.empty((self.T, self.O), np.float32)
        self.L.oracle_output_trajectory(self.h, _f32(x0).reshape(-1), _f32(u).reshape(-1), xs, ys)
        return xs, ys

    def model_step(self, x, u, dt=None):
        x = _f32(x).reshape(-1).copy()
        u = _f32(u).reshape(-1).copy()
        self.L.oracle_model_step(self.h, x, u, self.dt if dt is None else dt)
        return x, u

    def enforce_leash(self, x_true, x_nominal, leash):
        out = np.zeros(self.S, np.float32)
        self.L.oracle_enforce_leash(self.h, _f32(x_true).reshape(-1), _f32(x_nominal).reshape(-1), _f32(leash).reshape(-1), out)
        return out

    def model_step_full(self, x, u, dt=None):
        """initializeDynamics + one step(): (next state, state derivative, output)"""
        xn, xd, y = np.zeros(self.S, np.float32), np.zeros(self.S, np.float32), np.zeros(self.O, np.float32)
        self.L.oracle_model_step_full(self.h, _f32(x).reshape(-1), _f32(u).reshape(-1), self.dt if dt is None else dt, xn, xd, y)
        return xn, xd, y

    # controller level
    def set_nominal_control(self, u):
        self.L.oracle_set_nominal_control(self.h, _f32(u).reshape(-1))

    def vanilla_compute_control(self, x0, stride, eps):
        self.L.oracle_vanilla_compute_control(self.h, _f32(x0).reshape(-1), stride, _f32(eps).reshape(-1))

    def colored_compute_control(self, x0, stride, z, exponents, offset_decay_rate=0.97, fmin=0.0):
        """ColoredMPPI loop; z [num_iters][K][C][T+1][2] Gaussian spectrum"""
        self.L.oracle_colored_compute_control(self.h, _f32(x0).reshape(-1), stride, _f32(z).reshape(-1),
                                              _f32(exponents).reshape(-1), offset_decay_rate, fmin)

    def tube_compute_control(self, x0, stride, eps):
        self.L.oracle_tube_compute_control(self.h, _f32(x0).reshape(-1), stride, _f32(eps).reshape(-1))

    def vanilla_slide(self, steps):
        self.L.oracle_vanilla_slide(self.h, steps)

    def tube_slide(self, steps):
        self.L.oracle_tube_slide(self.h, steps)

    def _get(self, name, shape):
        out = np.empty(shape, np.float32)
        getattr(self.L, "oracle_get_" + name)(self.h, out)
        return out

    def control(self):
        return self._get("control", (self.T, self.C))

    def nominal_control(self):
        return self._get("nominal_control", (self.T, self.C))

    def state_traj(self):
        return self._get("state_traj", (self.T, self.S))

    def nominal_state_traj(self):
        return self._get("nominal_state_traj", (self.T, self.S))

    def costs(self):
        return self._get("costs", (self.D, self.K))

    def weights(self):
        return self._get("weights", (self.D, self.K))

    def samples(self):
        return self._get("samples", (self.D, self.K, self.T, self.C))

    def stats(self):
        s = self._get("stats", (11,))
        return {"baseline": s[0:2], "normalizer": s[2:4], "free_energy": s[4:6], "free_energy_var": s[6:8],
                "free_energy_mod": s[8:10], "nominal_state_used": int(s[10])}

    def time_iterations(self, x0, mean, eps, iters, threads):
        return self.L.oracle_time_iterations(self.h, _f32(x0).reshape(-1), _f32(mean).reshape(-1),
                                             _f32(eps).reshape(-1), iters, threads)


# free functions
def philox4x32_10(ctr, key):
    out = np.zeros(4, np.uint32)
    lib().oracle_philox4x32_10(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
    return out


def philox_normal(seed, generation, K, T, Cdim, k_begin=0, k_end=None, stream=0):
    k_end = K if k_end is None else k_end
    out = np.empty((k_end - k_begin, T, Cdim), np.float32)
    lib().oracle_philox_normal(seed, generation, stream, K, T, Cdim, k_begin, k_end, out)
    return out


def fnn_forward(layers, theta, x, split_output_sum=False):
    """split_output_sum: the output layer's summation order of the matrix-core networks (oracle_models.hpp: FNN)"""
    layers = list(layers)
    arr = (C.c_int * len(layers))(*layers)
    out = np.zeros(layers[-1], np.float32)
    lib().oracle_fnn_forward2(arr, len(layers), _f32(theta).reshape(-1), _f32(x).reshape(-1), out, int(split_output_sum))
    return out


def lstm_forward(input_dim, hidden_dim, out_layers, lstm_blob, fnn_blob, inputs, split_output_sum=False):
    """inputs [steps][input_dim] -> outputs [steps][out_layers[-1]], state carried from the blob's (h0, c0)"""
    out_layers = list(out_layers)
    arr = (C.c_int * len(out_layers))(*out_layers)
    x = _f32(inputs).reshape(-1, input_dim)
    out = np.zeros((x.shape[0], out_layers[-1]), np.float32)
    lib().oracle_lstm_forward2(input_dim, hidden_dim, arr, len(out_layers), _f32(lstm_blob).reshape(-1),
                               _f32(fnn_blob).reshape(-1), x, x.shape[0], out, int(split_output_sum))
    return out


class RobustOracle:
    """RobustMPPIController logic on top of an Oracle created with D = 2 (system 0 nominal, 1 real)"""

    def __init__(self, oracle, value_function_threshold=1000.0, num_candidates=9, samples_per_candidate=32):
        assert oracle.D == 2
        self.o, self.L = oracle, oracle.L
        self.r = self.L.oracle_rmppi_create(oracle.h)
        self.nc, self.ns = num_candidates, samples_per_candidate
        self.L.oracle_rmppi_set_params(self.r, value_function_threshold, num_candidates, samples_per_candidate)

    def __del__(self):
        if getattr(self, "r", None):
            self.L.oracle_rmppi_destroy(self.r)
            self.r = None

    def set_gains(self, gains, accumulate_all_states=False):
        g = _f32(gains)
        assert g.shape == (self.o.T, self.o.S, self.o.C)
        self.L.oracle_rmppi_set_gains(self.r, g.reshape(-1), int(accumulate_all_states))

    def feedback(self, x_act, x_goal, t):
        out = np.zeros(self.o.C, np.float32)
        self.L.oracle_rmppi_feedback(self.r, _f32(x_act).reshape(-1), _f32(x_goal).reshape(-1), t, out)
        return out

    def line_search(self, stride):
        w = np.zeros((3, self.nc), np.float32)
        s = np.zeros(self.nc, np.int32)
        self.L.oracle_rmppi_line_search(self.r, stride, w, s)
        return w, s

    def candidates(self, x_k, x_kp1, real_kp1):
        out = np.zeros((self.nc, self.o.S), np.float32)
        self.L.oracle_rmppi_candidates(self.r, _f32(x_k).reshape(-1), _f32(x_kp1).reshape(-1), _f32(real_kp1).reshape(-1), out)
        return out

    def best_index(self, candidate_costs):
        fe = np.zeros(self.nc, np.float32)
        c = _f32(candidate_costs).reshape(-1)
        assert c.size == self.nc * self.ns
        return self.L.oracle_rmppi_best_index(self.r, c, fe), fe

    def rollout_costs(self, x0, mean, v):
        o = self.o
        v = _f32(v).reshape(2, o.K, o.T, o.C).copy()
        costs = np.zeros((2, o.K), np.float32)
        self.L.oracle_rmppi_rollout_costs(self.r, _f32(x0).reshape(-1), _f32(mean).reshape(-1), v, costs)
        return costs, v

    def update_importance_sampling(self, state, stride, eps=None):
        o = self.o
        e = np.zeros((o.K, o.T, o.C), np.float32) if eps is None else _f32(eps).reshape(o.K, o.T, o.C)
        self.L.oracle_rmppi_update_importance_sampling(self.r, _f32(state).reshape(-1), stride, e)

    def compute_control(self, state, stride, eps):
        self.L.oracle_rmppi_compute_control(self.r, _f32(state).reshape(-1), stride, _f32(eps).reshape(-1))

    def state(self, with_costs=False):
        ns = np.zeros(self.o.S, np.float32)
        bs = np.zeros(2, np.int32)
        fe = np.zeros(self.nc, np.float32)
        self.L.oracle_rmppi_get_state(self.r, ns, bs, fe, None)
        return ns, int(bs[0]), int(bs[1]), fe


def colored_noise(z, exponents, offset_decay_rate=0.97, fmin=0.0, offset_t=1, flavour="gemm"):
    """z [K][C][T+1][2] Gaussian spectrum -> eps [K][T][C]; flavour "definition" (double inverse DFT, the reference's
    pipeline step by step), "gemm" / "engine" (the engine's arithmetic: radix-4 butterfly + quarter-size GEMM where T is a
    multiple of 4, the dense folded table otherwise) or "dense" (the dense folded table + fp32 fma chains for any T)"""
    z = _f32(z)
    K, Cd, F, _ = z.shape
    T = F - 1
    eps = np.zeros((K, T, Cd), np.float32)
    lib().oracle_colored_noise({"definition": 0, "gemm": 1, "engine": 1, "dense": 2}[flavour], K, T, Cd, _f32(exponents).reshape(-1),
                               offset_decay_rate, fmin, offset_t, z.reshape(-1), eps)
    return eps


def colored_weights(T, exponents, fmin=0.0):
    e = _f32(exponents).reshape(-1)
    w = np.zeros((e.size, T + 1), np.float32)
    sigma = np.zeros(e.size, np.float32)
    lib().oracle_colored_weights(T, e.size, e, fmin, w, sigma)
    return w, sigma


def philox_spectrum(seed, generation, K, T, Cd, k_begin=0, k_end=None):
    k_end = K if k_end is None else k_end
    z = np.zeros((k_end - k_begin, Cd, T + 1, 2), np.float32)
    lib().oracle_philox_spectrum(seed, generation, T, Cd, k_begin, k_end, z)
    return z


def texture2d_query(data, points, frame, origin=(0, 0, 0), rotations=(1, 0, 0, 0, 1, 0, 0, 0, 1), resolution=(1, 1, 1),
                    address_mode=(0, 0), filter_mode=0, border_color=(0, 0, 0, 0)):
    """data[h][w] or [h][w][channels]; points[n][3]; frame 0 texture coordinate / 1 map pose / 2 world pose"""
    d = _f32(data)
    if d.ndim == 2:
        d = d[:, :, None]
    h, w, ch = d.shape
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros((pts.shape[0], ch), np.float32)
    am = (C.c_int * 2)(*address_mode)
    lib().oracle_texture2d_query(d.reshape(-1), w, h, ch, am, filter_mode, _f32(border_color), _f32(origin),
                                 _f32(rotations).reshape(-1), _f32(resolution), pts.reshape(-1), pts.shape[0], frame,
                                 out.reshape(-1))
    return out


def det_eval(func, x):
    x = _f32(x).reshape(-1)
    y = np.empty_like(x)
    lib().oracle_det_eval(func, x, y, x.size)
    return y


def baseline(costs):
    c = _f32(costs).reshape(-1)
    return float(lib().oracle_baseline(c, c.size))


def best_index(costs):
    c = _f32(costs).reshape(-1)
    return int(lib().oracle_best_index(c, c.size))


def norm_exp(costs, lambda_inv, base):
    w = _f32(costs).reshape(-1).copy()
    lib().oracle_norm_exp(w, w.size, lambda_inv, base)
    return w


def normalizer(w):
    w = _f32(w).reshape(-1)
    return float(lib().oracle_normalizer(w, w.size))


def free_energy(w, base, lambda_):
    w = _f32(w).reshape(-1)
    out = np.zeros(3, np.float32)
    lib().oracle_free_energy(w, w.size, base, lambda_, out)
    return out


def weighted_reduction(w, v, eta, sum_stride=32):
    v = _f32(v)
    K, T, Cd = v.shape
    u = np.empty((T, Cd), np.float32)
    lib().oracle_weighted_reduction(_f32(w).reshape(-1), v, eta, K, T, Cd, sum_stride, u)
    return u


def set_reduction_fma(fma):
    """process-wide: `inter += weight * v` of every weighted reduction as ONE fma (what nvcc's default -fmad=true makes of it on
    the reference's GPU path) instead of a rounded product followed by an addition (the default: the reference's CPU statement)"""
    lib().oracle_set_reduction_fma(1 if fma else 0)


def smooth(u, history):
    u = _f32(u).copy()
    T, Cd = u.shape
    lib().oracle_smooth(u, _f32(history).reshape(-1), T, Cd)
    return u


def slide(u, steps, zero_control=None, slide_scale=None):
    u = _f32(u).copy()
    T, Cd = u.shape
    z = _f32(np.zeros(Cd) if zero_control is None else zero_control)
    s = _f32(np.zeros(Cd) if slide_scale is None else slide_scale)
    lib().oracle_slide(u, T, Cd, steps, z, s)
    return u


def save_history(steps, u, history):
    h = _f32(history).copy()
    lib().oracle_save_history(steps, _f32(u).reshape(-1), h.reshape(-1), _f32(u).shape[1])
    return h


def max_threads():
    return lib().oracle_max_threads()
