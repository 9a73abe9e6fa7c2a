"""ctypes declarations for include/mppi_amd.h (the C ABI of libmppi_amd.so).

The library is the product; this file only declares its entry points.  There is no Python or CPU fallback: if the
library is missing it is built (hipcc), and if no HIP device is visible mppi_create() fails with MPPI_ERR_NO_DEVICE.
"""
import ctypes as C
import os

import numpy as np

from . import buildlib as _build

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


class MppiConfig(C.Structure):
    """struct mppi_config"""
    _fields_ = [
        ("model", C.c_char_p),
        ("controller", C.c_int),
        ("num_rollouts", C.c_int),
        ("num_timesteps", C.c_int),
        ("dt", C.c_float),
        ("lambda_", C.c_float),
        ("alpha", C.c_float),
        ("num_iters", C.c_int),
        ("seed", C.c_uint64),
        ("noise_source", C.c_int),
        ("block_x", C.c_int),
        ("block_y", C.c_int),
        ("device", C.c_int),
        ("stream", C.c_void_p),
        ("rank", C.c_int),
        ("world_size", C.c_int),
        ("save_samples", C.c_int),
        ("kernel_variant", C.c_int),
        ("force_exchange", C.c_int),
    ]


class MppiGaussianParams(C.Structure):
    _fields_ = [
        ("std_dev", C.POINTER(C.c_float)),
        ("control_cost_coeff", C.POINTER(C.c_float)),
        ("pure_noise_trajectories_percentage", C.c_float),
        ("std_dev_decay", C.c_float),
        ("sum_strides", C.c_int),
    ]


class MppiSystemStats(C.Structure):
    _fields_ = [
        ("baseline", C.c_float),
        ("normalizer", C.c_float),
        ("free_energy_mean", C.c_float),
        ("free_energy_variance", C.c_float),
        ("free_energy_modified_variance", C.c_float),
    ]


class MppiStats(C.Structure):
    _fields_ = [("real_sys", MppiSystemStats), ("nominal_sys", MppiSystemStats), ("nominal_state_used", C.c_int)]


# every symbol include/mppi_amd.h declares: name -> (restype, argtypes)
H = C.c_void_p
SIGNATURES = {
    "mppi_version": (C.c_char_p, []),
    "mppi_status_string": (C.c_char_p, [C.c_int]),
    "mppi_device_count": (C.c_int, []),
    "mppi_list_models": (C.c_char_p, []),
    "mppi_source_hash": (C.c_char_p, []),
    "mppi_register_model": (C.c_int, [C.c_char_p, C.c_int, C.c_void_p, C.c_int]),
    "mppi_register_model_checked": (C.c_int, [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_uint]),
    "mppi_load_plugin": (C.c_int, [C.c_char_p]),
    "mppi_create": (C.c_int, [C.POINTER(MppiConfig), C.POINTER(H)]),
    "mppi_destroy": (None, [H]),
    "mppi_last_error": (C.c_char_p, [H]),
    "mppi_get_dims": (C.c_int, [H] + [C.POINTER(C.c_int)] * 4),
    "mppi_get_local_rollouts": (C.c_int, [H, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mppi_get_launch_counts": (C.c_int, [H, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "mppi_debug_host_stamps": (C.c_int, [H, C.POINTER(C.c_double)]),
    "mppi_set_dynamics_params": (C.c_int, [H, C.c_void_p, C.c_size_t]),
    "mppi_set_cost_params": (C.c_int, [H, C.c_void_p, C.c_size_t]),
    "mppi_set_sampler_params": (C.c_int, [H, C.POINTER(MppiGaussianParams)]),
    "mppi_set_independent_noise": (C.c_int, [H, C.c_int]),
    "mppi_set_time_specific_std_dev": (C.c_int, [H, C.c_void_p]),
    "mppi_set_colored_noise_params": (C.c_int, [H, _f32p, C.c_float, C.c_float]),
    "mppi_sample_noise": (C.c_int, [H, C.c_int, _f32p]),
    "mppi_set_rmppi_params": (C.c_int, [H, C.c_float, C.c_int, C.c_int]),
    "mppi_set_feedback_gains": (C.c_int, [H, _f32p, C.c_int]),
    "mppi_update_importance_sampling_control": (C.c_int, [H, _f32p, C.c_int]),
    "mppi_get_rmppi_state": (C.c_int, [H, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]),
    "mppi_set_colored_mppi_params": (C.c_int, [H, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int]),
    "mppi_set_control_ranges": (C.c_int, [H, _f32p]),
    "mppi_set_control_deadband": (C.c_int, [H, _f32p]),
    "mppi_set_lambda_alpha": (C.c_int, [H, C.c_float, C.c_float]),
    "mppi_set_num_iters": (C.c_int, [H, C.c_int]),
    "mppi_set_reduction_mode": (C.c_int, [H, C.c_int]),
    "mppi_set_slide_control_scale": (C.c_int, [H, _f32p]),
    "mppi_set_nominal_threshold": (C.c_int, [H, C.c_float]),
    "mppi_set_model_blob": (C.c_int, [H, C.c_char_p, _f32p, C.c_size_t, C.POINTER(C.c_int), C.c_int]),
    "mppi_set_lstm_initial_state": (C.c_int, [H, _f32p, _f32p]),
    "mppi_lstm_lstm_initialize": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, _f32p, _f32p, C.c_int, C.c_int,
                                           _f32p, C.c_int, _f32p]),
    "mppi_load_npz": (C.c_int, [H, C.c_char_p, C.c_char_p, C.c_char_p]),
    "mppi_npz_read_array": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mppi_set_seed": (C.c_int, [H, C.c_uint64]),
    "mppi_set_nominal_control": (C.c_int, [H, _f32p]),
    "mppi_inject_noise": (C.c_int, [H, C.c_void_p, C.c_int]),
    "mppi_compute_control": (C.c_int, [H, _f32p, C.c_int]),
    "mppi_get_control_seq": (C.c_int, [H, _f32p]),
    "mppi_get_state_seq": (C.c_int, [H, _f32p]),
    "mppi_get_output_seq": (C.c_int, [H, _f32p]),
    "mppi_get_nominal_control_seq": (C.c_int, [H, _f32p]),
    "mppi_get_nominal_state_seq": (C.c_int, [H, _f32p]),
    "mppi_slide": (C.c_int, [H, C.c_int]),
    "mppi_get_costs": (C.c_int, [H, _f32p]),
    "mppi_get_stats": (C.c_int, [H, C.POINTER(MppiStats)]),
    "mppi_get_sampled_controls": (C.c_int, [H, _f32p]),
    "mppi_get_optimal_control": (C.c_int, [H, _f32p]),
    "mppi_optimize": (C.c_int, [H, C.c_int, C.c_int]),
    "mppi_upload_state": (C.c_int, [H, _f32p]),
    "mppi_time_iterations": (C.c_int, [H, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mppi_choose_kernel": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mppi_synchronize": (C.c_int, [H]),
    "mppi_get_exchange_buffers": (C.c_int, [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "mppi_read_send_record": (C.c_int, [H, _f32p]),
    "mppi_write_recv_records": (C.c_int, [H, _f32p]),
    "mppi_iteration_local": (C.c_int, [H]),
    "mppi_iteration_merge": (C.c_int, [H]),
    "mppi_p2p_mailbox_handle": (C.c_int, [H, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mppi_p2p_reset": (C.c_int, [H]),
    "mppi_p2p_connect": (C.c_int, [H, C.c_void_p, C.c_size_t]),
    "mppi_p2p_connect_local": (C.c_int, [H, C.POINTER(C.c_void_p)]),
    "mppi_rccl_unique_id": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mppi_comm_init_rccl": (C.c_int, [H, C.c_void_p, C.c_size_t]),
    "mppi_rollout_costs": (C.c_int, [H, _f32p, C.c_int]),
    "mppi_enforce_constraints": (C.c_int, [H, C.c_void_p, _f32p]),
    "mppi_model_step": (C.c_int, [H, _f32p, _f32p, C.c_float, C.c_int]),
    "mppi_norm_exp": (C.c_int, [_f32p, C.c_int, C.c_float, C.c_float, C.c_int]),
    "mppi_compute_weights": (C.c_int, [_f32p, C.c_int, C.c_float, _f32p, C.c_int]),
    "mppi_weighted_reduction": (C.c_int, [_f32p, _f32p, C.c_float, C.c_int, C.c_int, C.c_int, _f32p, C.c_int]),
    "mppi_compute_weights_reference_order": (C.c_int, [_f32p, C.c_int, C.c_float, _f32p, C.c_int]),
    "mppi_weighted_reduction_reference_order": (C.c_int, [_f32p, _f32p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                                           C.c_int, _f32p, C.c_int]),
    "mppi_philox_normal": (C.c_int, [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int]),
    "mppi_measure_launch_boundary": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "mppi_measure_issue_interval": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "mppi_det_eval": (C.c_int, [C.c_int, _f32p, _f32p, C.c_int, C.c_int]),
    "mppi_texture2d_query": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, _f32p, C.c_int, C.c_int, _f32p, C.c_int]),
}

class MppiTexture2dParams(C.Structure):
    """mppi_texture2d_params"""
    _fields_ = [("address_mode", C.c_int * 2), ("filter_mode", C.c_int), ("border_color", C.c_float * 4),
                ("origin", C.c_float * 3), ("rotations", C.c_float * 9), ("resolution", C.c_float * 3)]

    def __init__(self, origin=(0, 0, 0), rotations=(1, 0, 0, 0, 1, 0, 0, 0, 1), resolution=(1, 1, 1), address_mode=(0, 0),
                 filter_mode=0, border_color=(0, 0, 0, 0)):
        super().__init__()
        self.origin[:] = origin
        self.rotations[:] = rotations
        self.resolution[:] = resolution
        self.address_mode[:] = address_mode
        self.filter_mode = filter_mode
        self.border_color[:] = border_color


_lib = None


def library_path():
    """the library load_library() opens: the in-tree build, or the experimental build MPPI_AMD_LIB names (tools/ A/B runs)"""
    return os.environ.get("MPPI_AMD_LIB") or _build.LIB


def load_library(build_if_missing=True):
    """dlopen libmppi_amd.so (building it first if needed) and attach the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MPPI_AMD_LIB") or _build.LIB  # MPPI_AMD_LIB: an experimental build (tools/ A/B runs only)
    if path == _build.LIB and build_if_missing and _build.needs_build():
        _build.build()
    if not os.path.exists(path):
        raise RuntimeError(
            "libmppi_amd.so is missing (%s); run `python mppi-generic_amd/buildlib.py` — there is no fallback path" % path)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)  # plugins (mppi_load_plugin) resolve mppi_register_model against it
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
