"""mppi-generic_amd — MI355X-native MPPI rollout-and-reduce engine (host-side Python mirror over the C ABI).

The product is ``lib/libmppi_amd.so`` (hand-written HIP kernels for gfx950 behind include/mppi_amd.h).  This package
only (a) builds it with hipcc, (b) declares its C entry points for ctypes, and (c) mirrors the reference's controller
classes (same method names) on top of those entry points so tests read like the reference's own.
"""
from . import buildlib as _buildlib
from .buildlib import build
from .capi import (MppiTexture2dParams, MppiConfig, MppiGaussianParams, MppiStats, MppiSystemStats, SIGNATURES, library_path, load_library)
from .controllers import (MPPI_KERNEL_AUTO, MPPI_KERNEL_FUSED, MPPI_KERNEL_PIPELINE, MPPI_CONTROLLER_TUBE, MPPI_CONTROLLER_VANILLA, MPPI_NOISE_INJECTED, MPPI_NOISE_PHILOX_FUSED,
                          MPPIError, MPPIController, TubeMPPIController, VanillaMPPIController, ColoredMPPIController, RobustMPPIController,
                          MPPI_CONTROLLER_COLORED, CartpoleDynamicsParams,
                          CartpoleQuadraticCostParams, DoubleIntegratorParams, DoubleIntegratorCircleCostParams,
                          ARStandardCostParams, RacerDubinsParams, RacerDubinsElevationParams, RacerDubinsSuspensionParams, RacerDubinsUncertaintyParams, QuadraticCostParams28, fnn_blob_from_npz_dict, lstm_blob_from_npz_dict,
                          det_eval, LSTMLSTMHelper, texture2d_query, npz_read_array, philox_normal, launch_boundary_us, issue_interval_ns, norm_exp, compute_weights, weighted_reduction,
                          compute_weights_reference_order, weighted_reduction_reference_order,
                          MPPI_REDUCTION_FUSED, MPPI_REDUCTION_REFERENCE_ORDER, MPPI_REDUCTION_REFERENCE_ORDER_FMA)
from .plant import BasePlant, BufferedPlantMixin, SimulatedPlant, interpolateControls, interpolateFeedback, interpolateState

__all__ = [
    "build", "load_library", "library_path", "MPPIError", "MPPIController", "VanillaMPPIController",
    "TubeMPPIController", "MppiConfig", "SIGNATURES", "BasePlant", "SimulatedPlant",
]
