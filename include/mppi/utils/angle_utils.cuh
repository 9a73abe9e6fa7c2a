/* forwarding header: the reference's include path (include/mppi/utils/angle_utils.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_UTILS_ANGLE_UTILS_CUH
#define MPPI_FWD_UTILS_ANGLE_UTILS_CUH
#include "mppi_amd/plugin/math_utils.hpp"
#endif
