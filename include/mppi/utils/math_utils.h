/* forwarding header: the reference's include path (include/mppi/utils/math_utils.h) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_UTILS_MATH_UTILS_H
#define MPPI_FWD_UTILS_MATH_UTILS_H
#include "mppi_amd/plugin/math_utils.hpp"
#endif
