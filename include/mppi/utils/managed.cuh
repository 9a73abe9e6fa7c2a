/* forwarding header: the reference's include path (include/mppi/utils/managed.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_UTILS_MANAGED_CUH
#define MPPI_FWD_UTILS_MANAGED_CUH
#include "mppi_amd/plugin/managed.hpp"
#endif
