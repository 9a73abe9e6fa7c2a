/* forwarding header: the reference's include path (include/mppi/dynamics/dynamics.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_DYNAMICS_DYNAMICS_CUH
#define MPPI_FWD_DYNAMICS_DYNAMICS_CUH
#include "mppi_amd/plugin/dynamics.hpp"
#endif
