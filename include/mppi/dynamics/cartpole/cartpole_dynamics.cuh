/* forwarding header: the reference's include path (include/mppi/dynamics/cartpole/cartpole_dynamics.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_DYNAMICS_CARTPOLE_CARTPOLE_DYNAMICS_CUH
#define MPPI_FWD_DYNAMICS_CARTPOLE_CARTPOLE_DYNAMICS_CUH
#include "mppi_amd/dynamics/cartpole/cartpole_dynamics.hpp"
#endif
