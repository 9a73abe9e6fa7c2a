/* forwarding header: the reference's include path (include/mppi/dynamics/double_integrator/di_dynamics.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_DYNAMICS_DOUBLE_INTEGRATOR_DI_DYNAMICS_CUH
#define MPPI_FWD_DYNAMICS_DOUBLE_INTEGRATOR_DI_DYNAMICS_CUH
#include "mppi_amd/dynamics/double_integrator/di_dynamics.hpp"
#endif
