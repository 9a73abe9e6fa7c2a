/* forwarding header: the reference's include path (include/mppi/cost_functions/double_integrator/double_integrator_robust_cost.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_COST_FUNCTIONS_DOUBLE_INTEGRATOR_DOUBLE_INTEGRATOR_ROBUST_COST_CUH
#define MPPI_FWD_COST_FUNCTIONS_DOUBLE_INTEGRATOR_DOUBLE_INTEGRATOR_ROBUST_COST_CUH
#include "mppi_amd/cost_functions/double_integrator/double_integrator_robust_cost.hpp"
#endif
