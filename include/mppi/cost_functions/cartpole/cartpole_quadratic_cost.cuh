/* forwarding header: the reference's include path (include/mppi/cost_functions/cartpole/cartpole_quadratic_cost.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_COST_FUNCTIONS_CARTPOLE_CARTPOLE_QUADRATIC_COST_CUH
#define MPPI_FWD_COST_FUNCTIONS_CARTPOLE_CARTPOLE_QUADRATIC_COST_CUH
#include "mppi_amd/cost_functions/cartpole/cartpole_quadratic_cost.hpp"
#endif
