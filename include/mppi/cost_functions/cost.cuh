/* forwarding header: the reference's include path (include/mppi/cost_functions/cost.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_COST_FUNCTIONS_COST_CUH
#define MPPI_FWD_COST_FUNCTIONS_COST_CUH
#include "mppi_amd/plugin/cost.hpp"
#endif
