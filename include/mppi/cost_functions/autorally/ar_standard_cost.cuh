/* forwarding header: the reference's include path (include/mppi/cost_functions/autorally/ar_standard_cost.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_COST_FUNCTIONS_AUTORALLY_AR_STANDARD_COST_CUH
#define MPPI_FWD_COST_FUNCTIONS_AUTORALLY_AR_STANDARD_COST_CUH
#include "mppi_amd/cost_functions/autorally/ar_standard_cost.hpp"
#endif
