/* forwarding header: the reference's include path (include/mppi/utils/nn_helpers/fnn_helper.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_UTILS_NN_HELPERS_FNN_HELPER_CUH
#define MPPI_FWD_UTILS_NN_HELPERS_FNN_HELPER_CUH
#include "mppi_amd/utils/nn_helpers/fnn_helper.hpp"
#endif
