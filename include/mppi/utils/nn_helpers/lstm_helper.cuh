/* forwarding header: the reference's include path (include/mppi/utils/nn_helpers/lstm_helper.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_UTILS_NN_HELPERS_LSTM_HELPER_CUH
#define MPPI_FWD_UTILS_NN_HELPERS_LSTM_HELPER_CUH
#include "mppi_amd/utils/nn_helpers/lstm_helper.hpp"
#endif
