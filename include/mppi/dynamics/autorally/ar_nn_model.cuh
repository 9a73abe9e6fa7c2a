/* forwarding header: the reference's include path (include/mppi/dynamics/autorally/ar_nn_model.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_DYNAMICS_AUTORALLY_AR_NN_MODEL_CUH
#define MPPI_FWD_DYNAMICS_AUTORALLY_AR_NN_MODEL_CUH
#include "mppi_amd/dynamics/autorally/ar_nn_model.hpp"
#endif
