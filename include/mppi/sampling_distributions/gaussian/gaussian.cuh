/* forwarding header: the reference's include path (include/mppi/sampling_distributions/gaussian/gaussian.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_SAMPLING_DISTRIBUTIONS_GAUSSIAN_GAUSSIAN_CUH
#define MPPI_FWD_SAMPLING_DISTRIBUTIONS_GAUSSIAN_GAUSSIAN_CUH
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#endif
