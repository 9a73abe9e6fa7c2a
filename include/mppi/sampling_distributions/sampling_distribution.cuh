/* forwarding header: the reference's include path (include/mppi/sampling_distributions/sampling_distribution.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_SAMPLING_DISTRIBUTIONS_SAMPLING_DISTRIBUTION_CUH
#define MPPI_FWD_SAMPLING_DISTRIBUTIONS_SAMPLING_DISTRIBUTION_CUH
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#endif
