/* forwarding header: the reference's include path (include/mppi/sampling_distributions/colored_noise/colored_noise.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_SAMPLING_DISTRIBUTIONS_COLORED_NOISE_COLORED_NOISE_CUH
#define MPPI_FWD_SAMPLING_DISTRIBUTIONS_COLORED_NOISE_COLORED_NOISE_CUH
#include "mppi_amd/sampling_distributions/colored_noise.hpp"
#endif
