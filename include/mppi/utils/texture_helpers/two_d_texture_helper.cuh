/* forwarding header: the reference's include path (include/mppi/utils/texture_helpers/two_d_texture_helper.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_UTILS_TEXTURE_HELPERS_TWO_D_TEXTURE_HELPER_CUH
#define MPPI_FWD_UTILS_TEXTURE_HELPERS_TWO_D_TEXTURE_HELPER_CUH
#include "mppi_amd/utils/texture_helpers/two_d_texture_helper.hpp"
#endif
