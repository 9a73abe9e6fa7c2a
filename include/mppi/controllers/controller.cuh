/* forwarding header: the reference's include path (include/mppi/controllers/controller.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_CONTROLLERS_CONTROLLER_CUH
#define MPPI_FWD_CONTROLLERS_CONTROLLER_CUH
#include "mppi_amd/controllers_templated.hpp"
#endif
