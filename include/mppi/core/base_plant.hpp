/* forwarding header: the reference's include path (include/mppi/core/base_plant.hpp) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_CORE_BASE_PLANT_HPP
#define MPPI_FWD_CORE_BASE_PLANT_HPP
#include "mppi_amd/plant.hpp"
#endif
