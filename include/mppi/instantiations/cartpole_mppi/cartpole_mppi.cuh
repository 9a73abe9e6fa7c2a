/* forwarding header: the reference's include path (include/mppi/instantiations/cartpole_mppi/cartpole_mppi.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_INSTANTIATIONS_CARTPOLE_MPPI_CARTPOLE_MPPI_CUH
#define MPPI_FWD_INSTANTIATIONS_CARTPOLE_MPPI_CARTPOLE_MPPI_CUH
#include "mppi/controllers/MPPI/mppi_controller.cuh"
#include "mppi/cost_functions/cartpole/cartpole_quadratic_cost.cuh"
#include "mppi/dynamics/cartpole/cartpole_dynamics.cuh"
#include "mppi/feedback_controllers/DDP/ddp.cuh"
#endif
