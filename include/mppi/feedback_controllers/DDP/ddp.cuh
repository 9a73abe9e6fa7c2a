/* forwarding header: the reference's include path (include/mppi/feedback_controllers/DDP/ddp.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_FEEDBACK_CONTROLLERS_DDP_DDP_CUH
#define MPPI_FWD_FEEDBACK_CONTROLLERS_DDP_DDP_CUH
#include "mppi_amd/feedback_controllers/ddp_feedback.hpp"
#endif
