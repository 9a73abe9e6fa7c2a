/* forwarding header: the reference's include path (include/mppi/feedback_controllers/feedback.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_FEEDBACK_CONTROLLERS_FEEDBACK_CUH
#define MPPI_FWD_FEEDBACK_CONTROLLERS_FEEDBACK_CUH
#include "mppi_amd/feedback_controllers/ddp_feedback.hpp"
#endif
