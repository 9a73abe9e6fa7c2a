/* forwarding header: the reference's include path (include/mppi/dynamics/racer_dubins/racer_dubins_elevation_suspension_lstm.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_DYNAMICS_RACER_DUBINS_RACER_DUBINS_ELEVATION_SUSPENSION_LSTM_CUH
#define MPPI_FWD_DYNAMICS_RACER_DUBINS_RACER_DUBINS_ELEVATION_SUSPENSION_LSTM_CUH
#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation_suspension.hpp"
#endif
