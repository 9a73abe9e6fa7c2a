/* forwarding header: the reference's include path (include/mppi/dynamics/racer_dubins/racer_dubins_elevation_lstm_unc.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_DYNAMICS_RACER_DUBINS_RACER_DUBINS_ELEVATION_LSTM_UNC_CUH
#define MPPI_FWD_DYNAMICS_RACER_DUBINS_RACER_DUBINS_ELEVATION_LSTM_UNC_CUH
#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation_lstm_unc.hpp"
#endif
