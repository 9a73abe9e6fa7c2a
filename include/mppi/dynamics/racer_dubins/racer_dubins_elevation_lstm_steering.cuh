/* forwarding header: the reference's include path (include/mppi/dynamics/racer_dubins/racer_dubins_elevation_lstm_steering.cuh) -> this engine's header.  Paths only. */
#ifndef MPPI_FWD_DYNAMICS_RACER_DUBINS_RACER_DUBINS_ELEVATION_LSTM_STEERING_CUH
#define MPPI_FWD_DYNAMICS_RACER_DUBINS_RACER_DUBINS_ELEVATION_LSTM_STEERING_CUH
#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation_lstm_steering.hpp"
#endif
