/**
 * racer_dubins_elevation_lstm_unc.hip — registered instantiation(s) of libmppi_amd.so: the complete RACER vehicle model
 * (suspension + LSTM steering + mean LSTM + uncertainty LSTM + static settling, 26 states) + QuadraticCost, Gaussian and
 * colored-noise samplers.
 *
 * The analogue of the reference's include/mppi/instantiations/ + src/controllers/ (explicit template instantiations
 * compiled into shared libraries, e.g. src/controllers/cartpole/cartpole_mppi.cu:30-42).  One translation unit per
 * model, so a new or changed model recompiles alone (buildlib.py compiles the units in parallel).
 *
 * Block shapes: (64, 4) — the default; (32, 4, 2) for two systems — runs a rollout on four replica lanes
 * (RacerDubinsElevationLSTMUncertaintyQuad: wheels, covariance rows, hidden units and output-network neurons shared out, the
 * mean / uncertainty networks' weights kept once per 16-lane row and fetched with DPP row broadcasts).  BY == 1: one lane per
 * rollout, the three networks on registers with scalar-unit weights (racer_dubins_elevation_lstm_unc.hpp).
 * Robust MPPI (RMPPI = true): both of its kernels run the four-lanes-per-rollout form.
 */
#include "mppi_amd/engine/model_registry.hpp"
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#include "mppi_amd/sampling_distributions/colored_noise.hpp"
#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation_lstm_unc.hpp"
#include "mppi_amd/cost_functions/quadratic_cost/quadratic_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

using UncertaintyCost = QuadraticCost<RacerDubinsElevationLSTMUncertainty, /*SKIP_ZERO_COEFF=*/true>;
using RacerUncertaintyModel =
    ModelT<RacerDubinsElevationLSTMUncertainty, UncertaintyCost,
           sampling_distributions::GaussianDistribution<RacerDubinsElevationUncertaintyParams>,
           Shapes<Shape<64, 1, 1>, Shape<32, 1, 1>, Shape<64, 1, 2>>, /*FIN_BY=*/2,
           /* four lanes per rollout: a wheel, a covariance row, a hidden unit and five output-network neurons of each of the
              three networks per lane */
           RacerDubinsElevationLSTMUncertaintyQuad, Shapes<Shape<64, 4, 1>, Shape<32, 4, 2>>, /*PIPELINE=*/false, /*RMPPI=*/true>;
using RacerUncertaintyColoredModel =
    ModelT<RacerDubinsElevationLSTMUncertainty, UncertaintyCost,
           sampling_distributions::ColoredNoiseDistribution<RacerDubinsElevationUncertaintyParams>, Shapes<Shape<64, 1, 1>>,
           /*FIN_BY=*/2, RacerDubinsElevationLSTMUncertaintyQuad, Shapes<Shape<64, 4, 1>>, /*PIPELINE=*/false>;
MPPI_REGISTER_MODEL("racer_dubins_elevation_lstm_unc", MPPI_SAMPLER_GAUSSIAN, RacerUncertaintyModel, 64, 4)
MPPI_REGISTER_MODEL("racer_dubins_elevation_lstm_unc", MPPI_SAMPLER_COLORED, RacerUncertaintyColoredModel, 64, 4)
