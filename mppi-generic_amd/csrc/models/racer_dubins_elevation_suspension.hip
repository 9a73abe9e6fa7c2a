/**
 * racer_dubins_elevation_suspension.hip — registered instantiation(s) of libmppi_amd.so: the elevation-map RACER Dubins car
 * with LSTM steering and a spring / damper suspension (24 states; terrain heights + terrain normals maps) + QuadraticCost,
 * Gaussian and colored-noise samplers.
 *
 * The analogue of the reference's include/mppi/instantiations/ + src/controllers/ (explicit template instantiations
 * compiled into shared libraries, e.g. src/controllers/cartpole/cartpole_mppi.cu:30-42).  One translation unit per
 * model, so a new or changed model recompiles alone (buildlib.py compiles the units in parallel).
 *
 * Block shapes: (64, 4) — the default — runs a rollout on four replica lanes that share out the wheels of the suspension,
 * the hidden units and neurons of the steering network and the rows of the covariance update
 * (RacerDubinsElevationSuspensionQuad; default network shape only).  Two systems (Tube): (32, 4, 2).  A (64, 4, 2) block is 512
 * threads, i.e. 256 registers per lane.  While every lane kept its own copy of the steering weights (81 registers, ~200
 * values of the step spilled) that instantiation returned NaN costs for injected noise when compiled at -O3 — correct at
 * -O2, correct with the in-kernel draw, correct for every other model, and correct again since the weights moved into
 * DPP rows (lstm_quad.hpp) and the spills went away.  Not understood beyond that, so it stays un-instantiated; the
 * 256-thread block keeps the step in registers with room to spare.  BY == 1: one lane per rollout; fused rollout kernel
 * (the LDS fallback of the steering network needs a block barrier in initializeDynamics, see
 * racer_dubins_elevation_lstm_steering.hip).
 * Robust MPPI (RMPPI = true): both of its kernels run the four-lanes-per-rollout form.
 */
#include "mppi_amd/engine/model_registry.hpp"
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#include "mppi_amd/sampling_distributions/colored_noise.hpp"
#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation_suspension.hpp"
#include "mppi_amd/cost_functions/quadratic_cost/quadratic_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

using SuspensionCost = QuadraticCost<RacerDubinsElevationSuspension, /*SKIP_ZERO_COEFF=*/true>;
using RacerSuspensionModel =
    ModelT<RacerDubinsElevationSuspension, SuspensionCost,
           sampling_distributions::GaussianDistribution<RacerDubinsElevationSuspensionParams>,
           Shapes<Shape<64, 1, 1>, Shape<32, 1, 1>, Shape<64, 1, 2>>, /*FIN_BY=*/2,
           /* four lanes per rollout: a wheel, a covariance row, a hidden unit and five MLP neurons each */
           RacerDubinsElevationSuspensionQuad, Shapes<Shape<64, 4, 1>, Shape<32, 4, 2>>, /*PIPELINE=*/false, /*RMPPI=*/true>;
using RacerSuspensionColoredModel =
    ModelT<RacerDubinsElevationSuspension, SuspensionCost,
           sampling_distributions::ColoredNoiseDistribution<RacerDubinsElevationSuspensionParams>, Shapes<Shape<64, 1, 1>>,
           /*FIN_BY=*/2, RacerDubinsElevationSuspensionQuad, Shapes<Shape<64, 4, 1>>, /*PIPELINE=*/false>;
MPPI_REGISTER_MODEL("racer_dubins_elevation_suspension", MPPI_SAMPLER_GAUSSIAN, RacerSuspensionModel, 64, 4)
MPPI_REGISTER_MODEL("racer_dubins_elevation_suspension", MPPI_SAMPLER_COLORED, RacerSuspensionColoredModel, 64, 4)
