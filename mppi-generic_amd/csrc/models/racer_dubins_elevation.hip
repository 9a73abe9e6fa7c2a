/**
 * racer_dubins_elevation.hip — registered instantiation(s) of libmppi_amd.so: the RACER Dubins car on an elevation map
 * (static settling on a TwoDTextureHelper map + 4x4 covariance propagation) + QuadraticCost, Gaussian and colored-noise
 * samplers.
 *
 * The analogue of the reference's include/mppi/instantiations/ + src/controllers/ (explicit template instantiations
 * compiled into shared libraries, e.g. src/controllers/cartpole/cartpole_mppi.cu:30-42).  One translation unit per
 * model, so a new or changed model recompiles alone (buildlib.py compiles the units in parallel).
 *
 * Block shapes (BX rollouts, BY lanes per rollout, BZ systems per launch): BY == 1 — a rollout is one lane with its
 * 19 states and the three 4x4 matrices of the covariance step in VGPRs — or BY == 4: four replica lanes per rollout
 * that share out the wheels, the covariance rows and the trigonometry of a step (RacerDubinsElevationQuad).
 *
 * The cost is QuadraticCost over the 28 outputs with SKIP_ZERO_COEFF: the model marks the three wheel-force outputs it
 * does not produce with NaN (racer_dubins_elevation.cu:131-139); give them coefficient 0.
 * Robust MPPI (RMPPI = true): both of its kernels run the four-lanes-per-rollout form.
 */
#include "mppi_amd/engine/model_registry.hpp"
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#include "mppi_amd/sampling_distributions/colored_noise.hpp"
#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation.hpp"
#include "mppi_amd/cost_functions/quadratic_cost/quadratic_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

using ElevationCost = QuadraticCost<RacerDubinsElevation, /*SKIP_ZERO_COEFF=*/true>;
using RacerElevationModel =
    ModelT<RacerDubinsElevation, ElevationCost, sampling_distributions::GaussianDistribution<RacerDubinsElevationParams>,
           Shapes<Shape<64, 1, 1>, Shape<32, 1, 1>, Shape<64, 1, 2>>, /*FIN_BY=*/1,
           /* four lanes per rollout: one wheel, one covariance row, one angle each (racer_dubins_elevation.hpp) */
           RacerDubinsElevationQuad, Shapes<Shape<64, 4, 1>, Shape<64, 4, 2>>, /*PIPELINE=*/true, /*RMPPI=*/true>;
using RacerElevationColoredModel =
    ModelT<RacerDubinsElevation, ElevationCost, sampling_distributions::ColoredNoiseDistribution<RacerDubinsElevationParams>,
           Shapes<Shape<64, 1, 1>>, /*FIN_BY=*/1, RacerDubinsElevationQuad, Shapes<Shape<64, 4, 1>>, /*PIPELINE=*/true>;
MPPI_REGISTER_MODEL("racer_dubins_elevation", MPPI_SAMPLER_GAUSSIAN, RacerElevationModel, 64, 4)
MPPI_REGISTER_MODEL("racer_dubins_elevation", MPPI_SAMPLER_COLORED, RacerElevationColoredModel, 64, 4)
