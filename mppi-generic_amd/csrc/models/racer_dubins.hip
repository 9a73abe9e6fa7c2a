/**
 * racer_dubins.hip — registered instantiation(s) of libmppi_amd.so: RACER Dubins car + QuadraticCost, Gaussian and colored-noise samplers.
 *
 * The analogue of the reference's include/mppi/instantiations/ + src/controllers/ (explicit template instantiations
 * compiled into shared libraries, e.g. src/controllers/cartpole/cartpole_mppi.cu:30-42).  One translation unit per
 * model and sampler, so a new or changed model recompiles alone (buildlib.py compiles the units in parallel).
 *
 * Block shapes (BX rollouts, BY lanes per rollout, BZ systems per launch):
 *   BY == 1 : one lane per rollout, state in VGPRs, no barriers      — analytic models (cartpole, double integrator)
 *   BY  > 1 : the reference's LDS + barrier scheme                     — kept for contract coverage and NN-sized models
 *   BZ == 2 : Tube / RMPPI (actual + nominal system share one launch, tube_mppi_controller.cu:192-209)
 */
#include "mppi_amd/engine/model_registry.hpp"
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#include "mppi_amd/sampling_distributions/colored_noise.hpp"
#include "mppi_amd/dynamics/racer_dubins/racer_dubins.hpp"
#include "mppi_amd/cost_functions/quadratic_cost/quadratic_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

/* RACER Dubins car + QuadraticCost over its 28 outputs (dynamics/racer_dubins/racer_dubins.cuh,
 * cost_functions/quadratic_cost/quadratic_cost.cuh) */
using RacerSampler = sampling_distributions::GaussianDistribution<RacerDubinsParams>;
using RacerDubinsModel = ModelT<RacerDubins, QuadraticCost<RacerDubins>, RacerSampler,
                                Shapes<Shape<64, 1, 1>, Shape<32, 1, 1>, Shape<64, 1, 2>, Shape<16, 1, 1>, Shape<16, 1, 2>>, /*FIN_BY=*/1,
                                void, Shapes<>, /*PIPELINE=*/true, /*RMPPI=*/true>;
using RacerDubinsColoredModel =
    ModelT<RacerDubins, QuadraticCost<RacerDubins>, sampling_distributions::ColoredNoiseDistribution<RacerDubinsParams>,
           Shapes<Shape<64, 1, 1>>, /*FIN_BY=*/1, void, Shapes<>, /*PIPELINE=*/true>;
MPPI_REGISTER_MODEL("racer_dubins", MPPI_SAMPLER_GAUSSIAN, RacerDubinsModel, 64, 1)
MPPI_REGISTER_MODEL("racer_dubins", MPPI_SAMPLER_COLORED, RacerDubinsColoredModel, 64, 1)
