/**
 * double_integrator_robust_colored.hip — registered instantiation(s) of libmppi_amd.so: DoubleIntegrator + DoubleIntegratorRobustCost, colored-noise sampler.
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
#include "mppi_amd/sampling_distributions/colored_noise.hpp"
#include "mppi_amd/dynamics/double_integrator/di_dynamics.hpp"
#include "mppi_amd/cost_functions/double_integrator/double_integrator_robust_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

using DIRobustColoredModel =
    ModelT<DoubleIntegratorDynamics, DoubleIntegratorRobustCost,
           sampling_distributions::ColoredNoiseDistribution<DoubleIntegratorParams>, Shapes<Shape<64, 1, 1>>, /*FIN_BY=*/1,
           void, Shapes<>, /*PIPELINE=*/true>;
MPPI_REGISTER_MODEL("double_integrator_robust", MPPI_SAMPLER_COLORED, DIRobustColoredModel, 64, 1)
