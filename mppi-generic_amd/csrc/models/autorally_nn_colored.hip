/**
 * autorally_nn_colored.hip — registered instantiation(s) of libmppi_amd.so: AutoRally NeuralNetModel<7,2,3> + ARStandardCost, colored-noise sampler.
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
#include "mppi_amd/dynamics/autorally/ar_nn_model.hpp"
#include "mppi_amd/cost_functions/autorally/ar_standard_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

using ARModelDyn = NeuralNetModel<7, 2, 3>;
using ARColoredModel = ModelT<ARModelDyn, ARStandardCost, sampling_distributions::ColoredNoiseDistribution<NNDynamicsParams>,
                              Shapes<Shape<16, 8, 1>>, /*FIN_BY=*/32, NeuralNetModelMFMA<7, 2, 3>,
                              Shapes<Shape<64, 4, 1>, Shape<32, 4, 1>>>;
MPPI_REGISTER_MODEL("autorally_nn", MPPI_SAMPLER_COLORED, ARColoredModel, 64, 4)
