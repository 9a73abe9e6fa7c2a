/**
 * bicycle_slip_lstm_colored.hip — registered instantiation(s) of libmppi_amd.so: LSTM bicycle-slip dynamics + ARStandardCost, colored-noise sampler (BASELINE config 5).
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
#include "mppi_amd/dynamics/bicycle_slip/bicycle_slip_lstm.hpp"
#include "mppi_amd/cost_functions/autorally/ar_standard_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

using BSLColoredModel =
    ModelT<BicycleSlipLSTM, ARStandardCost, sampling_distributions::ColoredNoiseDistribution<BicycleSlipLSTMParams>,
           Shapes<Shape<16, 8, 1>>, /*FIN_BY=*/32, BicycleSlipLSTMMFMA, Shapes<Shape<64, 4, 1>, Shape<32, 4, 1>>>;
MPPI_REGISTER_MODEL("bicycle_slip_lstm", MPPI_SAMPLER_COLORED, BSLColoredModel, 64, 4)
