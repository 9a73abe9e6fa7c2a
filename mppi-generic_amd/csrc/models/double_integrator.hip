/**
 * double_integrator.hip — registered instantiation(s) of libmppi_amd.so: DoubleIntegrator + DoubleIntegratorCircleCost, Gaussian sampler.
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
#include "mppi_amd/dynamics/double_integrator/di_dynamics.hpp"
#include "mppi_amd/cost_functions/double_integrator/double_integrator_circle_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

using DISampler = sampling_distributions::GaussianDistribution<DoubleIntegratorParams>;
using DIModel = ModelT<DoubleIntegratorDynamics, DoubleIntegratorCircleCost, DISampler,
                       Shapes<Shape<64, 1, 1>, Shape<64, 1, 2>, Shape<32, 2, 2>, Shape<64, 2, 1>, Shape<16, 1, 1>, Shape<16, 1, 2>>,
                       /*FIN_BY=*/1, void, Shapes<>, /*PIPELINE=*/true, /*RMPPI=*/true>;
MPPI_REGISTER_MODEL("double_integrator", MPPI_SAMPLER_GAUSSIAN, DIModel, 64, 1)

#if defined(MPPI_PIPE_TIMING)
/* A/B instrumentation read-back (tools/pipe_timing_cartpole.py di_tube): ticks[blocks][waves][slots] of the last pipelined launch */
extern "C" int mppi_debug_read_pipe_timing_double_integrator(unsigned long long* out, int capacity)
{
  constexpr int N = kernels::PIPE_TIMING_BLOCKS * kernels::PIPE_TIMING_WAVES * kernels::PIPE_TIMING_SLOTS;
  if (!out || capacity < N)
    return -N;
  if (hipDeviceSynchronize() != hipSuccess)
    return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(kernels::g_pipe_timing), sizeof(unsigned long long) * N) != hipSuccess)
    return -2;
  return N;
}
#endif
