/**
 * autorally_nn.hip — registered instantiation(s) of libmppi_amd.so: AutoRally NeuralNetModel<7,2,3> + ARStandardCost, Gaussian sampler.
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
#include "mppi_amd/dynamics/autorally/ar_nn_model.hpp"
#include "mppi_amd/cost_functions/autorally/ar_standard_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

/* AutoRally: MLP dynamics + costmap cost (reference: instantiations/autorally_mppi/autorally_mppi.cuh:10-13 uses
 * dynamics_rollout_dim (8, 16, 1)).  BY lanes of a rollout share the neurons of a layer. */
using ARModelDyn = NeuralNetModel<7, 2, 3>;
using ARSampler = sampling_distributions::GaussianDistribution<NNDynamicsParams>;
using ARModel = ModelT<ARModelDyn, ARStandardCost, ARSampler,
                       Shapes<Shape<8, 16, 1>, Shape<16, 8, 1>, Shape<16, 4, 1>, Shape<64, 1, 1>, Shape<8, 16, 2>>,
                       /*FIN_BY=*/32,
                       /* MFMA forward: BX rollouts x 4 k-group lanes per block (BX/16 waves) */
                       NeuralNetModelMFMA<7, 2, 3>, Shapes<Shape<64, 4, 1>, Shape<32, 4, 1>, Shape<64, 4, 2>, Shape<32, 4, 2>>,
                       /*PIPELINE=*/false, /*RMPPI=*/true>;  // Robust MPPI runs on the MFMA forward too
/* default (64, 4): MFMA variant, 64 rollouts x 4 lanes = 4 waves, one per SIMD of a CU */
MPPI_REGISTER_MODEL("autorally_nn", MPPI_SAMPLER_GAUSSIAN, ARModel, 64, 4)

#if defined(MPPI_PIPE_TIMING)
/* A/B instrumentation read-back (tools/pipe_timing.py): ticks[blocks][waves][slots] of the last pipelined launch */
extern "C" int mppi_debug_read_pipe_timing(unsigned long long* out, int capacity)
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
