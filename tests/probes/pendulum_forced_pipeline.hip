/**
 * Negative probe of tests/test_plugin_model.py: the reference-style pendulum (examples/my_model/pendulum_reference_style.cuh — a
 * step() with two __syncthreads(), no MPPI_BARRIER_FREE_STEP declaration) FORCED onto the role-pipelined kernels with
 * PIPELINE = true.  Launched, the dynamics wave would wait at its barrier for sampler and cost waves that never execute
 * step(): a GPU hang.  The registration must be refused instead (mppi_register_model_checked, MPPI_ERR_INVALID_ARG), so
 * mppi_load_plugin of this library fails and the model name never appears in the table.  It is only ever compiled and loaded,
 * never run.
 */
#include "../../examples/my_model/pendulum_reference_style.cuh"
#include "mppi_amd/engine/model_registry.hpp"

using namespace mppi::engine;
using ForcedSampler = mppi::sampling_distributions::GaussianDistribution<RefPendulumParams>;
using ForcedModel = ModelT<RefPendulumDynamics, RefPendulumCost, ForcedSampler, Shapes<Shape<64, 1, 1>>, /*FIN_BY=*/1, void, Shapes<>,
                           /*PIPELINE=*/true>;
static_assert(ForcedModel::ROLE_SEPARATED && !ForcedModel::BARRIER_FREE_DECLARED, "the probe must be the refused case");
MPPI_REGISTER_MODEL("user_pendulum_forced_pipeline", MPPI_SAMPLER_GAUSSIAN, ForcedModel, 64, 1)

/* The same instantiation through the UNCHECKED entry point (what a plugin built before round 6's macro, or a hand-written
 * registration, would call): it does enter the table — and mppi_create refuses it (ModelBase::undeclaredBarrierFreePlugins), so the
 * second line of defence is tested too (tests/test_plugin_model.py, on a GPU: creation needs a device). */
namespace
{
struct UncheckedRegistrar
{
  UncheckedRegistrar()
  {
    (void)mppi_register_model("user_pendulum_forced_pipeline_unchecked", MPPI_SAMPLER_GAUSSIAN, &modelFactory<ForcedModel, 64, 1>,
                              engineAbiFingerprint());
  }
} unchecked_registrar;
}  // namespace
