/**
 * model_registry.hpp — how a (Dynamics, Cost, Sampler) instantiation gets into the engine's model table.
 *
 * The reference's user instantiates the controller templates in a translation unit of their own and links it
 * (reference: src/controllers/cartpole/cartpole_mppi.cu:30-42, include/mppi/instantiations/cartpole_mppi/cartpole_mppi.cuh).
 * Here that translation unit is a .hip file that names one ModelT<...> and registers a factory for it under a model
 * name; the in-tree models (mppi-generic_amd/csrc/models/[*].hip) and a user's out-of-tree model (examples/my_model/)
 * do exactly the same thing:
 *
 *     using MyModel = mppi::engine::ModelT<MyDynamics, MyCost, MySampler, Shapes<Shape<64, 1, 1>>, 1, void, Shapes<>, true>;
 *     MPPI_REGISTER_MODEL("my_model", MPPI_SAMPLER_GAUSSIAN, MyModel, 64, 1)
 *
 * (the trailing `true` = PIPELINE asks for the role-separated rollout kernels: MyDynamics and MyCost then have to DECLARE that
 * their per-step device methods hold no block barrier — `static constexpr bool MPPI_BARRIER_FREE_STEP = true;`,
 * plugin/parallel_utils.hpp — or the registration is refused: a barrier there would hang the GPU.  A model written like the
 * reference's, __syncthreads() and all, registers with PIPELINE = false and runs on the fused kernel.)
 *
 * compiled with   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -I<include> my_model.hip -o libmy_model.so
 * and either linked next to libmppi_amd.so or loaded at run time with mppi_load_plugin("libmy_model.so").
 * The kernels of the model live in the user's library; libmppi_amd.so only ever calls them through ModelBase.
 */
#ifndef MPPI_AMD_ENGINE_MODEL_REGISTRY_HPP_
#define MPPI_AMD_ENGINE_MODEL_REGISTRY_HPP_

#include "mppi_amd.h"
#include "model_instance.hpp"

namespace mppi
{
namespace engine
{
template <class MODEL_T, int DEFAULT_BX, int DEFAULT_BY>
void* modelFactory()
{
  ModelBase* m = new MODEL_T();
  m->default_bx = DEFAULT_BX;
  m->default_by = DEFAULT_BY;
  return m;
}

/** mppi_register_model_checked's flags for an instantiation (mppi_amd.h: MPPI_MODEL_ROLE_SEPARATED, MPPI_MODEL_BARRIER_FREE_DECLARED) */
template <class MODEL_T>
constexpr unsigned modelFlags()
{
  return (MODEL_T::ROLE_SEPARATED ? MPPI_MODEL_ROLE_SEPARATED : 0u) |
         (MODEL_T::BARRIER_FREE_DECLARED ? MPPI_MODEL_BARRIER_FREE_DECLARED : 0u);
}

struct ModelRegistrar
{
  ModelRegistrar(const char* name, int sampler_kind, mppi_model_factory factory, unsigned flags)
  {
    (void)mppi_register_model_checked(name, sampler_kind, factory, engineAbiFingerprint(), flags);
  }
};
}  // namespace engine
}  // namespace mppi

#define MPPI_REGISTRY_CAT2(a, b) a##b
#define MPPI_REGISTRY_CAT(a, b) MPPI_REGISTRY_CAT2(a, b)
/** one line per instantiation, at namespace scope of a .hip file */
#define MPPI_REGISTER_MODEL(name, sampler_kind, MODEL_T, default_bx, default_by)                                         \
  static ::mppi::engine::ModelRegistrar MPPI_REGISTRY_CAT(mppi_model_registrar_, __LINE__)(                             \
      name, sampler_kind, &::mppi::engine::modelFactory<MODEL_T, default_bx, default_by>,                               \
      ::mppi::engine::modelFlags<MODEL_T>());

#endif
