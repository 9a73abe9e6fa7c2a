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

struct ModelRegistrar
{
  ModelRegistrar(const char* name, int sampler_kind, mppi_model_factory factory)
  {
    (void)mppi_register_model(name, sampler_kind, factory, engineAbiFingerprint());
  }
};
}  // namespace engine
}  // namespace mppi

#define MPPI_REGISTRY_CAT2(a, b) a##b
#define MPPI_REGISTRY_CAT(a, b) MPPI_REGISTRY_CAT2(a, b)
/** one line per instantiation, at namespace scope of a .hip file */
#define MPPI_REGISTER_MODEL(name, sampler_kind, MODEL_T, default_bx, default_by)                                         \
  static ::mppi::engine::ModelRegistrar MPPI_REGISTRY_CAT(mppi_model_registrar_, __LINE__)(                             \
      name, sampler_kind, &::mppi::engine::modelFactory<MODEL_T, default_bx, default_by>);

#endif
