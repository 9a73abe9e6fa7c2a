/**
 * models.hpp — the registered (Dynamics, Cost, Sampler) instantiations of libmppi_amd.so.
 *
 * This file is the analogue of the reference's include/mppi/instantiations/ + src/controllers/ (explicit template
 * instantiations compiled into shared libraries).  To add a model: write the plugins against include/mppi_amd/plugin/,
 * add one MODEL entry below with the block shapes it should be compiled for, rebuild.
 *
 * Block shapes (BX rollouts, BY lanes per rollout, BZ systems per launch):
 *   BY == 1 : one lane per rollout, state in VGPRs, no barriers      — analytic models (cartpole, double integrator)
 *   BY  > 1 : the reference's LDS + barrier scheme                     — kept for contract coverage and NN-sized models
 *   BZ == 2 : Tube / RMPPI (actual + nominal system share one launch, tube_mppi_controller.cu:192-209)
 */
#ifndef MPPI_AMD_MODELS_HPP_
#define MPPI_AMD_MODELS_HPP_

#include <string>

#include "model_instance.hpp"
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#include "mppi_amd/sampling_distributions/colored_noise.hpp"
#include "mppi_amd/dynamics/cartpole/cartpole_dynamics.hpp"
#include "mppi_amd/cost_functions/cartpole/cartpole_quadratic_cost.hpp"
#include "mppi_amd/dynamics/double_integrator/di_dynamics.hpp"
#include "mppi_amd/cost_functions/double_integrator/double_integrator_circle_cost.hpp"
#include "mppi_amd/dynamics/autorally/ar_nn_model.hpp"
#include "mppi_amd/cost_functions/autorally/ar_standard_cost.hpp"
#include "mppi_amd/dynamics/bicycle_slip/bicycle_slip_lstm.hpp"
#include "mppi_amd/dynamics/racer_dubins/racer_dubins.hpp"
#include "mppi_amd/cost_functions/quadratic_cost/quadratic_cost.hpp"

namespace mppi
{
namespace engine
{
using CartpoleSampler = sampling_distributions::GaussianDistribution<CartpoleDynamicsParams>;
using CartpoleModel = ModelT<CartpoleDynamics, CartpoleQuadraticCost, CartpoleSampler,
                             Shapes<Shape<64, 1, 1>, Shape<64, 1, 2>, Shape<32, 1, 1>, Shape<64, 4, 1>, Shape<16, 4, 1>,
                                    /* long horizons (the sample rows of a block live in LDS): */ Shape<16, 1, 1>, Shape<16, 1, 2>>,
                             /*FIN_BY=*/1, void, Shapes<>, /*PIPELINE=*/true, /*RMPPI=*/true>;

using DISampler = sampling_distributions::GaussianDistribution<DoubleIntegratorParams>;
using DIModel = ModelT<DoubleIntegratorDynamics, DoubleIntegratorCircleCost, DISampler,
                       Shapes<Shape<64, 1, 1>, Shape<64, 1, 2>, Shape<32, 2, 2>, Shape<64, 2, 1>, Shape<16, 1, 1>, Shape<16, 1, 2>>,
                       /*FIN_BY=*/1, void,
                       Shapes<>, /*PIPELINE=*/true, /*RMPPI=*/true>;

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

/* LSTM bicycle-slip dynamics (BASELINE config 5): LSTM(6, 16) + MLP {22, 32, 4}, AutoRally state layout and cost.
 * (BX, 4) = MFMA forward with the recurrent state in registers; the other shapes run LSTMHelper's LDS scheme. */
using BSLSampler = sampling_distributions::GaussianDistribution<BicycleSlipLSTMParams>;
using BSLModel = ModelT<BicycleSlipLSTM, ARStandardCost, BSLSampler,
                        Shapes<Shape<16, 8, 1>, Shape<16, 4, 1>, Shape<64, 1, 1>, Shape<8, 16, 1>, Shape<16, 8, 2>>,
                        /*FIN_BY=*/32, BicycleSlipLSTMMFMA, Shapes<Shape<64, 4, 1>, Shape<32, 4, 1>, Shape<64, 4, 2>, Shape<32, 4, 2>>,
                        /*PIPELINE=*/false, /*RMPPI=*/true>;

/* RACER Dubins car + QuadraticCost over its 28 outputs (dynamics/racer_dubins/racer_dubins.cuh,
 * cost_functions/quadratic_cost/quadratic_cost.cuh) */
using RacerSampler = sampling_distributions::GaussianDistribution<RacerDubinsParams>;
using RacerDubinsModel = ModelT<RacerDubins, QuadraticCost<RacerDubins>, RacerSampler,
                                Shapes<Shape<64, 1, 1>, Shape<32, 1, 1>, Shape<64, 1, 2>, Shape<16, 1, 1>, Shape<16, 1, 2>>, /*FIN_BY=*/1,
                                void, Shapes<>,
                                /*PIPELINE=*/true, /*RMPPI=*/true>;
using RacerDubinsColoredModel =
    ModelT<RacerDubins, QuadraticCost<RacerDubins>, sampling_distributions::ColoredNoiseDistribution<RacerDubinsParams>,
           Shapes<Shape<64, 1, 1>>, /*FIN_BY=*/1, void, Shapes<>, /*PIPELINE=*/true>;

/* ColoredMPPI instantiations (reference: controllers/ColoredMPPI/colored_mppi_controller.cuh with
 * ColoredNoiseDistribution as SAMPLING_T): the same plugins with the colored-noise sampler. */
using CartpoleColoredModel =
    ModelT<CartpoleDynamics, CartpoleQuadraticCost, sampling_distributions::ColoredNoiseDistribution<CartpoleDynamicsParams>,
           Shapes<Shape<64, 1, 1>, Shape<64, 4, 1>>, /*FIN_BY=*/1, void, Shapes<>, /*PIPELINE=*/true>;
using DIColoredModel =
    ModelT<DoubleIntegratorDynamics, DoubleIntegratorCircleCost,
           sampling_distributions::ColoredNoiseDistribution<DoubleIntegratorParams>, Shapes<Shape<64, 1, 1>>, /*FIN_BY=*/1,
           void, Shapes<>, /*PIPELINE=*/true>;
using ARColoredModel = ModelT<ARModelDyn, ARStandardCost, sampling_distributions::ColoredNoiseDistribution<NNDynamicsParams>,
                              Shapes<Shape<16, 8, 1>>, /*FIN_BY=*/32, NeuralNetModelMFMA<7, 2, 3>,
                              Shapes<Shape<64, 4, 1>, Shape<32, 4, 1>>>;
using BSLColoredModel =
    ModelT<BicycleSlipLSTM, ARStandardCost, sampling_distributions::ColoredNoiseDistribution<BicycleSlipLSTMParams>,
           Shapes<Shape<16, 8, 1>>, /*FIN_BY=*/32, BicycleSlipLSTMMFMA, Shapes<Shape<64, 4, 1>, Shape<32, 4, 1>>>;

inline ModelBase* makeModel(const std::string& name, bool colored = false)
{
  if (colored)
  {
    ModelBase* m = nullptr;
    if (name == "cartpole")
      m = new CartpoleColoredModel();
    else if (name == "double_integrator")
      m = new DIColoredModel();
    else if (name == "autorally_nn")
      m = new ARColoredModel();
    else if (name == "bicycle_slip_lstm")
      m = new BSLColoredModel();
    else if (name == "racer_dubins")
      m = new RacerDubinsColoredModel();
    if (m && (name == "autorally_nn" || name == "bicycle_slip_lstm"))
    {
      m->default_bx = 64;
      m->default_by = 4;
    }
    return m;
  }
  if (name == "bicycle_slip_lstm")
  {
    ModelBase* m = new BSLModel();
    m->default_bx = 64;
    m->default_by = 4;
    return m;
  }
  if (name == "autorally_nn")
  {
    ModelBase* m = new ARModel();
    m->default_bx = 64;  // MFMA variant: 64 rollouts x 4 lanes = 4 waves, one per SIMD of a CU
    m->default_by = 4;
    return m;
  }
  if (name == "cartpole")
    return new CartpoleModel();
  if (name == "double_integrator")
    return new DIModel();
  if (name == "racer_dubins")
    return new RacerDubinsModel();
  return nullptr;
}

inline const char* listModels()
{
  return "cartpole\ndouble_integrator\nautorally_nn\nbicycle_slip_lstm\nracer_dubins";
}

}  // namespace engine
}  // namespace mppi

#endif
