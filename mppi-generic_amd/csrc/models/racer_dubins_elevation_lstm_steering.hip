/**
 * racer_dubins_elevation_lstm_steering.hip — registered instantiation(s) of libmppi_amd.so: the elevation-map RACER Dubins
 * car with the LSTM steering column (LSTMHelper I = 4, H = 4, output MLP {8, 20, 1} by default) + QuadraticCost, Gaussian
 * and colored-noise samplers.
 *
 * The analogue of the reference's include/mppi/instantiations/ + src/controllers/ (explicit template instantiations
 * compiled into shared libraries, e.g. src/controllers/cartpole/cartpole_mppi.cu:30-42).  One translation unit per
 * model, so a new or changed model recompiles alone (buildlib.py compiles the units in parallel).
 *
 * Block shapes: (64, 4) — the default — runs a rollout on four replica lanes that share out the hidden units of the LSTM,
 * the neurons of the output network, the wheels of the static settling and the rows of the covariance update
 * (RacerDubinsElevationLSTMSteeringQuad; fused and role-pipelined kernel; default network shape only).  BY == 1: a rollout
 * is one lane; the default network runs on registers (lstm_registers.hpp), any other shape on LSTMHelper's LDS contract.
 * One-lane shapes use the fused rollout kernel only: LSTMHelper::initialize() is a whole-block load with a block barrier,
 * which the role-pipelined kernel's dynamics waves cannot execute on their own.  For the same reason the trajectory pass after the
 * iterations (finalize kernel) and the single model step run the two-lane contract form (FIN_BY = 2: every thread of
 * the block takes part in initialize()), not the one-lane-of-a-wave form of the analytic models.
 * Robust MPPI (RMPPI = true): both of its kernels run the four-lanes-per-rollout form.
 */
#include "mppi_amd/engine/model_registry.hpp"
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#include "mppi_amd/sampling_distributions/colored_noise.hpp"
#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation_lstm_steering.hpp"
#include "mppi_amd/cost_functions/quadratic_cost/quadratic_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

using SteeringCost = QuadraticCost<RacerDubinsElevationLSTMSteering, /*SKIP_ZERO_COEFF=*/true>;
using RacerLSTMSteeringModel =
    ModelT<RacerDubinsElevationLSTMSteering, SteeringCost,
           sampling_distributions::GaussianDistribution<RacerDubinsElevationParams>,
           Shapes<Shape<64, 1, 1>, Shape<32, 1, 1>, Shape<64, 1, 2>>, /*FIN_BY=*/2,
           /* four lanes per rollout: a wheel, a covariance row, a hidden unit and five MLP neurons each */
           RacerDubinsElevationLSTMSteeringQuad, Shapes<Shape<64, 4, 1>, Shape<64, 4, 2>>, /*PIPELINE=*/false, /*RMPPI=*/true>;
using RacerLSTMSteeringColoredModel =
    ModelT<RacerDubinsElevationLSTMSteering, SteeringCost,
           sampling_distributions::ColoredNoiseDistribution<RacerDubinsElevationParams>, Shapes<Shape<64, 1, 1>>,
           /*FIN_BY=*/2, RacerDubinsElevationLSTMSteeringQuad, Shapes<Shape<64, 4, 1>>, /*PIPELINE=*/false>;
MPPI_REGISTER_MODEL("racer_dubins_elevation_lstm_steering", MPPI_SAMPLER_GAUSSIAN, RacerLSTMSteeringModel, 64, 4)
MPPI_REGISTER_MODEL("racer_dubins_elevation_lstm_steering", MPPI_SAMPLER_COLORED, RacerLSTMSteeringColoredModel, 64, 4)
