/**
 * racer_dubins_elevation_suspension.hip — registered instantiation(s) of libmppi_amd.so: the elevation-map RACER Dubins car
 * with LSTM steering and a spring / damper suspension (24 states; terrain heights + terrain normals maps) + QuadraticCost,
 * Gaussian and colored-noise samplers.
 *
 * The analogue of the reference's include/mppi/instantiations/ + src/controllers/ (explicit template instantiations
 * compiled into shared libraries, e.g. src/controllers/cartpole/cartpole_mppi.cu:30-42).  One translation unit per
 * model, so a new or changed model recompiles alone (buildlib.py compiles the units in parallel).
 *
 * Block shapes: (64, 4) — the default — runs a rollout on four replica lanes that share out the wheels of the suspension,
 * the hidden units and neurons of the steering network and the rows of the covariance update
 * (RacerDubinsElevationSuspensionQuad; default network shape only).  Two systems (Tube): (64, 4, 2) or (32, 4, 2).
 * History of the (64, 4, 2) block (512 threads, i.e. at most 256 registers per lane): while every lane kept its own copy of
 * the steering weights (81 registers; 1032 B of scratch per lane, ~200 spilled values of the step) that instantiation
 * returned NaN costs for injected noise at -O3 and correct ones at -O2; since the weights live in DPP rows (lstm_quad.hpp)
 * the kernel has no scratch and is correct at -O3.  Round 3 looked for the cause in the OLD code objects (038afa8 rebuilt at
 * -O3 and -O2): no wait-state violation around any DPP, ds_bpermute, v_readlane / v_writelane (tools/dpp_hazard_lint.py,
 * extended for that, 25 418 cross-lane instructions), the two builds differ by four flat loads of spilled values — nothing
 * that names an instruction, so the evidence for the current form is empirical: tools/soak_four_lane.py (every four-lane
 * model x one and two systems x injected noise, 10^4 launches, 0 non-finite costs, results reproduced bit for bit) and the
 * oracle comparison of the shape at full size (tests/test_full_size_parity.py).  BY == 1: one lane per rollout; fused
 * rollout kernel (the LDS fallback of the steering network needs a block barrier in initializeDynamics, see
 * racer_dubins_elevation_lstm_steering.hip).
 * Robust MPPI (RMPPI = true): both of its kernels run the four-lanes-per-rollout form.
 */
#include "mppi_amd/engine/model_registry.hpp"
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#include "mppi_amd/sampling_distributions/colored_noise.hpp"
#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation_suspension.hpp"
#include "mppi_amd/cost_functions/quadratic_cost/quadratic_cost.hpp"

using namespace mppi;
using namespace mppi::engine;

using SuspensionCost = QuadraticCost<RacerDubinsElevationSuspension, /*SKIP_ZERO_COEFF=*/true>;
using RacerSuspensionModel =
    ModelT<RacerDubinsElevationSuspension, SuspensionCost,
           sampling_distributions::GaussianDistribution<RacerDubinsElevationSuspensionParams>,
           Shapes<Shape<64, 1, 1>, Shape<32, 1, 1>, Shape<64, 1, 2>>, /*FIN_BY=*/2,
           /* four lanes per rollout: a wheel, a covariance row, a hidden unit and five MLP neurons each */
           RacerDubinsElevationSuspensionQuad, Shapes<Shape<64, 4, 1>, Shape<64, 4, 2>, Shape<32, 4, 2>>, /*PIPELINE=*/false,
           /*RMPPI=*/true>;
using RacerSuspensionColoredModel =
    ModelT<RacerDubinsElevationSuspension, SuspensionCost,
           sampling_distributions::ColoredNoiseDistribution<RacerDubinsElevationSuspensionParams>, Shapes<Shape<64, 1, 1>>,
           /*FIN_BY=*/2, RacerDubinsElevationSuspensionQuad, Shapes<Shape<64, 4, 1>>, /*PIPELINE=*/false>;
MPPI_REGISTER_MODEL("racer_dubins_elevation_suspension", MPPI_SAMPLER_GAUSSIAN, RacerSuspensionModel, 64, 4)
MPPI_REGISTER_MODEL("racer_dubins_elevation_suspension", MPPI_SAMPLER_COLORED, RacerSuspensionColoredModel, 64, 4)
