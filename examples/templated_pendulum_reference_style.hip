/**
 * A model written like the reference's own — Dynamics::step with its two __syncthreads() (dynamics/dynamics.cu:130-142),
 * threadIdx.y-strided loops, no engine-specific line in either class (my_model/pendulum_reference_style.cuh) — driven through
 * the reference's TEMPLATED controller class, with the reference example's own rollout block shape (64, 4, 1)
 * (examples/cartpole_example.cu:50-51) and with (64, 1, 1).
 *
 * What this file is there to show (and tests/test_templated_controllers.py to run on the GPU): such a model FINISHES.  The
 * templated classes choose the role-pipelined kernels only for plugin classes that declare MPPI_BARRIER_FREE_STEP
 * (mppi_amd/plugin/parallel_utils.hpp); these say nothing, so VanillaMPPIController<...> instantiates the FUSED rollout kernel
 * for them — every thread of the block reaches every plugin call, the barriers complete — and dynamics_rollout_dim_.y = 4 runs
 * the plugin contract's LDS form (state in shared memory, four lanes per rollout), as upstream.
 *
 * Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I<repo>/include -I<repo>/examples \
 *               examples/templated_pendulum_reference_style.hip -L<repo>/mppi-generic_amd/lib -lmppi_amd \
 *               -Wl,-rpath,<repo>/mppi-generic_amd/lib -o templated_pendulum_reference_style
 * Run:    ./templated_pendulum_reference_style [steps] [lanes per rollout: 1 | 4 | anything else is refused]
 */
#include "my_model/pendulum_reference_style.cuh"
#include <mppi/controllers/MPPI/mppi_controller.cuh>
#include <mppi/controllers/Tube-MPPI/tube_mppi_controller.cuh>
#include <mppi/feedback_controllers/DDP/ddp.cuh>

#include <cstdio>
#include <cstdlib>

using Sampler = mppi::sampling_distributions::GaussianDistribution<RefPendulumParams>;
constexpr int HORIZON = 60;
constexpr int ROLLOUTS = 1024;
using Feedback = DDPFeedback<RefPendulumDynamics, HORIZON>;
using PendulumMPPI = VanillaMPPIController<RefPendulumDynamics, RefPendulumCost, Feedback, HORIZON, ROLLOUTS>;
using PendulumTube = TubeMPPIController<RefPendulumDynamics, RefPendulumCost, Feedback, HORIZON, ROLLOUTS>;

template <class CONTROLLER_T>
static int run(const char* label, int steps, int lanes)
{
  RefPendulumDynamics model;
  model.control_rngs_->x = -2;
  model.control_rngs_->y = 2;
  RefPendulumCost cost;
  auto sampler_params = Sampler::SAMPLING_PARAMS_T();
  sampler_params.std_dev[0] = 1.0f;
  Sampler sampler(sampler_params);
  const float dt = 0.02f, lambda = 1.0f, alpha = 0.0f;
  Feedback fb_controller(&model, dt);
  CONTROLLER_T controller(&model, &cost, &fb_controller, &sampler, dt, /*max_iter=*/1, lambda, alpha);
  auto controller_params = controller.getParams();
  controller_params.dynamics_rollout_dim_ = dim3(64, lanes, 1);
  controller_params.cost_rollout_dim_ = dim3(64, lanes, 1);
  controller.setParams(controller_params);

  RefPendulumDynamics::state_array x = RefPendulumDynamics::state_array::Zero();
  x[0] = 0.3f;  // off the hanging equilibrium (where both directions cost the same and MPPI's mean torque is zero)
  double sum = 0.0;
  try
  {
    for (int i = 0; i < steps; i++)
    {
      controller.computeControl(x, 1);
      const auto u_seq = controller.getControlSeq();
      // the plant: the same Euler step in float64 on the host (this model has no host methods — nothing on the path needs them)
      const double th = x[0], om = x[1], u = u_seq(0, 0);
      x[0] = (float)(th + om * dt);
      x[1] = (float)(om + (u - 0.1 * om - 9.81 * std::sin(th)) * dt);
      controller.slideControlSequence(1);
    }
    const auto u_seq = controller.getControlSeq();
    for (int t = 0; t < HORIZON; t++)
      sum += u_seq(0, t);
  }
  catch (const mppi_amd::Error& e)
  {
    printf("%s lanes %d: refused with status %d: %s\n", label, lanes, (int)e.status, e.what());
    return 2;
  }
  printf("%s lanes %d: %d control steps, angle %.5f rad, velocity %.5f rad/s, baseline %.5f, checksum %.6f\n", label, lanes, steps,
         x[0], x[1], controller.getBaselineCost(), sum);
  return 0;
}

int main(int argc, char** argv)
{
  const int steps = argc > 1 ? atoi(argv[1]) : 100;
  const int lanes = argc > 2 ? atoi(argv[2]) : 4;
  int rc = run<PendulumMPPI>("Vanilla MPPI", steps, lanes);
  if (rc == 0)
    rc = run<PendulumTube>("Tube MPPI", steps, lanes);
  return rc;
}
