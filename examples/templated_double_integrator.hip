/**
 * The four templated controller classes on the double integrator, set up the way the reference's
 * examples/double_integrator_CORL2020.cu does it: Vanilla and Tube MPPI on the circle cost, Robust MPPI on
 * DoubleIntegratorRobustCost (crash_cost = 100, value_function_threshold = 20), K = 1024, T = 50, dt = 0.02, lambda = 2 —
 * plus Colored MPPI.  The DDP gains the reference computes with its own solver are supplied through
 * DDPFeedback::setFeedbackGains (here: a fixed proportional-derivative gain for every time step).
 *
 * Build: see examples/templated_cartpole.hip.   Run: ./templated_double_integrator [steps]
 */
#include <mppi/dynamics/double_integrator/di_dynamics.cuh>
#include <mppi/cost_functions/double_integrator/double_integrator_circle_cost.cuh>
#include <mppi/cost_functions/double_integrator/double_integrator_robust_cost.cuh>
#include <mppi/controllers/MPPI/mppi_controller.cuh>
#include <mppi/controllers/Tube-MPPI/tube_mppi_controller.cuh>
#include <mppi/controllers/R-MPPI/robust_mppi_controller.cuh>
#include <mppi/controllers/ColoredMPPI/colored_mppi_controller.cuh>
#include <mppi/feedback_controllers/DDP/ddp.cuh>

#include <cstdio>
#include <cstdlib>

using Dyn = DoubleIntegratorDynamics;
using SCost = DoubleIntegratorCircleCost;
using RCost = DoubleIntegratorRobustCost;
constexpr int num_timesteps = 50;
constexpr int num_rollouts = 1024;
using Feedback = DDPFeedback<Dyn, num_timesteps>;
using Sampler = mppi::sampling_distributions::GaussianDistribution<Dyn::DYN_PARAMS_T>;
using ColoredSampler = mppi::sampling_distributions::ColoredNoiseDistribution<Dyn::DYN_PARAMS_T>;

const float dt = 0.02f, lambda = 2.0f, alpha = 0.0f;
const int max_iter = 1;

template <class CONTROLLER_T>
static double runLoop(CONTROLLER_T& controller, Dyn& model, int steps, bool robust)
{
  Dyn::state_array x = { 2.0f, 0.0f, 0.0f, 1.0f }, x_next = x, xdot = Dyn::state_array::Zero();
  Dyn::output_array y = Dyn::output_array::Zero();
  double checksum = 0.0;
  for (int i = 0; i < steps; i++)
  {
    if constexpr (std::is_same<CONTROLLER_T, RobustMPPIController<Dyn, RCost, Feedback, num_timesteps, num_rollouts>>::value)
      controller.updateImportanceSamplingControl(x, 1);
    (void)robust;
    controller.computeControl(x, 1);
    Dyn::control_array u = controller.getControlSeq().col(0);
    model.enforceConstraints(x, u);
    model.step(x, x_next, xdot, u, y, (float)i, dt);
    x = x_next;
    checksum += u[0] + 2.0 * u[1];
    controller.slideControlSequence(1);
  }
  printf("%-13s after %d steps: position (%.4f, %.4f), radius %.4f, free energy %.4f, checksum %.6f\n",
         controller.getControllerName().c_str(), steps, x[0], x[1], sqrtf(x[0] * x[0] + x[1] * x[1]),
         controller.getFreeEnergyStatistics().real_sys.free_energy_mean, checksum);
  return checksum;
}

int main(int argc, char** argv)
{
  const int steps = argc > 1 ? atoi(argv[1]) : 100;
  Dyn model;
  auto sampler_params = Sampler::SAMPLING_PARAMS_T();
  for (int i = 0; i < Dyn::CONTROL_DIM; i++)
    sampler_params.std_dev[i] = 1.0f;
  Sampler sampler(sampler_params);
  Feedback fb_controller(&model, dt);
  // u_fb = K^T (x - x*): position error pulls, velocity error damps — gains[t][state][control]
  std::vector<float> gains((size_t)num_timesteps * Dyn::STATE_DIM * Dyn::CONTROL_DIM, 0.0f);
  for (int t = 0; t < num_timesteps; t++)
    for (int c = 0; c < 2; c++)
    {
      gains[((size_t)t * Dyn::STATE_DIM + c) * Dyn::CONTROL_DIM + c] = -20.0f;
      gains[((size_t)t * Dyn::STATE_DIM + 2 + c) * Dyn::CONTROL_DIM + c] = -8.0f;
    }
  fb_controller.setFeedbackGains(gains, /*accumulate_all_states=*/true);

  SCost circle_cost;
  RCost robust_cost;
  auto robust_params = robust_cost.getParams();
  robust_params.crash_cost = 100;
  robust_cost.setParams(robust_params);

  VanillaMPPIController<Dyn, SCost, Feedback, num_timesteps, num_rollouts> vanilla(&model, &circle_cost, &fb_controller, &sampler,
                                                                                  dt, max_iter, lambda, alpha);
  runLoop(vanilla, model, steps, false);

  TubeMPPIController<Dyn, SCost, Feedback, num_timesteps, num_rollouts> tube(&model, &circle_cost, &fb_controller, &sampler, dt,
                                                                            max_iter, lambda, alpha);
  tube.setNominalThreshold(20.0f);
  runLoop(tube, model, steps, false);

  const float value_function_threshold = 20.0f;
  RobustMPPIController<Dyn, RCost, Feedback, num_timesteps, num_rollouts> robust(
      &model, &robust_cost, &fb_controller, &sampler, dt, max_iter, lambda, alpha, value_function_threshold);
  runLoop(robust, model, steps, true);

  ColoredSampler colored_sampler;
  colored_sampler.setParams(sampler_params);
  colored_sampler.exponents_[0] = 1.0f;
  colored_sampler.exponents_[1] = 1.0f;
  ColoredMPPIController<Dyn, SCost, Feedback, num_timesteps, num_rollouts> colored(&model, &circle_cost, &fb_controller,
                                                                                  &colored_sampler, dt, max_iter, lambda, alpha);
  runLoop(colored, model, steps, false);
  return 0;
}
