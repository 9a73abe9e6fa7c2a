/**
 * The reference's way of setting up a controller — plugin objects + the TEMPLATED controller class — on this engine.
 *
 * Written against the reference's include paths and class names (what examples/cartpole_example.cu of ACDSLab/MPPI-Generic
 * does: model, cost and sampler objects, DDPFeedback<Dyn, T>, VanillaMPPIController<Dyn, Cost, FB, T, K>(model, cost, fb,
 * sampler, dt, max_iter, lambda, alpha), a loop of computeControl / model->step / slideControlSequence); the only host type
 * that differs is the vector class (mppi::host::Array instead of an Eigen column, same layout).
 *
 * Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I<repo>/include examples/templated_cartpole.hip \
 *               -L<repo>/mppi-generic_amd/lib -lmppi_amd -Wl,-rpath,<repo>/mppi-generic_amd/lib -o templated_cartpole
 * Run:    ./templated_cartpole [steps] [lanes per rollout]    prints the state every 50 steps and a checksum of the last control
 *         sequence.  lanes per rollout = dynamics_rollout_dim_.y: 1 (default; the role-pipelined kernels, one lane per rollout) or
 *         4 — the reference example's own dim3(64, 4, 1) (examples/cartpole_example.cu:50-51): the LDS + barrier form of the
 *         plugin contract on the fused kernel; same trajectory costs, bit for bit
 */
#include <mppi/instantiations/cartpole_mppi/cartpole_mppi.cuh>

#include <chrono>
#include <cstdio>
#include <cstdlib>

using Sampler = mppi::sampling_distributions::GaussianDistribution<CartpoleDynamics::DYN_PARAMS_T>;
constexpr int HORIZON = 100;
constexpr int ROLLOUTS = 2048;
using Feedback = DDPFeedback<CartpoleDynamics, HORIZON>;
using CartpoleMPPI = VanillaMPPIController<CartpoleDynamics, CartpoleQuadraticCost, Feedback, HORIZON, ROLLOUTS>;

int main(int argc, char** argv)
{
  const int steps = argc > 1 ? atoi(argv[1]) : 500;
  const int lanes = argc > 2 ? atoi(argv[2]) : 1;

  CartpoleDynamics model(1.0f, 1.0f, 1.0f);  // cart mass, pole mass, pole length
  model.control_rngs_->x = -5;
  model.control_rngs_->y = 5;

  CartpoleQuadraticCost cost;
  CartpoleQuadraticCostParams cost_params;
  cost_params.cart_position_coeff = 50;
  cost_params.pole_angle_coeff = 200;
  cost_params.cart_velocity_coeff = 10;
  cost_params.pole_angular_velocity_coeff = 1;
  cost_params.control_cost_coeff[0] = 0;
  cost_params.terminal_cost_coeff = 0;
  cost_params.desired_terminal_state[0] = 20;
  cost_params.desired_terminal_state[1] = 0;
  cost_params.desired_terminal_state[2] = M_PI;
  cost_params.desired_terminal_state[3] = 0;
  cost.setParams(cost_params);

  auto sampler_params = Sampler::SAMPLING_PARAMS_T();
  for (int i = 0; i < CartpoleDynamics::CONTROL_DIM; i++)
    sampler_params.std_dev[i] = 5.0f;
  Sampler sampler(sampler_params);

  const float dt = 0.02f, lambda = 0.25f, alpha = 0.0f;
  const int max_iter = 1;
  Feedback fb_controller(&model, dt);

  CartpoleMPPI controller(&model, &cost, &fb_controller, &sampler, dt, max_iter, lambda, alpha);
  auto controller_params = controller.getParams();
  controller_params.dynamics_rollout_dim_ = dim3(64, lanes, 1);
  controller_params.cost_rollout_dim_ = dim3(64, lanes, 1);
  controller.setParams(controller_params);

  CartpoleDynamics::state_array x = CartpoleDynamics::state_array::Zero(), x_next = x, xdot = x;
  CartpoleDynamics::output_array y = CartpoleDynamics::output_array::Zero();

  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < steps; i++)
  {
    controller.computeControl(x, 1);
    CartpoleDynamics::control_array u = controller.getControlSeq().block(0, 0, CartpoleDynamics::CONTROL_DIM, 1);
    model.enforceConstraints(x, u);
    model.step(x, x_next, xdot, u, y, (float)i, dt);
    x = x_next;
    if (i % 50 == 0)
    {
      printf("t = %5.2f s   baseline cost %10.3f   ", i * dt, controller.getBaselineCost());
      model.printState(x.data());
    }
    controller.slideControlSequence(1);
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  double sum = 0.0;
  const auto u_seq = controller.getControlSeq();
  for (int t = 0; t < HORIZON; t++)
    sum += u_seq(0, t);
  printf("%s: %d control steps in %.1f ms, pole angle %.4f rad, checksum %.6f\n", controller.getControllerName().c_str(), steps,
         ms, x[2], sum);
  return 0;
}
