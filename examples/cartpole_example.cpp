/**
 * cartpole_example.cpp — the reference's examples/cartpole_example.cu:1-101 on the MI355X engine: Vanilla MPPI swinging
 * up a cart-pole, K = 2048 rollouts, T = 100, same cost weights, control range, sampler and loop structure
 * (computeControl -> take the first control -> step the model -> slideControlSequence).
 * Host-only C++ over the C ABI: g++ -std=c++11 -Iinclude examples/cartpole_example.cpp -Lmppi-generic_amd/lib -lmppi_amd
 * Usage: cartpole_example [steps (default 5000)]
 */
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "mppi_amd/controllers.hpp"

int main(int argc, char** argv)
{
  const int time_horizon = argc > 1 ? atoi(argv[1]) : 5000;
  const float dt = 0.02f, lambda = 0.25f, alpha = 0.0f;
  const int max_iter = 1, num_timesteps = 100, num_rollouts = 2048;
  try
  {
    mppi_amd::VanillaMPPIController controller("cartpole", num_rollouts, num_timesteps, dt, max_iter, lambda, alpha);
    controller.setDynamicsParams(mppi_cartpole_dynamics_params{ 1.0f, 1.0f, 1.0f });
    controller.setControlRanges({ -5.0f, 5.0f });
    mppi_cartpole_cost_params cost{};
    cost.control_cost_coeff[0] = 0;
    cost.discount = 1.0f;
    cost.cart_position_coeff = 50;
    cost.pole_angle_coeff = 200;
    cost.cart_velocity_coeff = 10;
    cost.pole_angular_velocity_coeff = 1;
    cost.terminal_cost_coeff = 0;
    cost.desired_terminal_state[0] = 20;
    cost.desired_terminal_state[1] = 0;
    cost.desired_terminal_state[2] = (float)M_PI;
    cost.desired_terminal_state[3] = 0;
    controller.setCostParams(cost);
    controller.setSamplingParams({ 5.0f });

    std::vector<float> current_state(4, 0.0f);
    const auto time_start = std::chrono::steady_clock::now();
    for (int i = 0; i < time_horizon; ++i)
    {
      controller.computeControl(current_state, 1);
      const std::vector<float> u_seq = controller.getControlSeq();
      std::vector<float> control(u_seq.begin(), u_seq.begin() + 1);
      controller.modelStep(current_state, control, dt);  // enforceConstraints + step
      if (i % 50 == 0)
        printf("Current Time: %f    Current Baseline Cost: %f    state: %f %f %f %f\n", i * dt, controller.getBaselineCost(),
               current_state[0], current_state[1], current_state[2], current_state[3]);
      controller.slideControlSequence(1);
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - time_start).count();
    printf("The elapsed time is: %f milliseconds (%.1f us per control step)\n", ms, ms * 1e3 / time_horizon);
    // the controller must have driven the cart to its goal position (desired_terminal_state[0] = 20)
    const bool at_goal = std::fabs(current_state[0] - 20.0f) < 3.0f && std::isfinite(current_state[2]);
    printf("final cart position %.3f (goal 20), pole angle %.3f rad: %s\n", current_state[0], current_state[2],
           at_goal ? "AT GOAL" : "not at goal");
    return (time_horizon >= 400 && !at_goal) ? 2 : 0;
  }
  catch (const mppi_amd::Error& e)
  {
    fprintf(stderr, "mppi error %d: %s\n", (int)e.status, e.what());
    return 1;
  }
}
