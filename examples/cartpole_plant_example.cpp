/**
 * cartpole_plant_example.cpp — the cart-pole of examples/cartpole_example.cpp driven through the BasePlant-style
 * real-time wrapper (include/mppi_amd/plant.hpp; reference: include/mppi/core/base_plant.hpp) in two ways:
 *   1. SimulatedPlant::runSimulation — single-threaded, robot time = simulated time;
 *   2. BasePlant::runControlLoop on its own thread while the main thread plays the state estimator (updateState at
 *      50 Hz of robot time) — the arrangement of a ROS plant.
 * Host-only C++: g++ -std=c++11 -pthread -Iinclude examples/cartpole_plant_example.cpp -Lmppi-generic_amd/lib -lmppi_amd
 * Usage: cartpole_plant_example [ticks (default 600)]
 */
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "mppi_amd/plant.hpp"

using Controller = mppi_amd::VanillaMPPIController;

static std::shared_ptr<Controller> makeController()
{
  auto c = std::make_shared<Controller>("cartpole", 2048, 100, 0.02f, 1, 0.25f, 0.0f);
  c->setDynamicsParams(mppi_cartpole_dynamics_params{ 1.0f, 1.0f, 1.0f });
  c->setControlRanges({ -5.0f, 5.0f });
  mppi_cartpole_cost_params cost{};
  cost.discount = 1.0f;
  cost.cart_position_coeff = 50;
  cost.pole_angle_coeff = 200;
  cost.cart_velocity_coeff = 10;
  cost.pole_angular_velocity_coeff = 1;
  cost.desired_terminal_state[0] = 20;
  cost.desired_terminal_state[2] = (float)M_PI;
  c->setCostParams(cost);
  c->setSamplingParams({ 5.0f });
  return c;
}

int main(int argc, char** argv)
{
  const int ticks = argc > 1 ? atoi(argv[1]) : 600;
  try
  {
    // 1. single-threaded simulated plant, optimisation stride 1 and 2
    for (int stride = 1; stride <= 2; stride++)
    {
      mppi_amd::SimulatedPlant<Controller> plant(makeController(), 50, stride, std::vector<float>(4, 0.0f));
      const std::vector<float>& x = plant.runSimulation(ticks);
      printf("simulated plant, stride %d: %d iterations, last stride %d, %d controls published, avg optimise %.3f ms, "
             "final state %.3f %.3f %.3f %.3f\n",
             stride, plant.getNumIter(), plant.getLastOptimizationStride(), plant.numPublished(),
             plant.getAvgOptimizationTime(), x[0], x[1], x[2], x[3]);
      if (stride == 1 && !(std::fabs(x[0] - 20.0f) < 3.0f))
      {
        printf("not at goal\n");
        return 2;
      }
    }
    // 2. control loop on its own thread, the main thread is the state estimator
    mppi_amd::SimulatedPlant<Controller> plant(makeController(), 50, 1, std::vector<float>(4, 0.0f));
    std::atomic<bool> alive(true);
    std::thread loop([&] { plant.runControlLoop(&alive); });
    std::vector<float> x(4, 0.0f);
    plant.updateState(x, 0.0);
    for (int i = 1; i <= 100; i++)
    {
      while (plant.getNumIter() < i)  // wait for the optimisation that uses the previous state
        std::this_thread::sleep_for(std::chrono::microseconds(50));
      plant.stepSimulation();
    }
    alive.store(false);
    loop.join();
    printf("threaded plant: %d iterations, avg loop %.3f ms\n", plant.getNumIter(), plant.getAvgLoopTime());
    printf("PLANT OK\n");
  }
  catch (const mppi_amd::Error& e)
  {
    fprintf(stderr, "error %d: %s\n", (int)e.status, e.what());
    return 1;
  }
  return 0;
}
