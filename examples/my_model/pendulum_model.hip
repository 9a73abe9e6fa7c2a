/**
 * An OUT-OF-TREE model: a user's own Dynamics and Cost classes, instantiated in the user's own translation unit and
 * registered with the engine — what the reference's user does in e.g. src/controllers/cartpole/cartpole_mppi.cu:30-42
 * (explicit instantiation of the controller templates with their classes, linked as a library of its own).
 *
 * Build (no part of libmppi_amd.so is recompiled; ~10 s):
 *     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I<repo>/include \
 *           examples/my_model/pendulum_model.hip -o libpendulum_model.so
 * Use:  mppi_load_plugin("libpendulum_model.so");  then  mppi_create() with cfg.model = "user_pendulum"
 *       (or link the library next to libmppi_amd.so: its static initialiser registers the model at load time).
 *
 * The classes follow the plugin contract of include/mppi_amd/plugin/ (== the reference's Dynamics / Cost CRTP bases):
 * a params struct with State / Control / Output index enums, computeDynamics(), computeStateCost(), terminalCost().
 */
#include "mppi_amd/engine/model_registry.hpp"
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#include "mppi_amd/plugin/dynamics.hpp"
#include "mppi_amd/plugin/cost.hpp"

struct PendulumParams : public DynamicsParams
{
  enum class StateIndex : int
  {
    THETA = 0,
    THETA_DOT,
    NUM_STATES
  };
  enum class ControlIndex : int
  {
    TORQUE = 0,
    NUM_CONTROLS
  };
  enum class OutputIndex : int
  {
    THETA = 0,
    THETA_DOT,
    NUM_OUTPUTS
  };
  float mass = 1.0f;
  float length = 1.0f;
  float damping = 0.1f;
  float gravity = 9.81f;
};

using namespace MPPI_internal;

/** damped pendulum, theta = 0 hanging down:  m l^2 theta'' = u - b theta' - m g l sin(theta) */
class PendulumDynamics : public Dynamics<PendulumDynamics, PendulumParams>
{
public:
  /** no __syncthreads() in the per-step methods below (mppi::lane_sync() where the reference's step has its barriers): the model
   *  may run on the role-separated kernels — without this line PIPELINE = true is refused (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  using PARENT_CLASS = Dynamics<PendulumDynamics, PendulumParams>;
  PendulumDynamics(hipStream_t stream = nullptr) : PARENT_CLASS(stream)
  {
  }
  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta_s = nullptr)
  {
    float s, c;
    mppi::det::sincos(state[0], &s, &c);
    const float inertia = this->params_.mass * this->params_.length * this->params_.length;
    const float gravity_torque = this->params_.mass * this->params_.gravity * this->params_.length * s;
    state_der[0] = state[1];
    state_der[1] = (control[0] - this->params_.damping * state[1] - gravity_torque) / inertia;
  }
};

struct PendulumCostParams : public CostParams<1>
{
  float angle_coeff = 10.0f;
  float velocity_coeff = 0.1f;
  float terminal_coeff = 0.0f;
  float goal_angle = 3.14159265f;  // upright
};

/** angle_coeff (1 - cos(theta - goal)) + velocity_coeff theta'^2 : periodic in the angle, no wrap needed */
class PendulumCost : public Cost<PendulumCost, PendulumCostParams, PendulumParams>
{
public:
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  PendulumCost(hipStream_t stream = nullptr)
  {
    bindToStream(stream);
  }
  __device__ inline float stateError(const float* y) const
  {
    float s, c;
    mppi::det::sincos(y[0] - params_.goal_angle, &s, &c);
    return params_.angle_coeff * (1.0f - c) + params_.velocity_coeff * (y[1] * y[1]);
  }
  __device__ inline float computeStateCost(float* y, int timestep = 0, float* theta_c = nullptr, int* crash_status = nullptr)
  {
    return stateError(y);
  }
  __device__ inline float terminalCost(float* y, float* theta_c)
  {
    return stateError(y) * params_.terminal_coeff;
  }
};

using namespace mppi::engine;
using PendulumSampler = mppi::sampling_distributions::GaussianDistribution<PendulumParams>;
/* block shapes: (64, 1, 1) one lane per rollout — also what the role-pipelined kernel runs on — and (64, 1, 2) for Tube */
using PendulumModel = ModelT<PendulumDynamics, PendulumCost, PendulumSampler, Shapes<Shape<64, 1, 1>, Shape<64, 1, 2>>,
                             /*FIN_BY=*/1, void, Shapes<>, /*PIPELINE=*/true>;
MPPI_REGISTER_MODEL("user_pendulum", MPPI_SAMPLER_GAUSSIAN, PendulumModel, 64, 1)
