/**
 * BicycleSlipLSTM — vehicle model with analytic kinematics and an LSTM for the dynamic (slip) states: the
 * "LSTM bicycle-slip dynamics" of BASELINE.json config 5.
 *
 * The reference snapshot has no such class any more (its tests still name bicycle_slip_{kinematic,hybrid}*.npz,
 * tests/templated_headers/racer_test_networks.h.in:13-16, but dynamics/bicycle_slip/ is purely parametric; SURVEY.md
 * Appendix C).  This plugin is therefore built from the two reference pieces that DO define the path:
 *   - the structure of NeuralNetModel (include/mppi/dynamics/autorally/ar_nn_model.cu:122-178): state
 *     [x, y, yaw, roll, vx_body, vy_body, yaw_rate], control [steering, throttle]; xdot[0..2] = planar kinematics,
 *     xdot[3..6] = network([roll, vx, vy, yaw_rate, steering, throttle]);
 *   - LSTMHelper in the rollout loop the way the RACER models use it (include/mppi/dynamics/racer_dubins/
 *     racer_dubins_elevation_lstm_steering.cu:117-167): initializeDynamics() seeds every rollout's (h, c) from the
 *     helper's initial state, step() feeds the network input and adds the network output to the state derivative.
 * Network: LSTM(I = 6, H = 16) + output MLP {22, 32, 4}  (1408 + 832 MAC per step).
 * Same state layout as the AutoRally model, so ARStandardCost applies unchanged.
 */
#ifndef MPPI_AMD_BICYCLE_SLIP_LSTM_HPP_
#define MPPI_AMD_BICYCLE_SLIP_LSTM_HPP_

#include "mppi_amd/plugin/dynamics.hpp"
#include "mppi_amd/utils/nn_helpers/lstm_helper.hpp"
#include "mppi_amd/utils/nn_helpers/lstm_mfma.hpp"
#include "mppi_amd/utils/nn_helpers/lstm_wave.hpp"

struct BicycleSlipLSTMParams : public DynamicsParams
{
  enum class StateIndex : int
  {
    POS_X = 0,
    POS_Y,
    YAW,
    ROLL,
    BODY_VEL_X,
    BODY_VEL_Y,
    YAW_RATE,
    NUM_STATES
  };
  enum class ControlIndex : int
  {
    STEERING = 0,
    THROTTLE,
    NUM_CONTROLS
  };
  enum class OutputIndex : int
  {
    POS_X = 0,
    POS_Y,
    YAW,
    ROLL,
    BODY_VEL_X,
    BODY_VEL_Y,
    YAW_RATE,
    FILLER_1,
    NUM_OUTPUTS
  };
};

namespace bicycle_slip_lstm
{
constexpr int LSTM_INPUT = 6, LSTM_HIDDEN = 16, MLP_HIDDEN = 32, NET_OUTPUT = 4;
}

class BicycleSlipLSTM : public MPPI_internal::Dynamics<BicycleSlipLSTM, BicycleSlipLSTMParams>
{
public:
  using PARENT_CLASS = MPPI_internal::Dynamics<BicycleSlipLSTM, BicycleSlipLSTMParams>;
  static const int DYNAMICS_DIM = 4;  ///< states produced by the network (roll, vx, vy, yaw rate)

  BicycleSlipLSTM(hipStream_t stream = 0) : PARENT_CLASS(stream)
  {
    using namespace bicycle_slip_lstm;
    const int out_layers[3] = { LSTM_HIDDEN + LSTM_INPUT, MLP_HIDDEN, NET_OUTPUT };
    lstm_.setStructure(LSTM_INPUT, LSTM_HIDDEN, out_layers, 3);
    lstm_.output_nn_.split_output_sum_ = true;  // this network's output layer in every form and in the oracle (fnn_helper.hpp)
  }
  static const char* getDynamicsModelName()
  {
    return "LSTM bicycle-slip model";
  }
  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return lstm_.getGrdSharedSizeBytes();
  }
  __host__ __device__ int getBlkSharedSizeBytes() const
  {
    return lstm_.getBlkSharedSizeBytes();
  }

  __device__ inline void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    PARENT_CLASS::initializeDynamics(state, control, output, theta_s, t_0, dt);
    lstm_.initialize(theta_s);
  }

  __device__ inline void computeKinematics(float* state, float* state_der)
  {
    float s, c;
    mppi::det::sincos(state[2], &s, &c);
    state_der[0] = c * state[4] - s * state[5];
    state_der[1] = s * state[4] + c * state[5];
    state_der[2] = -state[6];
  }

  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta_s = nullptr)
  {
    const int tdy = (int)__builtin_amdgcn_workitem_id_y();
    const int bdy = (int)__builtin_amdgcn_workgroup_size_y();
    float* in = lstm_.getInputLocation(theta_s);
    for (int i = tdy; i < DYNAMICS_DIM; i += bdy)
      in[i] = state[i + (STATE_DIM - DYNAMICS_DIM)];
    for (int i = tdy; i < CONTROL_DIM; i += bdy)
      in[DYNAMICS_DIM + i] = control[i];
    mppi::lane_sync();
    float* out = lstm_.forward(nullptr, theta_s);
    for (int i = tdy; i < DYNAMICS_DIM; i += bdy)
      state_der[i + (STATE_DIM - DYNAMICS_DIM)] = out[i];
    mppi::lane_sync();
  }

  mppi::LSTMHelper lstm_;
};

/**
 * The same model with the network on the matrix cores (utils/nn_helpers/lstm_mfma.hpp): block shape (BX, 4) runs as
 * BX rollouts x 4 replicated lanes (REPLICATED_LANES, csrc/rollout_kernel.hpp); kinematics, Euler step and constraints
 * are evaluated redundantly by the four lanes on private register copies; the LSTM state lives in registers.
 */
class BicycleSlipLSTMWave;

class BicycleSlipLSTMMFMA : public MPPI_internal::Dynamics<BicycleSlipLSTMMFMA, BicycleSlipLSTMParams>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  /**
   * Helper waves of the role-pipelined Robust kernel for this model (engine/rmppi_pipeline_kernel.hpp): one sampler and
   * one cost wave per block beside the eight dynamics waves, so that the block fits the register budget of the MFMA
   * network without spills; measured 760 -> 628 us on the 16384 x 150 Robust launch (2 cost waves: 708 us),
   * profiles/r06_robust_racer_ab.json
   */
  static constexpr int MPPI_RMPPI_PIPE_SAMPLERS = 1;
  static constexpr int MPPI_RMPPI_PIPE_COSTS = 1;
  using PARENT_CLASS = MPPI_internal::Dynamics<BicycleSlipLSTMMFMA, BicycleSlipLSTMParams>;
  static const int DYNAMICS_DIM = 4;
  static constexpr int REPLICATED_LANES = 4;
  /** the form the single-trajectory re-rollout runs on (engine: finalizeRepKernel): one rollout per wave */
  using FINALIZE_FORM = BicycleSlipLSTMWave;
  using NET = mppi::LSTMMfma<bicycle_slip_lstm::LSTM_INPUT, bicycle_slip_lstm::LSTM_HIDDEN,
                             bicycle_slip_lstm::MLP_HIDDEN, bicycle_slip_lstm::NET_OUTPUT>;

  BicycleSlipLSTMMFMA(const BicycleSlipLSTM& other) : PARENT_CLASS(other.stream_)
  {
    this->params_ = other.params_;
    for (int i = 0; i < CONTROL_DIM; i++)
    {
      this->control_rngs_[i] = other.control_rngs_[i];
      this->control_deadband_[i] = other.control_deadband_[i];
      this->zero_control_[i] = other.zero_control_[i];
    }
    lstm_d_ = other.lstm_.weights_d_;
    fnn_d_ = other.lstm_.output_nn_.theta_d_;
  }
  /** block-shared LDS: the output layer's weights per lane group (lstm_mfma.hpp: w2_lds), 512 B */
  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return NET::W2_LDS_FLOATS * (int)sizeof(float);
  }
  __host__ __device__ int getBlkSharedSizeBytes() const
  {
    return 0;
  }

  __device__ inline void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    PARENT_CLASS::initializeDynamics(state, control, output, theta_s, t_0, dt);
    net_.load(lstm_d_, fnn_d_, (int)(threadIdx.x & 63), theta_s);
  }

  __device__ inline void computeKinematics(float* state, float* state_der)
  {
    float s, c;
    mppi::det::sincos(state[2], &s, &c);
    state_der[0] = c * state[4] - s * state[5];
    state_der[1] = s * state[4] + c * state[5];
    state_der[2] = -state[6];
  }

  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta_s = nullptr)
  {
    float in[bicycle_slip_lstm::LSTM_INPUT], out[bicycle_slip_lstm::NET_OUTPUT];
#pragma unroll
    for (int i = 0; i < DYNAMICS_DIM; i++)
      in[i] = state[i + (STATE_DIM - DYNAMICS_DIM)];
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      in[DYNAMICS_DIM + i] = control[i];
    net_.forward(in, out, (int)(threadIdx.x & 63));
#pragma unroll
    for (int i = 0; i < DYNAMICS_DIM; i++)
      state_der[i + (STATE_DIM - DYNAMICS_DIM)] = out[i];
  }

  const float* lstm_d_ = nullptr;
  const float* fnn_d_ = nullptr;
  NET net_;  ///< weight fragments + recurrent state: per-thread registers (the object is a by-value kernel argument)
};

/**
 * The same model for ONE rollout on a whole wave (utils/nn_helpers/lstm_wave.hpp: lane = gate row of the LSTM / neuron of the
 * output network, activations by v_readlane); REPLICATED_LANES = 64, same fma chains and activations as the other forms — the
 * trajectories are the same bits.  Used for the re-rollout of the optimised control sequence (T dependent steps).
 */
class BicycleSlipLSTMWave : public MPPI_internal::Dynamics<BicycleSlipLSTMWave, BicycleSlipLSTMParams>
{
public:
  using PARENT_CLASS = MPPI_internal::Dynamics<BicycleSlipLSTMWave, BicycleSlipLSTMParams>;
  static const int DYNAMICS_DIM = 4;
  static constexpr int REPLICATED_LANES = 64;
  using NET = mppi::LSTMWave<bicycle_slip_lstm::LSTM_INPUT, bicycle_slip_lstm::LSTM_HIDDEN,
                             bicycle_slip_lstm::MLP_HIDDEN, bicycle_slip_lstm::NET_OUTPUT>;

  BicycleSlipLSTMWave(const BicycleSlipLSTM& other) : PARENT_CLASS(other.stream_)
  {
    this->params_ = other.params_;
    for (int i = 0; i < CONTROL_DIM; i++)
    {
      this->control_rngs_[i] = other.control_rngs_[i];
      this->control_deadband_[i] = other.control_deadband_[i];
      this->zero_control_[i] = other.zero_control_[i];
    }
    lstm_d_ = other.lstm_.weights_d_;
    fnn_d_ = other.lstm_.output_nn_.theta_d_;
  }
  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return 0;
  }
  __host__ __device__ int getBlkSharedSizeBytes() const
  {
    return 0;
  }
  __device__ inline void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    PARENT_CLASS::initializeDynamics(state, control, output, theta_s, t_0, dt);
    net_.load(lstm_d_, fnn_d_, (int)(threadIdx.x & 63));
  }
  __device__ inline void computeKinematics(float* state, float* state_der)
  {
    float s, c;
    mppi::det::sincos(state[2], &s, &c);
    state_der[0] = c * state[4] - s * state[5];
    state_der[1] = s * state[4] + c * state[5];
    state_der[2] = -state[6];
  }
  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta_s = nullptr)
  {
    float in[bicycle_slip_lstm::LSTM_INPUT], out[bicycle_slip_lstm::NET_OUTPUT];
#pragma unroll
    for (int i = 0; i < DYNAMICS_DIM; i++)
      in[i] = state[i + (STATE_DIM - DYNAMICS_DIM)];
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      in[DYNAMICS_DIM + i] = control[i];
    net_.forward(in, out, (int)(threadIdx.x & 63));
#pragma unroll
    for (int i = 0; i < DYNAMICS_DIM; i++)
      state_der[i + (STATE_DIM - DYNAMICS_DIM)] = out[i];
  }

  const float* lstm_d_ = nullptr;
  const float* fnn_d_ = nullptr;
  NET net_;
};

#endif
