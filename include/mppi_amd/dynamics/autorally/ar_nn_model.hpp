/**
 * NeuralNetModel<S_DIM, C_DIM, K_DIM> — AutoRally dynamics: analytic kinematics + an MLP for the dynamic states.
 * reference: include/mppi/dynamics/autorally/ar_nn_model.cuh:22-160, ar_nn_model.cu:122-178
 *   state  [x, y, yaw, roll, vx_body, vy_body, yaw_rate], control [steering, throttle], output = state + 1 filler
 *   xdot[0..2] = kinematics(yaw, vx, vy, yaw_rate)                         (ar_nn_model.cu:122-128, cosf/sinf -> det)
 *   xdot[3..6] = FNN([roll, vx, vy, yaw_rate, steering, throttle])         (ar_nn_model.cu:130-160)
 */
#ifndef MPPI_AMD_AR_NN_MODEL_HPP_
#define MPPI_AMD_AR_NN_MODEL_HPP_

#include "mppi_amd/plugin/dynamics.hpp"
#include "mppi_amd/utils/nn_helpers/fnn_helper.hpp"
#include "mppi_amd/utils/nn_helpers/fnn_mfma.hpp"
#include "mppi_amd/utils/nn_helpers/fnn_wave.hpp"

struct NNDynamicsParams : public DynamicsParams
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

using namespace MPPI_internal;

template <int S_DIM, int C_DIM, int K_DIM>
class NeuralNetModel : public Dynamics<NeuralNetModel<S_DIM, C_DIM, K_DIM>, NNDynamicsParams>
{
public:
  using PARENT_CLASS = Dynamics<NeuralNetModel<S_DIM, C_DIM, K_DIM>, NNDynamicsParams>;
  static const int DYNAMICS_DIM = S_DIM - K_DIM;  ///< number of inputs from state

  NeuralNetModel(hipStream_t stream = 0) : PARENT_CLASS(stream)
  {
    const int layers[4] = { 6, 32, 32, 4 };  // ar_nn_model.cu:7, :17
    helper_.setStructure(layers, 4);
    helper_.split_output_sum_ = true;  // this network's output layer in every form and in the oracle (fnn_helper.hpp)
  }
  static const char* getDynamicsModelName()
  {
    return "FCN Autorally Model";
  }

  /** LDS requests are the network's (ar_nn_model.cu:8-10) */
  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return helper_.getGrdSharedSizeBytes();
  }
  __host__ __device__ int getBlkSharedSizeBytes() const
  {
    return helper_.getBlkSharedSizeBytes();
  }

  __device__ inline void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    PARENT_CLASS::initializeDynamics(state, control, output, theta_s, t_0, dt);
    helper_.initialize(theta_s);
  }

  __device__ inline void computeKinematics(float* state, float* state_der)
  {
    float s, c;
    mppi::det::sincos(state[2], &s, &c);
    state_der[0] = c * state[4] - s * state[5];
    state_der[1] = s * state[4] + c * state[5];
    state_der[2] = -state[6];  // Pose estimate actually gives the negative yaw derivative
  }

  __device__ inline void computeDynamics(float* state, float* control, float* state_der, float* theta_s = nullptr)
  {
    const int tdy = (int)__builtin_amdgcn_workitem_id_y();
    const int bdy = (int)__builtin_amdgcn_workgroup_size_y();
    float* curr_act = helper_.getInputLocation(theta_s);
    for (int i = tdy; i < DYNAMICS_DIM; i += bdy)
      curr_act[i] = state[i + (S_DIM - DYNAMICS_DIM)];
    for (int i = tdy; i < C_DIM; i += bdy)
      curr_act[DYNAMICS_DIM + i] = control[i];
    mppi::lane_sync();
    curr_act = helper_.forward(nullptr, theta_s);
    for (int i = tdy; i < DYNAMICS_DIM; i += bdy)
      state_der[i + (S_DIM - DYNAMICS_DIM)] = curr_act[i];
    mppi::lane_sync();
  }

  mppi::FNNHelper helper_;
};

/**
 * The same model with the MLP on the matrix cores (utils/nn_helpers/fnn_mfma.hpp).  Block shape (16, 4): the four
 * lanes of a rollout are the MFMA k-groups.  Everything that is not the network (kinematics, Euler step, constraints)
 * is evaluated redundantly by the four lanes on private register copies of the rollout state
 * (REPLICATED_LANES, see csrc/rollout_kernel.hpp), so no LDS slot and no barrier is involved.
 */
template <int S_DIM, int C_DIM, int K_DIM>
class NeuralNetModelWave;

template <int S_DIM, int C_DIM, int K_DIM>
class NeuralNetModelMFMA : public Dynamics<NeuralNetModelMFMA<S_DIM, C_DIM, K_DIM>, NNDynamicsParams>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  using PARENT_CLASS = Dynamics<NeuralNetModelMFMA<S_DIM, C_DIM, K_DIM>, NNDynamicsParams>;
  static const int DYNAMICS_DIM = S_DIM - K_DIM;
  static constexpr int REPLICATED_LANES = 4;
  using NET = mppi::FNNMfma<DYNAMICS_DIM + C_DIM, 32, 4>;
  /** the form the single-trajectory re-rollout runs on (engine: finalizeRepKernel): one rollout per wave, lane = neuron */
  using FINALIZE_FORM = NeuralNetModelWave<S_DIM, C_DIM, K_DIM>;

  /** shares parameters, control ranges and the weight blob with the plain model */
  NeuralNetModelMFMA(const NeuralNetModel<S_DIM, C_DIM, K_DIM>& other) : PARENT_CLASS(other.stream_)
  {
    this->params_ = other.params_;
    for (int i = 0; i < C_DIM; i++)
    {
      this->control_rngs_[i] = other.control_rngs_[i];
      this->control_deadband_[i] = other.control_deadband_[i];
      this->zero_control_[i] = other.zero_control_[i];
    }
    theta_d_ = other.helper_.theta_d_;
  }

  /** weights and activations live in registers; block-shared LDS holds a copy of the hidden layers' biases per lane group
   *  (256 B) that register-starved kernels read instead (fnn_mfma.hpp: bias_lds) */
  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return NET::BIAS_LDS_FLOATS * (int)sizeof(float);
  }
  __host__ __device__ int getBlkSharedSizeBytes() const
  {
    return 0;
  }

  __device__ inline void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    PARENT_CLASS::initializeDynamics(state, control, output, theta_s, t_0, dt);
    net_.load(theta_d_, (int)(threadIdx.x & 63), theta_s);
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
    float in[DYNAMICS_DIM + C_DIM], out[4];
#pragma unroll
    for (int i = 0; i < DYNAMICS_DIM; i++)
      in[i] = state[i + (S_DIM - DYNAMICS_DIM)];
#pragma unroll
    for (int i = 0; i < C_DIM; i++)
      in[DYNAMICS_DIM + i] = control[i];
    net_.forward(in, out, (int)(threadIdx.x & 63));
#pragma unroll
    for (int i = 0; i < DYNAMICS_DIM; i++)
      state_der[i + (S_DIM - DYNAMICS_DIM)] = out[i];
  }

  const float* theta_d_ = nullptr;
  NET net_;  ///< per-thread weight fragments: the object is passed to the kernel by value, so this lives in VGPRs
};

/**
 * The same model for ONE rollout on a whole wave (utils/nn_helpers/fnn_wave.hpp: lane j is neuron j of a layer, the
 * activations travel by v_readlane).  All 64 lanes carry the rollout's state (REPLICATED_LANES = 64); same fma chains as the
 * other forms, so the trajectories are the same bits.  Used for the re-rollout of the optimised control sequence, a chain of
 * T dependent steps: 0.5 us per step against 1.2 us on the MFMA form, whose 18 dependent MFMAs per step only pay for 16
 * rollouts at a time.
 */
template <int S_DIM, int C_DIM, int K_DIM>
class NeuralNetModelWave : public Dynamics<NeuralNetModelWave<S_DIM, C_DIM, K_DIM>, NNDynamicsParams>
{
public:
  using PARENT_CLASS = Dynamics<NeuralNetModelWave<S_DIM, C_DIM, K_DIM>, NNDynamicsParams>;
  static const int DYNAMICS_DIM = S_DIM - K_DIM;
  static constexpr int REPLICATED_LANES = 64;
  using NET = mppi::FNNWave<DYNAMICS_DIM + C_DIM, 32, 4>;

  NeuralNetModelWave(const NeuralNetModel<S_DIM, C_DIM, K_DIM>& other) : PARENT_CLASS(other.stream_)
  {
    this->params_ = other.params_;
    for (int i = 0; i < C_DIM; i++)
    {
      this->control_rngs_[i] = other.control_rngs_[i];
      this->control_deadband_[i] = other.control_deadband_[i];
      this->zero_control_[i] = other.zero_control_[i];
    }
    theta_d_ = other.helper_.theta_d_;
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
    net_.load(theta_d_, (int)(threadIdx.x & 63));
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
    float in[DYNAMICS_DIM + C_DIM], out[4];
#pragma unroll
    for (int i = 0; i < DYNAMICS_DIM; i++)
      in[i] = state[i + (S_DIM - DYNAMICS_DIM)];
#pragma unroll
    for (int i = 0; i < C_DIM; i++)
      in[DYNAMICS_DIM + i] = control[i];
    net_.forward(in, out);
#pragma unroll
    for (int i = 0; i < DYNAMICS_DIM; i++)
      state_der[i + (S_DIM - DYNAMICS_DIM)] = out[i];
  }

  const float* theta_d_ = nullptr;
  NET net_;
};

#endif
