/**
 * RacerDubinsElevationLSTMSteering plugin — the elevation-map RACER Dubins car whose steering column is a second-order
 * parametric model plus an LSTM correction evaluated at every step of every rollout.
 *
 * Reference: include/mppi/dynamics/racer_dubins/racer_dubins_elevation_lstm_steering.cuh:18-117,
 * racer_dubins_elevation_lstm_steering.cu:131-167 (device computeLSTMSteering), :169-213 (device step), :115-129 (device
 * initializeDynamics), :243-268 (device updateState); everything else is RacerDubinsElevationImpl
 * (racer_dubins_elevation.hpp).  What changes against the plain elevation model:
 *   steering rate   d(rate)/dt = clamp((k (u1 k_cmd - steer) - rate) k_acc - rate k_drag, +-max_steer_rate) + 5 LSTM(.)[0]
 *   steering angle  d(steer)/dt = rate;  the rate is an integrated state (rate' = rate + d(rate)/dt dt)
 *   network input   [0.2 steer, 0.2 rate, u1, 0.2 parametric d(rate)/dt]   (I = 4)
 *   t = 0 outputs   setOutputs(state, state) — not the base class's state copy
 * Network: LSTMHelper(I = 4, H, output MLP {H + 4, ..., 1}); the shape the reference's tests use is H = 4, {8, 20, 1}
 * (tests/dynamics/racer_dubins_elevation_lstm_steering_model_test.cu:26-32) and is the default here; another hidden size
 * or output network comes with the "lstm_structure" blob before the weights.  The INITIAL hidden / cell state of every
 * rollout is the tail of the "lstm_weights" blob; producing it from the recent steering history is host work in the
 * reference too (LSTMLSTMHelper::initializeLSTM on the state-estimator thread, utils/nn_helpers/lstm_lstm_helper.cu:50-76)
 * — here mppi::LSTMLSTMHelper (utils/nn_helpers/lstm_lstm_helper.hpp, mppi_lstm_lstm_initialize in the C ABI).
 *
 * On the device the default shape runs on registers (utils/nn_helpers/lstm_registers.hpp): the 345 parameters come through
 * the scalar unit, the activations are VGPRs, and between two steps the 2 x 4 recurrent values of a rollout rest in LDS,
 * laid out [value][slot] so that the lanes of a wave hit 64 different banks (carrying them in members of the lane's copy
 * of this object would move the whole object — parameters, map descriptor — into scratch memory).  Any other shape falls back to the reference's LDS contract
 * (utils/nn_helpers/lstm_helper.hpp: parameters in LDS once per block, [h | c | activations] per rollout slot) — same
 * arithmetic, same bits, several times slower for a network this small.  The rest of the step runs on registers as in
 * the parent.
 */
#ifndef MPPI_AMD_RACER_DUBINS_ELEVATION_LSTM_STEERING_HPP_
#define MPPI_AMD_RACER_DUBINS_ELEVATION_LSTM_STEERING_HPP_

#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation.hpp"
#include "mppi_amd/utils/nn_helpers/lstm_helper.hpp"
#include "mppi_amd/utils/nn_helpers/lstm_registers.hpp"
#include "mppi_amd/utils/nn_helpers/lstm_quad.hpp"

/** reference: RacerDubinsElevationLSTMSteeringImpl<CLASS_T, PARAMS_T> (racer_dubins_elevation_lstm_steering.cuh:18-117);
 *  CLASS_T is the instantiated model — RacerDubinsElevationLSTMSteering below, the suspension models in their own header */
template <class CLASS_T, class PARAMS_T = RacerDubinsElevationParams>
class RacerDubinsElevationLSTMSteeringImpl : public RacerDubinsElevationImpl<CLASS_T, PARAMS_T>
{
public:
  using ELEVATION = RacerDubinsElevationImpl<CLASS_T, PARAMS_T>;
  using StepTrig = typename ELEVATION::StepTrig;
  static constexpr int STATE_DIM = ELEVATION::STATE_DIM, CONTROL_DIM = ELEVATION::CONTROL_DIM,
                       OUTPUT_DIM = ELEVATION::OUTPUT_DIM;
  using ELEVATION::computeParametricAccelDeriv;
  using ELEVATION::computeParametricDelayDeriv;
  using ELEVATION::computeStaticSettling;
  using ELEVATION::computeUncertaintyPropagation;
  using ELEVATION::setOutputs;
  using ELEVATION::stateTrig;
  static constexpr int LSTM_INPUT_DIM = 4;

  static constexpr int FAST_H = 4, FAST_L1 = 20;
  using FAST_NET = mppi::LSTMRegisters<LSTM_INPUT_DIM, FAST_H, FAST_L1, 1>;

  /** the prediction LSTM (reference: lstm_lstm_helper_->getLSTMModel(), network_d_ on the device) */
  mppi::LSTMHelper lstm_;
  /** true while the network has the shape FAST_NET is compiled for */
  bool register_form_ = true;

  RacerDubinsElevationLSTMSteeringImpl(hipStream_t stream = nullptr) : ELEVATION(stream)
  {
    const int out_layers[3] = { 4 + LSTM_INPUT_DIM, 20, 1 };
    lstm_.setStructure(LSTM_INPUT_DIM, 4, out_layers, 3);
  }
  static const char* getDynamicsModelName()
  {
    return "RACER Dubins LSTM Steering Model";
  }
  /** host: another hidden size / output network, {H, layer sizes of the output MLP ...}; the LSTM input stays 4 */
  bool setLSTMStructure(const int* desc, int n)
  {
    if (n < 3 || desc[1] != desc[0] + LSTM_INPUT_DIM || desc[n - 1] < 1)
      return false;
    if (!lstm_.setStructure(LSTM_INPUT_DIM, desc[0], desc + 1, n - 1))
      return false;
    register_form_ = (n == 4 && desc[0] == FAST_H && desc[2] == FAST_L1 && desc[3] == 1);
    return true;
  }
  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return register_form_ ? 0 : lstm_.getGrdSharedSizeBytes();
  }
  __host__ __device__ int getBlkSharedSizeBytes() const
  {
    return register_form_ ? 2 * FAST_H * (int)sizeof(float) : lstm_.getBlkSharedSizeBytes();
  }
  /** register form: value i of this lane's rollout slot at theta_s[i * slots + slot] */
  /** `net`: which of the rollout's networks (0 = steering; the uncertainty model keeps two more sets behind it) */
  __device__ static inline void loadRecurrent(const float* theta_s, float (&h)[FAST_H], float (&c)[FAST_H], const int net = 0)
  {
    const int slots = (int)(blockDim.x * blockDim.z), slot = (int)(blockDim.x * threadIdx.z + threadIdx.x);
#pragma unroll
    for (int i = 0; i < FAST_H; i++)
    {
      h[i] = theta_s[(2 * FAST_H * net + i) * slots + slot];
      c[i] = theta_s[(2 * FAST_H * net + FAST_H + i) * slots + slot];
    }
  }
  __device__ static inline void storeRecurrent(float* theta_s, const float (&h)[FAST_H], const float (&c)[FAST_H],
                                               const int net = 0)
  {
    const int slots = (int)(blockDim.x * blockDim.z), slot = (int)(blockDim.x * threadIdx.z + threadIdx.x);
#pragma unroll
    for (int i = 0; i < FAST_H; i++)
    {
      theta_s[(2 * FAST_H * net + i) * slots + slot] = h[i];
      theta_s[(2 * FAST_H * net + FAST_H + i) * slots + slot] = c[i];
    }
  }

  /** racer_dubins_elevation_lstm_steering.cu:115-129: network parameters and (h0, c0) into LDS, outputs from the state */
  __device__ __forceinline__ void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    if (register_form_)
    {
      float h[FAST_H], c[FAST_H];
      FAST_NET::initialState(lstm_.weights_d_, h, c);
      storeRecurrent(theta_s, h, c);  // this lane's own slot: nothing to synchronise with
    }
    else
      lstm_.initialize(theta_s);
    int first, stride;
    mppi::p1::getParallel1DIndex<mppi::p1::Parallel1Dir::THREAD_Y>(first, stride);
    if (first == 0)
    {
      output[RDE_O(BASELINK_POS_I_Z)] = 0.0f;  // not written by setOutputs: defined instead of the buffer's content
      output[RDE_O(FILLER_1)] = 0.0f;
      setOutputs(state, state, output);
    }
    mppi::lane_sync();
  }

  /** racer_dubins_elevation_lstm_steering.cu:131-167 */
  __device__ __forceinline__ void computeLSTMSteering(const float* state, const float* control, float* state_der,
                                             float* theta_s) const
  {
    const PARAMS_T& p = this->S().params_;
    const float steer = state[RDE_S(STEER_ANGLE)], rate = state[RDE_S(STEER_ANGLE_RATE)];
    const float parametric_accel = (control[RDE_C(STEER_CMD)] * p.steer_command_angle_scale - steer) * p.steering_constant;
    float rate_dot = fmaxf(fminf((parametric_accel - rate) * p.steer_accel_constant - rate * p.steer_accel_drag_constant,
                                 p.max_steer_rate),
                           -p.max_steer_rate);
    // network input; the parametric part is its fourth entry
    const float input[LSTM_INPUT_DIM] = { steer * 0.2f, rate * 0.2f, control[RDE_C(STEER_CMD)], rate_dot * 0.2f };
    if (register_form_)
    {
      float nn_output[1], h[FAST_H], c[FAST_H];
      loadRecurrent(theta_s, h, c);
      FAST_NET::forward(lstm_.weights_d_, lstm_.output_nn_.theta_d_, input, h, c, nn_output);
      mppi::lane_sync();  // BY > 1: sibling lanes have read the slot before anyone rewrites it (with the same values)
      storeRecurrent(theta_s, h, c);
      rate_dot += nn_output[0] * 5.0f;
    }
    else
    {
      float* input_loc = lstm_.getInputLocation(theta_s);
      if (__builtin_amdgcn_workitem_id_y() == 0)
      {
#pragma unroll
        for (int i = 0; i < LSTM_INPUT_DIM; i++)
          input_loc[i] = input[i];
      }
      mppi::lane_sync();
      const float* nn_output = lstm_.forward(nullptr, theta_s);
      rate_dot += nn_output[0] * 5.0f;
    }
    state_der[RDE_S(STEER_ANGLE_RATE)] = rate_dot;
    state_der[RDE_S(STEER_ANGLE)] = rate;
  }

  /** racer_dubins_elevation_lstm_steering.cu:243-268: as the parent, but the steering rate is integrated */
  __device__ inline void updateState(const float* state, float* next_state, const float* state_der, const float dt) const
  {
    ELEVATION::updateState(state, next_state, state_der, dt);
    next_state[RDE_S(STEER_ANGLE_RATE)] = state[RDE_S(STEER_ANGLE_RATE)] + state_der[RDE_S(STEER_ANGLE_RATE)] * dt;
  }

  /** racer_dubins_elevation_lstm_steering.cu:169-213 */
  __device__ __forceinline__ void step(float* state, float* next_state, float* state_der, float* control, float* output,
                              float* theta_s, const float t, const float dt)
  {
    float x[STATE_DIM], xn[STATE_DIM], xd[RDE_S(STEER_ANGLE_RATE) + 1], u[CONTROL_DIM];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      x[i] = state[i];
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      u[i] = control[i];
    const StepTrig g = stateTrig(x);
    computeParametricDelayDeriv(x, u, xd);
    computeParametricAccelDeriv(x, u, xd, g);
    computeLSTMSteering(x, u, xd, theta_s);
    updateState(x, xn, xd, dt);
    computeUncertaintyPropagation(x, xd, xn, dt, g);
    float roll, pitch, height;
    computeStaticSettling(xn[RDE_S(YAW)], xn[RDE_S(POS_X)], xn[RDE_S(POS_Y)], g, roll, pitch, height);
    xn[RDE_S(PITCH)] = pitch;
    xn[RDE_S(ROLL)] = roll;
    mppi::lane_sync();
#pragma unroll
    for (int i = 0; i < 6; i++)
      state_der[i] = xd[i];
    state_der[RDE_S(STEER_ANGLE_RATE)] = xd[RDE_S(STEER_ANGLE_RATE)];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      next_state[i] = xn[i];
    output[RDE_O(BASELINK_POS_I_Z)] = height;
    setOutputs(xd, xn, output);
  }
};

class RacerDubinsElevationLSTMSteering : public RacerDubinsElevationLSTMSteeringImpl<RacerDubinsElevationLSTMSteering>
{
public:
  using PARAMS_T = RacerDubinsElevationParams;
  RacerDubinsElevationLSTMSteering(hipStream_t stream = nullptr)
    : RacerDubinsElevationLSTMSteeringImpl<RacerDubinsElevationLSTMSteering>(stream)
  {
  }
};

/**
 * Four lanes per rollout (see RacerDubinsElevationQuad): on top of the shared-out trigonometry, wheels and covariance
 * rows, replica r of a rollout evaluates hidden unit r of the LSTM (its four gates: 32 multiply-adds, 5 transcendentals)
 * and five of the twenty neurons of the output network's hidden layer; the new hidden state (4 values) and the layer's
 * activations (20 values) are exchanged with `__shfl`, the last layer's 20-term sum runs on every replica in the order
 * of the one-lane form.  The weights a replica needs differ from lane to lane, so they cannot be scalar operands: each
 * lane loads its 81 once in initializeDynamics() and keeps them in VGPRs for the whole rollout (members of the lane's
 * by-value copy of this object, like the recurrent state: h of all four units, c of its own).  Default network shape
 * only (H = 4, MLP {8, 20, 1}).  Same arithmetic per value as LSTMRegisters / LSTMHelper / the oracle.
 */
class RacerDubinsElevationLSTMSteeringQuad : public RacerDubinsElevationImpl<RacerDubinsElevationLSTMSteeringQuad>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  /** S() of RacerDubinsElevationImpl: 0 = the object itself.  1 (argument block, s_load) and 2 (copy in LDS, ds_read) remove
   *  most of the spilled-SGPR reads of the step loop and are SLOWER (profiles/r06_step_source_ab.json) — A/B: -DMPPI_STEP_SOURCE_QUAD=1|2 */
  static constexpr int MPPI_STEP_SOURCE = MPPI_STEP_SOURCE_QUAD;
  /** helper waves of the role-pipelined Robust kernel for this model (engine/rmppi_pipeline_kernel.hpp): one sampler and ONE cost
   *  wave per system = 11 waves per block = 3 per SIMD = 168 VGPRs per lane instead of 15 waves / 128 (K = 16384, T = 100, us per
   *  launch: elevation 356 -> 352, LSTM steering 601 -> 489, suspension 1013 -> 615; two cost waves per system = 13 waves change
   *  nothing — profiles/r06_robust_racer_ab.json) */
  static constexpr int MPPI_RMPPI_PIPE_SAMPLERS = 1;
  static constexpr int MPPI_RMPPI_PIPE_COSTS = 1;
  using ELEVATION = RacerDubinsElevationImpl<RacerDubinsElevationLSTMSteeringQuad>;
  using PARAMS_T = RacerDubinsElevationParams;
  static constexpr int REPLICATED_LANES = 4;
  using NET = mppi::LSTMQuadRows<4, 20, 1>;

  const float* lstm_d_ = nullptr;
  const float* fnn_d_ = nullptr;
  NET net_ = {};  ///< this lane's weights and the recurrent state

  RacerDubinsElevationLSTMSteeringQuad(const RacerDubinsElevationLSTMSteering& other) : ELEVATION(other.stream_)
  {
    this->params_ = other.params_;
    for (int i = 0; i < CONTROL_DIM; i++)
    {
      this->control_rngs_[i] = other.control_rngs_[i];
      this->control_deadband_[i] = other.control_deadband_[i];
      this->zero_control_[i] = other.zero_control_[i];
    }
    this->tex_helper_ = other.tex_helper_;
    lstm_d_ = other.lstm_.weights_d_;
    fnn_d_ = other.lstm_.output_nn_.theta_d_;
  }

  __device__ __forceinline__ void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    this->stageStepSource(theta_s);  // the read-only members' copy in LDS first: setOutputs below reads through S()
    net_.load((int)(threadIdx.x & 63) >> 4, lstm_d_, fnn_d_);
    output[RDE_O(BASELINK_POS_I_Z)] = 0.0f;
    output[RDE_O(FILLER_1)] = 0.0f;
    setOutputs(state, state, output);
  }

  /** racer_dubins_elevation_lstm_steering.cu:131-167, the network shared out over the four replicas */
  __device__ __forceinline__ void computeLSTMSteering(const float* state, const float* control, float* state_der)
  {
    const PARAMS_T& p = this->S().params_;
    const float steer = state[RDE_S(STEER_ANGLE)], rate = state[RDE_S(STEER_ANGLE_RATE)];
    const float parametric_accel = (control[RDE_C(STEER_CMD)] * p.steer_command_angle_scale - steer) * p.steering_constant;
    float rate_dot = fmaxf(fminf((parametric_accel - rate) * p.steer_accel_constant - rate * p.steer_accel_drag_constant,
                                 p.max_steer_rate),
                           -p.max_steer_rate);
    const float input[4] = { steer * 0.2f, rate * 0.2f, control[RDE_C(STEER_CMD)], rate_dot * 0.2f };
    float out[1] = { 0.0f };
    net_.forward(this->S().fnn_d_, input, out);
    rate_dot += out[0] * 5.0f;
    state_der[RDE_S(STEER_ANGLE_RATE)] = rate_dot;
    state_der[RDE_S(STEER_ANGLE)] = rate;
  }

  __device__ __forceinline__ void step(float* state, float* next_state, float* state_der, float* control, float* output,
                              float* theta_s, const float t, const float dt)
  {
    stepFourLanes<RDE_S(STEER_ANGLE_RATE) + 1>(
        state, next_state, state_der, control, output, dt,
        [this](const float* x, const float* u, float* xd) { computeLSTMSteering(x, u, xd); },
        [dt](const float* x, const float* xd, float* xn) {
          xn[RDE_S(STEER_ANGLE_RATE)] = x[RDE_S(STEER_ANGLE_RATE)] + xd[RDE_S(STEER_ANGLE_RATE)] * dt;
        });
  }
};

#endif
