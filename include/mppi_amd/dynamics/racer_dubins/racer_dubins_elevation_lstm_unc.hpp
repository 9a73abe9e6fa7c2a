/**
 * RacerDubinsElevationLSTMUncertainty plugin — the complete RACER vehicle model of the reference: the suspension model
 * (LSTM steering column, spring / damper body) with a quadratic brake lag, a "mean" LSTM that corrects the longitudinal
 * acceleration and the yaw rate, an "uncertainty" LSTM whose outputs are the process noise of the covariance
 * propagation, and the static settling of the plain elevation model kept alongside as two extra states.
 *
 * Reference: include/mppi/dynamics/racer_dubins/racer_dubins_elevation_lstm_unc.cuh:5-48 (parameters, the 26-entry state
 * layout), racer_dubins_elevation_lstm_unc.cu:496-605 (device step), :300-494 (computeQ), :607-618 (device
 * initializeDynamics).  Per step, in the reference's order:
 *   brake lag       clamp([e > 0] (c0+ e + c1+ e |e|) + [e < 0] (c0- e + c1- e |e|), -r_neg, +r_pos),  e = brake_cmd - brake
 *   acceleration, yaw and position rates (elevation model), LSTM steering, suspension forces (parents)
 *   forward gear:   (dv/dt, dyaw/dt) += mean LSTM([v, omega_z, brake, steer, steer rate, throttle cmd, brake cmd, steer cmd,
 *                   sin(static pitch), dv/dt, dyaw/dt])
 *   omega_z' = dyaw/dt;  Euler step (suspension model);  covariance with Q from the uncertainty LSTM:
 *                   o = |sigmoid(LSTM([..., sin(static roll), sin(static pitch), dv/dt, dyaw/dt])) * unc_scale|
 *                   Q_vv = o0 + (c_b [idx == 0 ? v : 1])^2 o4 (+ o5),  Q_yaw = o1 + (v / L / (cos^2(delta) k))^2 o3 (+ o6),
 *                   Q_xy block = o2 [sin^2, -sin cos; -sin cos, cos^2](yaw)
 *   static settling from (static roll, static pitch) at the next pose -> the next static roll / pitch
 * (In reverse gear the reference first calls the parent's computeQ and then overwrites every entry with the network's —
 * :306-309 has no return; only the overwrite is restated here.  Network inputs the step does not fill are zero, as the
 * reference's host path has them.)
 *
 * Networks: for the shapes of the reference's tests (tests/dynamics/racer_dubins_elevation_lstm_uncertainty_model_test.cu:
 * 26-48: steering LSTM(4, 4) + {8, 20, 1}, mean LSTM(12, 4) + {16, 20, 2}, uncertainty LSTM(13, 4) + {17, 20, 5}) all three
 * run on registers (utils/nn_helpers/lstm_registers.hpp: parameters through the scalar unit, activations in VGPRs; the
 * 3 x (4 + 4) recurrent values of a rollout rest in LDS between steps, laid out [value][slot]) or, four lanes per rollout,
 * on LSTMQuadRows (the class at the end of this file).  Any other shape ("lstm_structure", "mean_lstm_structure",
 * "unc_lstm_structure": {H, H + inputs, output-network layers ...}; mppi_load_npz sizes the networks from the archive as the
 * reference's constructor does) runs the general form: three LSTMHelper regions [parameters | slots x (h, c, activations)]
 * in LDS, the reference's contract, one lane per rollout.  Blobs: the steering pair of the parent, "mean_lstm_weights" /
 * "mean_lstm_output_weights", "unc_lstm_weights" / "unc_lstm_output_weights" (layouts of lstm_helper.hpp, initial hidden /
 * cell state in the tail), and "mean_lstm_state" / "unc_lstm_state" ([hidden | cell]) for the per-cycle update the
 * reference does in updateFromBuffer (:98-141; host: mppi::LSTMLSTMHelper).
 */
#ifndef MPPI_AMD_RACER_DUBINS_ELEVATION_LSTM_UNC_HPP_
#define MPPI_AMD_RACER_DUBINS_ELEVATION_LSTM_UNC_HPP_

#include "mppi_amd/dynamics/racer_dubins/racer_dubins_elevation_suspension.hpp"

/** reference: racer_dubins_elevation_lstm_unc.cuh:5-48 */
struct RacerDubinsElevationUncertaintyParams : public RacerDubinsElevationSuspensionParams
{
  enum class StateIndex : int
  {
    VEL_X = 0,
    YAW,
    POS_X,
    POS_Y,
    STEER_ANGLE,
    BRAKE_STATE,
    ROLL,
    PITCH,
    CG_POS_Z,
    CG_VEL_I_Z,
    ROLL_RATE,
    PITCH_RATE,
    STEER_ANGLE_RATE,
    OMEGA_Z,
    STATIC_ROLL,
    STATIC_PITCH,
    UNCERTAINTY_POS_X,
    UNCERTAINTY_POS_Y,
    UNCERTAINTY_YAW,
    UNCERTAINTY_VEL_X,
    UNCERTAINTY_POS_X_Y,
    UNCERTAINTY_POS_X_YAW,
    UNCERTAINTY_POS_X_VEL_X,
    UNCERTAINTY_POS_Y_YAW,
    UNCERTAINTY_POS_Y_VEL_X,
    UNCERTAINTY_YAW_VEL_X,
    NUM_STATES
  };
  float unc_scale[7] = { 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f };
  float pos_quad_brake_c[3] = { 2.0f, 0.5f, 0.3f };
  float neg_quad_brake_c[3] = { 5.84f, 0.15f, 1.7f };
  int use_static_settling = 1;  ///< carried for layout compatibility (the reference's bool); the device step always settles
};

/** racer_dubins_elevation_lstm_unc.cu:440-494: the process noise from the five outputs of the uncertainty network (shared by
 *  the one-lane and the four-lane form) */
template <class PARAMS_T, class TRIG>
__device__ __forceinline__ void networkOutputsToQ(const PARAMS_T& p, const float vx, const TRIG& g, const int index, float* o,
                                                  float* Q)
{
  constexpr int UD = 4;
  auto cm = [](int r, int c) { return r + UD * c; };
#pragma unroll
  for (int i = 0; i < 5; i++)
    o[i] = fabsf(mppi::det::sigmoid(o[i]) * p.unc_scale[i]);
#pragma unroll
  for (int i = 0; i < UD * UD; i++)
    Q[i] = 0.0f;
  const float b0 = p.c_b[0], b1 = p.c_b[1], b2 = p.c_b[2];  // values first, then selects (not a select of addresses)
  const float c_b = index == 0 ? b0 : (index == 1 ? b1 : b2);
  const float brake_gain = c_b * (index == 0 ? vx : 1.0f);
  Q[cm(RDE_U(VEL_X), RDE_U(VEL_X))] = o[0] + (brake_gain * brake_gain) * o[4];
  const float yaw_gain = (vx / p.wheel_base) * 1.0f / ((g.cos_delta * g.cos_delta) * p.steer_angle_scale);
  Q[cm(RDE_U(YAW), RDE_U(YAW))] = o[1] + (yaw_gain * yaw_gain) * o[3];
  Q[cm(RDE_U(POS_X), RDE_U(POS_X))] = o[2] * g.sin_yaw * g.sin_yaw;
  Q[cm(RDE_U(POS_X), RDE_U(POS_Y))] = -o[2] * g.sin_yaw * g.cos_yaw;
  Q[cm(RDE_U(POS_Y), RDE_U(POS_Y))] = o[2] * g.cos_yaw * g.cos_yaw;
  Q[cm(RDE_U(POS_Y), RDE_U(POS_X))] = -o[2] * g.sin_yaw * g.cos_yaw;
}

/** racer_dubins_elevation_lstm_unc.cu:527-533: the brake lag with linear + quadratic gains */
template <class PARAMS_T>
__device__ __forceinline__ float quadraticBrakeLag(const PARAMS_T& p, const float throttle_brake, const float brake_state)
{
  const bool enable_brake = throttle_brake < 0.0f;
  const float e = (enable_brake * -throttle_brake - brake_state);
  return fminf(fmaxf((e > 0) * (e * p.pos_quad_brake_c[0] + e * fabsf(e) * p.pos_quad_brake_c[1]) +
                         (e < 0) * (e * p.neg_quad_brake_c[0] + e * fabsf(e) * p.neg_quad_brake_c[1]),
                     -p.max_brake_rate_neg),
               p.max_brake_rate_pos);
}

class RacerDubinsElevationLSTMUncertainty
  : public RacerDubinsElevationSuspensionImpl<RacerDubinsElevationLSTMUncertainty, RacerDubinsElevationUncertaintyParams>
{
public:
  using PARAMS_T = RacerDubinsElevationUncertaintyParams;
  using SUSPENSION = RacerDubinsElevationSuspensionImpl<RacerDubinsElevationLSTMUncertainty, PARAMS_T>;
  using MEAN_NET = mppi::LSTMRegisters<12, 4, 20, 2>;
  using UNC_NET = mppi::LSTMRegisters<13, 4, 20, 5>;
  static constexpr int NUM_NETWORKS = 3, NET_H = 4;

  const float* mean_lstm_d_ = nullptr;  ///< mean network: LSTM blob / output-network blob (device, owned by the engine)
  const float* mean_fnn_d_ = nullptr;
  const float* unc_lstm_d_ = nullptr;   ///< uncertainty network
  const float* unc_fnn_d_ = nullptr;

  /** the two networks in the general form (LSTMHelper's LDS contract): used when any of the three networks has another
   *  shape than the register forms above are compiled for (`register_form_` false) */
  mppi::LSTMHelper mean_lstm_, unc_lstm_;
  static constexpr int MEAN_INPUT_DIM = 12, MEAN_OUTPUT_DIM = 2, UNC_INPUT_DIM = 13, UNC_OUTPUT_DIM = 5;

  RacerDubinsElevationLSTMUncertainty(hipStream_t stream = nullptr) : SUSPENSION(stream)
  {
    const int mean_layers[3] = { NET_H + MEAN_INPUT_DIM, 20, MEAN_OUTPUT_DIM };
    const int unc_layers[3] = { NET_H + UNC_INPUT_DIM, 20, UNC_OUTPUT_DIM };
    mean_lstm_.setStructure(MEAN_INPUT_DIM, NET_H, mean_layers, 3);
    unc_lstm_.setStructure(UNC_INPUT_DIM, NET_H, unc_layers, 3);
  }
  static const char* getDynamicsModelName()
  {
    return "RACER Dubins LSTM Uncertainty Model";
  }
  /** all three networks have the shapes the register forms are compiled for */
  bool defaultShapes() const
  {
    auto is = [](const mppi::LSTMHelper& n, int I, int OUT) {
      return n.HIDDEN_DIM == NET_H && n.INPUT_DIM == I && n.output_nn_.NUM_LAYERS == 3 && n.output_nn_.net_structure_[1] == 20 &&
             n.OUTPUT_DIM == OUT;
    };
    return is(this->lstm_, 4, 1) && is(mean_lstm_, MEAN_INPUT_DIM, MEAN_OUTPUT_DIM) && is(unc_lstm_, UNC_INPUT_DIM, UNC_OUTPUT_DIM);
  }
  /** host: the steering network's shape ("lstm_structure") */
  bool setLSTMStructure(const int* desc, int n)
  {
    const bool ok = SUSPENSION::setLSTMStructure(desc, n);
    this->register_form_ = defaultShapes();
    return ok;
  }
  /** host: the mean (which = 1) or uncertainty (2) network's shape, {H, H + inputs, ..., outputs} ("mean_lstm_structure",
   *  "unc_lstm_structure"); the input and output sizes are the model's (12 -> 2, 13 -> 5) */
  bool setNetworkStructure(const int which, const int* desc, const int n)
  {
    const int I = which == 1 ? MEAN_INPUT_DIM : UNC_INPUT_DIM, OUT = which == 1 ? MEAN_OUTPUT_DIM : UNC_OUTPUT_DIM;
    if ((which != 1 && which != 2) || n < 3 || desc[1] != desc[0] + I || desc[n - 1] != OUT)
      return false;
    if (!(which == 1 ? mean_lstm_ : unc_lstm_).setStructure(I, desc[0], desc + 1, n - 1))
      return false;
    this->register_form_ = defaultShapes();
    return true;
  }
  /** LDS: register form — the 3 x (4 + 4) recurrent values per slot; general form — three LSTMHelper regions
   *  [parameters | slots x (h, c, activations)] one after the other */
  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return this->register_form_ ? 0 :
                                  this->lstm_.getGrdSharedSizeBytes() + mean_lstm_.getGrdSharedSizeBytes() +
                                      unc_lstm_.getGrdSharedSizeBytes();
  }
  __host__ __device__ int getBlkSharedSizeBytes() const
  {
    return this->register_form_ ? NUM_NETWORKS * 2 * NET_H * (int)sizeof(float) :
                                  this->lstm_.getBlkSharedSizeBytes() + mean_lstm_.getBlkSharedSizeBytes() +
                                      unc_lstm_.getBlkSharedSizeBytes();
  }
  /** general form: start of the mean (1) / uncertainty (2) network's region */
  __device__ inline float* networkRegion(float* theta_s, const int which) const
  {
    const int slots = (int)(blockDim.x * blockDim.z);
    int off = (this->lstm_.getGrdSharedSizeBytes() + slots * this->lstm_.getBlkSharedSizeBytes()) / (int)sizeof(float);
    if (which == 2)
      off += (mean_lstm_.getGrdSharedSizeBytes() + slots * mean_lstm_.getBlkSharedSizeBytes()) / (int)sizeof(float);
    return theta_s + off;
  }
  /** general form: one forward pass of `net` living at `region` on the N inputs */
  template <int N>
  __device__ inline const float* generalForward(const mppi::LSTMHelper& net, float* region, const float (&input)[N]) const
  {
    float* input_loc = net.getInputLocation(region);
    if (__builtin_amdgcn_workitem_id_y() == 0)
    {
#pragma unroll
      for (int i = 0; i < N; i++)
        input_loc[i] = input[i];
    }
    mppi::lane_sync();
    return net.forward(nullptr, region);
  }

  /** racer_dubins_elevation_lstm_unc.cu:607-618: the parent's (steering state + outputs from the state), then the two
   *  other networks' initial hidden / cell state */
  __device__ __forceinline__ void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                            float dt)
  {
    SUSPENSION::initializeDynamics(state, control, output, theta_s, t_0, dt);
    if (this->register_form_)
    {
      float h[NET_H], c[NET_H];
      MEAN_NET::initialState(mean_lstm_d_, h, c);
      storeRecurrent(theta_s, h, c, 1);
      UNC_NET::initialState(unc_lstm_d_, h, c);
      storeRecurrent(theta_s, h, c, 2);
    }
    else
    {
      mean_lstm_.initialize(networkRegion(theta_s, 1));  // whole-block loads, each ends with a block barrier
      unc_lstm_.initialize(networkRegion(theta_s, 2));
    }
  }

  /** racer_dubins_elevation_lstm_unc.cu:300-494 (device branch) */
  __device__ __forceinline__ void computeNetworkQ(const float* state, const float* control, const float* state_der,
                                         const StepTrig& g, float* theta_s, float* Q) const
  {
    const PARAMS_T& p = this->S().params_;
    const float vx = state[RDE_S(VEL_X)];
    const float tb = control[RDE_C(THROTTLE_BRAKE)];
    const float input[13] = { vx,
                              state[RDE_S(OMEGA_Z)],
                              state[RDE_S(BRAKE_STATE)],
                              state[RDE_S(STEER_ANGLE)],
                              state[RDE_S(STEER_ANGLE_RATE)],
                              tb >= 0.0f ? tb : 0.0f,
                              tb <= 0.0f ? -tb : 0.0f,
                              control[RDE_C(STEER_CMD)],
                              mppi::det::sin(state[RDE_S(STATIC_ROLL)]),
                              mppi::det::sin(state[RDE_S(STATIC_PITCH)]),
                              state_der[RDE_S(VEL_X)],
                              state_der[RDE_S(YAW)],
                              0.0f };
    float o[5];
    if (this->register_form_)
    {
      float h[NET_H], c[NET_H];
      loadRecurrent(theta_s, h, c, 2);
      UNC_NET::forward(unc_lstm_d_, unc_fnn_d_, input, h, c, o);
      mppi::lane_sync();
      storeRecurrent(theta_s, h, c, 2);
    }
    else
    {
      const float* out = generalForward(unc_lstm_, networkRegion(theta_s, 2), input);
#pragma unroll
      for (int i = 0; i < 5; i++)
        o[i] = out[i];
    }
    networkOutputsToQ(p, vx, g, speedRegime(vx), o, Q);
  }

  /** racer_dubins_elevation_lstm_unc.cu:496-605 */
  __device__ __forceinline__ void step(float* state, float* next_state, float* state_der, float* control, float* output,
                              float* theta_s, const float t, const float dt)
  {
    const PARAMS_T& p = this->S().params_;
    float x[STATE_DIM], xn[STATE_DIM], xd[XD], u[CONTROL_DIM], wheel_out[OUTPUT_DIM];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      x[i] = state[i];
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      u[i] = control[i];
    const StepTrig g = stateTrig(x);
    xd[RDE_S(BRAKE_STATE)] = quadraticBrakeLag(p, u[RDE_C(THROTTLE_BRAKE)], x[RDE_S(BRAKE_STATE)]);
    computeParametricAccelDeriv(x, u, xd, g);
    computeLSTMSteering(x, u, xd, theta_s);
    computeSimpleSuspensionStep(x, xd, g, wheel_out);
    if (p.gear_sign == 1)
    {
      const float tb = u[RDE_C(THROTTLE_BRAKE)];
      const float input[12] = { x[RDE_S(VEL_X)],
                                x[RDE_S(OMEGA_Z)],
                                x[RDE_S(BRAKE_STATE)],
                                x[RDE_S(STEER_ANGLE)],
                                x[RDE_S(STEER_ANGLE_RATE)],
                                tb >= 0.0f ? tb : 0.0f,
                                tb <= 0.0f ? -tb : 0.0f,
                                u[RDE_C(STEER_CMD)],
                                mppi::det::sin(x[RDE_S(STATIC_PITCH)]),
                                xd[RDE_S(VEL_X)],
                                xd[RDE_S(YAW)],
                                0.0f };
      float mean_output[2];
      if (this->register_form_)
      {
        float h[NET_H], c[NET_H];
        loadRecurrent(theta_s, h, c, 1);
        MEAN_NET::forward(mean_lstm_d_, mean_fnn_d_, input, h, c, mean_output);
        mppi::lane_sync();
        storeRecurrent(theta_s, h, c, 1);
      }
      else
      {
        const float* out = generalForward(mean_lstm_, networkRegion(theta_s, 1), input);
        mean_output[0] = out[0];
        mean_output[1] = out[1];
      }
      xd[RDE_S(VEL_X)] += mean_output[0];
      xd[RDE_S(YAW)] += mean_output[1];
    }
    updateState(x, xn, xd, dt);
    xn[RDE_S(OMEGA_Z)] = xd[RDE_S(YAW)];
    computeUncertaintyPropagation(x, xd, xn, dt, g,
                                  [this, &x, &u, &xd, &g, theta_s](float* Q) { computeNetworkQ(x, u, xd, g, theta_s, Q); });
    // static settling from the settled angles of the current state (the body's own roll / pitch are suspension states)
    {
      StepTrig gs = g;
      mppi::det::sincos(angle_utils::normalizeAngle(x[RDE_S(STATIC_ROLL)]), &gs.sin_roll, &gs.cos_roll);
      mppi::det::sincos(angle_utils::normalizeAngle(x[RDE_S(STATIC_PITCH)]), &gs.sin_pitch, &gs.cos_pitch);
      float roll, pitch, height;
      computeStaticSettling(xn[RDE_S(YAW)], xn[RDE_S(POS_X)], xn[RDE_S(POS_Y)], gs, roll, pitch, height);
      xn[RDE_S(STATIC_PITCH)] = pitch;
      xn[RDE_S(STATIC_ROLL)] = roll;
    }
    mppi::lane_sync();
#pragma unroll
    for (int i = 0; i < XD; i++)
      state_der[i] = xd[i];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      next_state[i] = xn[i];
    output[RDE_O(WHEEL_FORCE_UP_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_UP_MAX)];
    output[RDE_O(WHEEL_FORCE_FWD_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_FWD_MAX)];
    output[RDE_O(WHEEL_FORCE_SIDE_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_SIDE_MAX)];
    setSuspensionOutputs(xd, xn, output);
  }
};

/**
 * Four lanes per rollout (racer_dubins_elevation_suspension.hpp: RacerDubinsElevationSuspensionQuadImpl has the angles, the
 * steering network, the wheels and the covariance rows).  Added here: the static angles' trigonometry in a third pass (the
 * two raw sines the networks read, the two wrapped sine / cosine pairs of the settling), the mean and the uncertainty
 * network on LSTMQuadRows (a hidden unit and five neurons of the output network per replica, the replica's weights kept once
 * per 16-lane row and fetched with DPP row broadcasts), the static settling as in the elevation model's four-lane form.
 * Same arithmetic per value as the one-lane class above.
 */
template <class CLASS_T>
class RacerDubinsElevationLSTMUncertaintyQuadImpl
  : public RacerDubinsElevationSuspensionQuadImpl<CLASS_T, RacerDubinsElevationUncertaintyParams>
{
public:
  using PARAMS_T = RacerDubinsElevationUncertaintyParams;
  using QUAD = RacerDubinsElevationSuspensionQuadImpl<CLASS_T, PARAMS_T>;
  using MEAN_NET = mppi::LSTMQuadRows<12, 20, 2>;
  using UNC_NET = mppi::LSTMQuadRows<13, 20, 5>;
  using StepTrig = typename QUAD::StepTrig;
  using MATH = typename QUAD::MATH;
  static constexpr int STATE_DIM = QUAD::STATE_DIM, CONTROL_DIM = QUAD::CONTROL_DIM, OUTPUT_DIM = QUAD::OUTPUT_DIM, XD = QUAD::XD;
  using QUAD::copyFrom;
  using QUAD::fromReplica;
  using QUAD::quadSteering;
  using QUAD::quadSuspension;
  using QUAD::quadTrig;
  using QUAD::replica;
  using QUAD::speedRegime;

  const float* mean_lstm_d_ = nullptr;
  const float* mean_fnn_d_ = nullptr;
  const float* unc_lstm_d_ = nullptr;
  const float* unc_fnn_d_ = nullptr;
  MEAN_NET mean_ = {};
  UNC_NET unc_ = {};

  RacerDubinsElevationLSTMUncertaintyQuadImpl(const RacerDubinsElevationLSTMUncertainty& other) : QUAD(other.stream_)
  {
    copyFrom(other);
    mean_lstm_d_ = other.mean_lstm_d_;
    mean_fnn_d_ = other.mean_fnn_d_;
    unc_lstm_d_ = other.unc_lstm_d_;
    unc_fnn_d_ = other.unc_fnn_d_;
  }

  __device__ __forceinline__ void initializeDynamics(float* state, float* control, float* output, float* theta_s, float t_0,
                                                     float dt)
  {
    QUAD::initializeDynamics(state, control, output, theta_s, t_0, dt);
    mean_.load(replica(), mean_lstm_d_, mean_fnn_d_);
    unc_.load(replica(), unc_lstm_d_, unc_fnn_d_);
  }

  /** racer_dubins_elevation_lstm_unc.cu:496-605 */
  __device__ __forceinline__ void step(float* state, float* next_state, float* state_der, float* control, float* output,
                                       float* theta_s, const float t, const float dt)
  {
    const PARAMS_T& p = this->S().params_;
    const int rep = replica();
    float x[STATE_DIM], xn[STATE_DIM], xd[XD], u[CONTROL_DIM], wheel_out[OUTPUT_DIM];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
    {
      x[i] = state[i];
      xn[i] = state[i];
    }
#pragma unroll
    for (int i = 0; i < CONTROL_DIM; i++)
      u[i] = control[i];
    StepTrig g;
    float sin_wheel_yaw, cos_wheel_yaw;
    quadTrig(x, rep, g, sin_wheel_yaw, cos_wheel_yaw);
    // ---- pass 3, the settled angles: replica 0 / 1 the raw static roll / pitch (network inputs), 2 / 3 the wrapped ones
    float sin_static_roll, sin_static_pitch;
    StepTrig gs = g;
    {
      const float raw = (rep & 1) ? x[RDE_S(STATIC_PITCH)] : x[RDE_S(STATIC_ROLL)];
      float s, c;
      mppi::det::sincos(rep < 2 ? raw : angle_utils::normalizeAngle(raw), &s, &c);
      sin_static_roll = fromReplica(s, 0);
      sin_static_pitch = fromReplica(s, 1);
      gs.sin_roll = fromReplica(s, 2);
      gs.cos_roll = fromReplica(c, 2);
      gs.sin_pitch = fromReplica(s, 3);
      gs.cos_pitch = fromReplica(c, 3);
    }
    xd[RDE_S(BRAKE_STATE)] = quadraticBrakeLag(p, u[RDE_C(THROTTLE_BRAKE)], x[RDE_S(BRAKE_STATE)]);
    this->computeParametricAccelDeriv(x, u, xd, g);
    quadSteering(x, u, xd);
    quadSuspension(x, g, rep, sin_wheel_yaw, cos_wheel_yaw, xd, wheel_out);
    const float tb = u[RDE_C(THROTTLE_BRAKE)];
    if (p.gear_sign == 1)
    {
      const float input[12] = { x[RDE_S(VEL_X)],
                                x[RDE_S(OMEGA_Z)],
                                x[RDE_S(BRAKE_STATE)],
                                x[RDE_S(STEER_ANGLE)],
                                x[RDE_S(STEER_ANGLE_RATE)],
                                tb >= 0.0f ? tb : 0.0f,
                                tb <= 0.0f ? -tb : 0.0f,
                                u[RDE_C(STEER_CMD)],
                                sin_static_pitch,
                                xd[RDE_S(VEL_X)],
                                xd[RDE_S(YAW)],
                                0.0f };
      float mean_output[2];
      mean_.forward(this->S().mean_fnn_d_, input, mean_output);
      xd[RDE_S(VEL_X)] += mean_output[0];
      xd[RDE_S(YAW)] += mean_output[1];
    }
    this->updateState(x, xn, xd, dt);
    xn[RDE_S(STEER_ANGLE_RATE)] = x[RDE_S(STEER_ANGLE_RATE)] + xd[RDE_S(STEER_ANGLE_RATE)] * dt;
    xn[RDE_S(OMEGA_Z)] = xd[RDE_S(YAW)];
    // ---- covariance, the process noise from the uncertainty network (:300-494)
    this->covarianceFourLanes(x, xd, g, dt, rep, xn, [&](float* Q) {
      const float input[13] = { x[RDE_S(VEL_X)],
                                x[RDE_S(OMEGA_Z)],
                                x[RDE_S(BRAKE_STATE)],
                                x[RDE_S(STEER_ANGLE)],
                                x[RDE_S(STEER_ANGLE_RATE)],
                                tb >= 0.0f ? tb : 0.0f,
                                tb <= 0.0f ? -tb : 0.0f,
                                u[RDE_C(STEER_CMD)],
                                sin_static_roll,
                                sin_static_pitch,
                                xd[RDE_S(VEL_X)],
                                xd[RDE_S(YAW)],
                                0.0f };
      float o[5];
      unc_.forward(this->S().unc_fnn_d_, input, o);
      networkOutputsToQ(p, x[RDE_S(VEL_X)], g, speedRegime(x[RDE_S(VEL_X)]), o, Q);
    });
    // ---- static settling at the next pose from the settled angles of the current state
    {
      float sin_psi, cos_psi, roll, pitch, height;
      mppi::det::sincos(angle_utils::normalizeAngle(xn[RDE_S(YAW)]), &sin_psi, &cos_psi);
      this->settleFourLanes(gs, sin_psi, cos_psi, xn[RDE_S(POS_X)], xn[RDE_S(POS_Y)], rep, roll, pitch, height);
      xn[RDE_S(STATIC_PITCH)] = pitch;
      xn[RDE_S(STATIC_ROLL)] = roll;
    }
#pragma unroll
    for (int i = 0; i < XD; i++)
      state_der[i] = xd[i];
#pragma unroll
    for (int i = 0; i < STATE_DIM; i++)
      next_state[i] = xn[i];
    output[RDE_O(WHEEL_FORCE_UP_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_UP_MAX)];
    output[RDE_O(WHEEL_FORCE_FWD_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_FWD_MAX)];
    output[RDE_O(WHEEL_FORCE_SIDE_MAX)] = wheel_out[RDE_O(WHEEL_FORCE_SIDE_MAX)];
    MATH::setSuspensionOutputs(p, xd, xn, output);
  }
};

class RacerDubinsElevationLSTMUncertaintyQuadRobust;
/** the four-lane form the rollout / re-rollout / init-eval kernels run */
class RacerDubinsElevationLSTMUncertaintyQuad
  : public RacerDubinsElevationLSTMUncertaintyQuadImpl<RacerDubinsElevationLSTMUncertaintyQuad>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  /** S() of RacerDubinsElevationImpl: 0 = the object itself.  1 (argument block, s_load) and 2 (copy in LDS, ds_read) remove
   *  most of the spilled-SGPR reads of the step loop and are SLOWER here (profiles/r06_step_source_ab.json) — A/B:
   *  -DMPPI_STEP_SOURCE_QUAD=1|2 */
  static constexpr int MPPI_STEP_SOURCE = MPPI_STEP_SOURCE_QUAD;
  /** the form the role-pipelined ROBUST rollout kernel runs instead (engine/model_instance.hpp: withRmppiPipelineDynamics) */
  using RMPPI_PIPELINE_FORM = RacerDubinsElevationLSTMUncertaintyQuadRobust;
  RacerDubinsElevationLSTMUncertaintyQuad(const RacerDubinsElevationLSTMUncertainty& other)
    : RacerDubinsElevationLSTMUncertaintyQuadImpl<RacerDubinsElevationLSTMUncertaintyQuad>(other)
  {
  }
};

/**
 * The same model for rolloutRMPPIPipelineKernel — the one kernel where registers, not the dependent chain, are what is short:
 * two systems x four dynamics waves + helpers = 15 waves per block leave every wave 128 VGPRs, and this model's dynamics waves
 * then spill 200+ of them to scratch memory (round 5: 3399 us per launch at K = 16384, T = 100 — 4.2 x the one-system kernel).
 * Three choices that LOSE everywhere else win here (profiles/r06_robust_racer_ab.json):
 *   - one sampler wave and ONE cost wave per system: 11 waves = 3 per SIMD = 168 VGPRs (spilled 210 -> 99; 3072 -> 2047 us);
 *   - the read-only members through the kernel's argument block (S() source 1) and the cost class off the argument block instead
 *     of a VGPR-pinned copy: the parameters leave the register file altogether (spilled 99 -> 63-68; 2047 -> 1598 us).
 * Same arithmetic, same bits (tests/test_full_size_parity.py::test_robust_complete_racer_4096x100_vs_oracle, tests/test_rmppi.py).
 */
class RacerDubinsElevationLSTMUncertaintyQuadRobust
  : public RacerDubinsElevationLSTMUncertaintyQuadImpl<RacerDubinsElevationLSTMUncertaintyQuadRobust>
{
public:
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  static constexpr int MPPI_STEP_SOURCE = MPPI_KERNARG_RELOAD ? 1 : 0;
  static constexpr bool MPPI_RMPPI_COST_VIEW = true;
  static constexpr int MPPI_RMPPI_PIPE_SAMPLERS = 1;
  static constexpr int MPPI_RMPPI_PIPE_COSTS = 1;
  RacerDubinsElevationLSTMUncertaintyQuadRobust(const RacerDubinsElevationLSTMUncertainty& other)
    : RacerDubinsElevationLSTMUncertaintyQuadImpl<RacerDubinsElevationLSTMUncertaintyQuadRobust>(other)
  {
  }
};

#endif
