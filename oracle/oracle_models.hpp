/**
 * oracle_models.hpp — CPU restatement of the concrete Dynamics / Cost plugins on the hot path.  TEST INFRASTRUCTURE ONLY.
 * Device flavour of every formula (see oracle_core.hpp); paths relative to the reference's include/mppi/.
 */
#ifndef MPPI_ORACLE_MODELS_HPP_
#define MPPI_ORACLE_MODELS_HPP_

#include "oracle_core.hpp"
#include "oracle_texture.hpp"
#include "mppi_amd/model_params.h"

#define ORACLE_SQ(a) ((a) * (a)) /* reference: utils/math_utils.h SQ() */

namespace oracle
{
/* ------------------------------------------------------------------ Cartpole -------------------------------------- */
/** reference: dynamics/cartpole/cartpole_dynamics.cu:89-107 (device computeDynamics), cartpole_dynamics.cuh:101 gravity */
struct CartpoleDynamics : Dynamics
{
  mppi_cartpole_dynamics_params p{ 1.0f, 1.0f, 1.0f };
  const float gravity_ = 9.81;
  CartpoleDynamics() : Dynamics(4, 1, 4)
  {
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(p))
      return -1;
    memcpy(&p, pod, n);
    return 0;
  }
  void computeDynamics(const float* state, const float* control, float* state_der, float* theta_s) override
  {
    float theta = det::normalizeAngle(state[2]);
    float sin_theta, cos_theta;
    det::sincos(theta, &sin_theta, &cos_theta); /* reference: __sinf/__cosf */
    float theta_dot = state[3];
    float force = control[0];
    float m_c = p.cart_mass;
    float m_p = p.pole_mass;
    float l_p = p.pole_length;

    state_der[0] = state[1];
    state_der[1] = 1.0f / (m_c + m_p * ORACLE_SQ(sin_theta)) *
                   (force + m_p * sin_theta * (l_p * ORACLE_SQ(theta_dot) + gravity_ * cos_theta));
    state_der[2] = theta_dot;
    state_der[3] = 1.0f / (l_p * (m_c + m_p * ORACLE_SQ(sin_theta))) *
                   (-force * cos_theta - m_p * l_p * ORACLE_SQ(theta_dot) * cos_theta * sin_theta -
                    (m_c + m_p) * gravity_ * sin_theta);
  }
};

/** reference: cost_functions/cartpole/cartpole_quadratic_cost.cu:20-43 */
struct CartpoleQuadraticCost : Cost
{
  mppi_cartpole_cost_params params_{ { 10.0f }, 1.0f, 1000.0f, 100.0f, 2000.0f, 100.0f, 0.0f, { 0.0f, 0.0f, (float)M_PI, 0.0f } };
  CartpoleQuadraticCost() : Cost(1, 4)
  {
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(params_))
      return -1;
    memcpy(&params_, pod, n);
    return 0;
  }
  float quad(const float* state) const
  {
    return (state[0] - params_.desired_terminal_state[0]) * (state[0] - params_.desired_terminal_state[0]) *
               params_.cart_position_coeff +
           (state[1] - params_.desired_terminal_state[1]) * (state[1] - params_.desired_terminal_state[1]) *
               params_.cart_velocity_coeff +
           (state[2] - params_.desired_terminal_state[2]) * (state[2] - params_.desired_terminal_state[2]) *
               params_.pole_angle_coeff +
           (state[3] - params_.desired_terminal_state[3]) * (state[3] - params_.desired_terminal_state[3]) *
               params_.pole_angular_velocity_coeff;
  }
  float computeStateCost(const float* s, int t, int* crash) override
  {
    return quad(s);
  }
  float terminalCost(const float* s) override
  {
    return quad(s) * params_.terminal_cost_coeff;
  }
};

/* ------------------------------------------------------------------ Double integrator ------------------------------ */
/** reference: dynamics/double_integrator/di_dynamics.cu:46-53 */
struct DoubleIntegratorDynamics : Dynamics
{
  mppi_di_dynamics_params p{ 1.0f };
  DoubleIntegratorDynamics() : Dynamics(4, 2, 4)
  {
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(p))
      return -1;
    memcpy(&p, pod, n);
    return 0;
  }
  void computeDynamics(const float* state, const float* control, float* state_der, float* theta_s) override
  {
    state_der[0] = state[2];
    state_der[1] = state[3];
    state_der[2] = control[0];
    state_der[3] = control[1];
  }
};

/** reference: cost_functions/double_integrator/double_integrator_circle_cost.cu:8-32 */
struct DoubleIntegratorCircleCost : Cost
{
  mppi_di_circle_cost_params params_{ { 0.01f, 0.01f }, 1.0f, 1.0f, 1000.0f, 2.0f, 1.875f * 1.875f, 2.125f * 2.125f, 4.0f };
  DoubleIntegratorCircleCost() : Cost(2, 4)
  {
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(params_))
      return -1;
    memcpy(&params_, pod, n);
    return 0;
  }
  /** powf(discount, timestep): exact 1 for the default discount == 1; otherwise det::pow_pos (shared with the engine) */
  static float discountPow(float discount, int timestep)
  {
    return discount == 1.0f ? 1.0f : det::pow_pos(discount, (float)timestep);
  }
  float computeStateCost(const float* s, int timestep, int* crash) override
  {
    float radial_position = s[0] * s[0] + s[1] * s[1];
    float current_velocity = det::sqrt(s[2] * s[2] + s[3] * s[3]);
    float current_angular_momentum = s[0] * s[3] - s[1] * s[2];
    float cost = 0;
    if ((radial_position < params_.inner_path_radius2) || (radial_position > params_.outer_path_radius2))
    {
      cost += discountPow(params_.discount, timestep) * params_.crash_cost;
    }
    cost += params_.velocity_cost * fabsf(current_velocity - params_.velocity_desired);
    cost += params_.velocity_cost * fabsf(current_angular_momentum - params_.angular_momentum_desired);
    return cost;
  }
  float terminalCost(const float* s) override
  {
    return 0;
  }
};

/**
 * reference: cost_functions/double_integrator/double_integrator_robust_cost.cu:10-41 — the DEVICE overload (knee at half
 * the track width with half the crash cost; the host overload's 0.75 / 0.1 constants, :55-56, are never what a rollout
 * kernel evaluates), utils/math_utils.h:90-94 (linInterp), :149-156 (normDistFromCenter).  powf(x, 2) == x * x.
 */
struct DoubleIntegratorRobustCost : DoubleIntegratorCircleCost
{
  float computeStateCost(const float* s, int timestep, int* crash) override
  {
    float radial_position = s[0] * s[0] + s[1] * s[1];
    float current_velocity = det::sqrt(s[2] * s[2] + s[3] * s[3]);
    float current_angular_momentum = s[0] * s[3] - s[1] * s[2];
    float r = det::sqrt(radial_position), r_in = det::sqrt(params_.inner_path_radius2),
          r_out = det::sqrt(params_.outer_path_radius2);
    float r_center = (r_in + r_out) / 2.0f;
    float r_width = (r_out - r_in);
    float normalized_dist_from_center = fabsf(r - r_center) / (r_width * 0.5f);
    float steep_percent_boundary = 0.5f;
    float steep_cost = 0.5f * params_.crash_cost;
    float cost = 0;
    if (normalized_dist_from_center <= steep_percent_boundary)
    {
      cost += lin(normalized_dist_from_center, 0, steep_percent_boundary, 0, steep_cost);
    }
    if (normalized_dist_from_center > steep_percent_boundary && normalized_dist_from_center <= 1.0f)
    {
      cost += lin(normalized_dist_from_center, steep_percent_boundary, 1, steep_cost, params_.crash_cost);
    }
    if (normalized_dist_from_center > 1.0f)
    {
      cost += params_.crash_cost;
    }
    float dv = current_velocity - params_.velocity_desired;
    float dl = current_angular_momentum - params_.angular_momentum_desired;
    cost += params_.velocity_cost * (dv * dv);
    cost += params_.velocity_cost * (dl * dl);
    return cost;
  }
  static float lin(float x, float x_min, float x_max, float y_min, float y_max)
  {
    return (x - x_min) / (x_max - x_min) * (y_max - y_min) + y_min;
  }
};

/* ------------------------------------------------------------------ AutoRally NN ---------------------------------- */
/**
 * Fully connected network, device flavour of FNNHelper::forward (utils/nn_helpers/fnn_helper.cu:420-484):
 * neuron j: acc = 0; for k ascending acc = fma(W[j][k], act[k], acc); acc += b[j]; hidden: tanh.
 * The reference's `tmp += W*act` is contracted to an FMA by nvcc; the k-ordered fma chain is also exactly what the
 * engine's MFMA formulation computes.  Blob layout [W1 (out x in) | b1 | W2 | b2 | ...] (fnn_helper.cu:176-183).
 *
 * split_output_sum (round 5; the two networks whose forward runs on the matrix cores: AutoRally 6-32-32-4 and the bicycle
 * LSTM's output net {22, 32, 4}): the OUTPUT layer's dot product is not the reference's single k-ascending chain but four
 * interleaved ones — chain g takes the inputs k with (k >> 2) & 3 == g, in ascending k — combined as (c0 + c1) + (c2 + c3),
 * then + b.  A deviation from fnn_helper.cu:458-462 in summation ORDER only (every product and the set of terms are the
 * reference's); it is what lets the engine evaluate that layer where the previous layer's outputs already sit instead of on
 * 12 padded matrix-core rows out of 16.  tests/test_fnn_output_order.py bounds its effect (u* and trajectory costs of
 * both BASELINE networks against this same oracle with the flag off); the all-ones known answers are integer-exact in any
 * order.
 */
struct FNN
{
  std::vector<int> layers;
  std::vector<float> theta;
  bool split_output_sum = false;
  void setStructure(const std::vector<int>& l)
  {
    layers = l;
    size_t n = 0;
    for (size_t i = 0; i + 1 < l.size(); i++)
      n += (size_t)l[i] * l[i + 1] + l[i + 1];
    theta.assign(n, 0.0f);
  }
  size_t numParams() const
  {
    return theta.size();
  }
  void forward(const float* in, float* out) const
  {
    std::vector<float> cur(in, in + layers[0]), nxt;
    size_t off = 0;
    for (size_t i = 0; i + 1 < layers.size(); i++)
    {
      const int n_in = layers[i], n_out = layers[i + 1];
      const float* W = &theta[off];
      const float* b = &theta[off + (size_t)n_in * n_out];
      nxt.assign(n_out, 0.0f);
      const bool last = i + 2 == layers.size();
      for (int j = 0; j < n_out; j++)
      {
        float tmp = 0.0f;
        if (last && split_output_sum)
        {
          float c[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
          for (int k = 0; k < n_in; k++)
            c[(k >> 2) & 3] = det::fma(W[(size_t)j * n_in + k], cur[k], c[(k >> 2) & 3]);
          tmp = (c[0] + c[1]) + (c[2] + c[3]);
        }
        else
        {
          for (int k = 0; k < n_in; k++)
            tmp = det::fma(W[(size_t)j * n_in + k], cur[k], tmp);
        }
        tmp += b[j];
        if (!last)
          tmp = det::tanh(tmp);
        nxt[j] = tmp;
      }
      cur = nxt;
      off += (size_t)n_in * n_out + n_out;
    }
    for (int j = 0; j < layers.back(); j++)
      out[j] = cur[j];
  }
};

/** reference: dynamics/autorally/ar_nn_model.cu:122-160 (device computeKinematics / computeDynamics), S=7, C=2, K=3 */
struct ARNeuralNetModel : Dynamics
{
  FNN net;
  ARNeuralNetModel() : Dynamics(7, 2, 8)
  {
    net.setStructure({ 6, 32, 32, 4 });
    net.split_output_sum = true;  // see FNN: the output layer's summation order of the matrix-core networks
  }
  int setParams(const void* pod, size_t n) override
  {
    return n == 0 ? 0 : -1;
  }
  int setWeights(const float* w, size_t n)
  {
    if (n != net.numParams())
      return -1;
    std::copy(w, w + n, net.theta.begin());
    return 0;
  }
  void computeKinematics(const float* state, float* state_der) override
  {
    float s, c;
    det::sincos(state[2], &s, &c);
    state_der[0] = c * state[4] - s * state[5];
    state_der[1] = s * state[4] + c * state[5];
    state_der[2] = -state[6];
  }
  void computeDynamics(const float* state, const float* control, float* state_der, float* theta_s) override
  {
    float in[6], out[4];
    for (int i = 0; i < 4; i++)
      in[i] = state[i + 3];
    in[4] = control[0];
    in[5] = control[1];
    net.forward(in, out);
    for (int i = 0; i < 4; i++)
      state_der[i + 3] = out[i];
  }
};

/* ------------------------------------------------------------------ LSTM bicycle-slip ----------------------------- */
/**
 * One-layer LSTM + output network, device flavour of LSTMHelper::forward (utils/nn_helpers/lstm_helper.cu:342-463).
 * Blob [W_im W_fm W_om W_cm (H x H) | W_ii W_fi W_oi W_ci (H x I) | b_i b_f b_o b_c | h0 | c0] (lstm_helper.cu:71-88).
 * Per unit i: g = 0; x part (j ascending), then h part (j ascending) as fma chains (nvcc contracts the reference's
 * `temp += W * x`), g += b;  sigmoid = device flavour (activation_functions.cuh:49-59);
 * c[i] = g_i * tanh(g_c) + g_f * c[i] (two roundings for the products, one for the sum), then h[i] = tanh(c[i]) * g_o[i]
 * with the NEW cell state (lstm_helper.cu:450); output = FNN([h ; x]) (lstm_helper.cu:455-462).
 */
struct LSTM
{
  int I = 0, H = 0;
  FNN out_net;
  std::vector<float> w; /* 4HH + 4HI + 4H + 2H */
  void setStructure(int input_dim, int hidden_dim, const std::vector<int>& output_layers)
  {
    I = input_dim;
    H = hidden_dim;
    out_net.setStructure(output_layers);
    w.assign((size_t)4 * H * H + 4 * H * I + 6 * H, 0.0f);
  }
  size_t numParams() const
  {
    return w.size();
  }
  const float* h0() const
  {
    return &w[(size_t)4 * H * H + 4 * H * I + 4 * H];
  }
  const float* c0() const
  {
    return h0() + H;
  }
  /** h, c: in/out [H];  out: [output dim] */
  void forward(const float* x, float* h, float* c, float* out) const
  {
    const float* W_im = w.data();
    const float* W_fm = W_im + H * H;
    const float* W_om = W_fm + H * H;
    const float* W_cm = W_om + H * H;
    const float* W_ii = W_cm + H * H;
    const float* W_fi = W_ii + H * I;
    const float* W_oi = W_fi + H * I;
    const float* W_ci = W_oi + H * I;
    const float* b_i = W_ci + H * I;
    const float* b_f = b_i + H;
    const float* b_o = b_f + H;
    const float* b_c = b_o + H;
    std::vector<float> g_o(H), c_new(H);
    for (int i = 0; i < H; i++)
    {
      float gi = 0.0f, gf = 0.0f, go = 0.0f, gc = 0.0f;
      for (int j = 0; j < I; j++)
      {
        gi = det::fma(W_ii[i * I + j], x[j], gi);
        gf = det::fma(W_fi[i * I + j], x[j], gf);
        go = det::fma(W_oi[i * I + j], x[j], go);
        gc = det::fma(W_ci[i * I + j], x[j], gc);
      }
      for (int j = 0; j < H; j++)
      {
        gi = det::fma(W_im[i * H + j], h[j], gi);
        gf = det::fma(W_fm[i * H + j], h[j], gf);
        go = det::fma(W_om[i * H + j], h[j], go);
        gc = det::fma(W_cm[i * H + j], h[j], gc);
      }
      gi += b_i[i];
      gf += b_f[i];
      go += b_o[i];
      gc += b_c[i];
      gi = det::sigmoid(gi);
      gf = det::sigmoid(gf);
      gc = det::tanh(gc);
      g_o[i] = det::sigmoid(go);
      const float in_part = gi * gc;
      const float keep_part = gf * c[i];
      c_new[i] = in_part + keep_part;
    }
    std::vector<float> act((size_t)H + I);
    for (int i = 0; i < H; i++)
    {
      c[i] = c_new[i];
      h[i] = det::tanh(c[i]) * g_o[i];
      act[i] = h[i];
    }
    for (int j = 0; j < I; j++)
      act[(size_t)H + j] = x[j];
    out_net.forward(act.data(), out);
  }
};

/**
 * LSTM bicycle-slip dynamics (BASELINE config 5; see include/mppi_amd/dynamics/bicycle_slip/bicycle_slip_lstm.hpp for
 * why this model is assembled from ar_nn_model.cu:122-178 and racer_dubins_elevation_lstm_steering.cu:117-167).
 * theta_s slot = [h (16) | c (16)], seeded from the blob's (h0, c0) by initializeDynamics.
 */
struct BicycleSlipLSTM : Dynamics
{
  LSTM net;
  BicycleSlipLSTM() : Dynamics(7, 2, 8)
  {
    net.setStructure(6, 16, { 22, 32, 4 });
    net.out_net.split_output_sum = true;  // see FNN
  }
  int setParams(const void* pod, size_t n) override
  {
    return n == 0 ? 0 : -1;
  }
  int scratchFloats() const override
  {
    return 2 * net.H;
  }
  void initializeDynamics(const float* x, const float* u, float* y, float* theta_s, float t0, float dt) override
  {
    Dynamics::initializeDynamics(x, u, y, theta_s, t0, dt);
    for (int i = 0; i < net.H; i++)
    {
      theta_s[i] = net.h0()[i];
      theta_s[net.H + i] = net.c0()[i];
    }
  }
  void computeKinematics(const float* state, float* state_der) override
  {
    float s, c;
    det::sincos(state[2], &s, &c);
    state_der[0] = c * state[4] - s * state[5];
    state_der[1] = s * state[4] + c * state[5];
    state_der[2] = -state[6];
  }
  void computeDynamics(const float* state, const float* control, float* state_der, float* theta_s) override
  {
    float in[6], out[4];
    for (int i = 0; i < 4; i++)
      in[i] = state[i + 3];
    in[4] = control[0];
    in[5] = control[1];
    net.forward(in, theta_s, theta_s + net.H, out);
    for (int i = 0; i < 4; i++)
      state_der[i + 3] = out[i];
  }
};

/** reference: cost_functions/autorally/ar_standard_cost.cu:224-243, 283-413 (device flavour) */
struct ARStandardCost : Cost
{
  mppi_ar_standard_cost_params params_{ { 0.0f, 0.0f }, 1.0f, 6.0f, 4.25f, 200.0f, 1.25f, 10.0f, 0.0f, 10000.0f, 0.65f, 10,
                                        { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
  std::vector<float> costmap; /* channel 0, [height][width] */
  int width_ = -1, height_ = -1;
  const float FRONT_D = 0.5, BACK_D = -0.5;
  ARStandardCost() : Cost(2, 8)
  {
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(params_))
      return -1;
    memcpy(&params_, pod, n);
    return 0;
  }
  int setCostmap(const float* data, int height, int width)
  {
    costmap.assign(data, data + (size_t)height * width);
    height_ = height;
    width_ = width;
    return 0;
  }
  /** reference: ar_standard_cost.cu:199-206 (known answer: tests/cost_functions/autorally_standard_cost_test.cu:184-210) */
  void coorTransform(float x, float y, float* u, float* v, float* w) const
  {
    *u = params_.r_c1[0] * x + params_.r_c2[0] * y + params_.trs[0];
    *v = params_.r_c1[1] * x + params_.r_c2[1] * y + params_.trs[1];
    *w = params_.r_c1[2] * x + params_.r_c2[2] * y + params_.trs[2];
  }
  /** reference: ar_standard_cost.cu:315-324 (known answer: autorally_standard_cost_test.cu:803-828) */
  float getCrashCost(const int* crash) const
  {
    return crash[0] > 0 ? params_.crash_coeff : 0.0f;
  }
  /** CUDA point sampling with clamp addressing and normalised coordinates: texel = clamp(floor(u * size)) */
  float queryTextureTransformed(float x, float y) const
  {
    float u, v, w;
    coorTransform(x, y, &u, &v, &w);
    const float fx = floorf(u / w * (float)width_);
    const float fy = floorf(v / w * (float)height_);
    const int ix = (fx >= 0.0f) ? ((fx < (float)width_) ? (int)fx : width_ - 1) : 0;
    const int iy = (fy >= 0.0f) ? ((fy < (float)height_) ? (int)fy : height_ - 1) : 0;
    return costmap[(size_t)iy * width_ + ix];
  }
  float getTrackCost(const float* s, int* crash) const
  {
    float sy, cy;
    det::sincos(s[2], &sy, &cy);
    const float x_front = s[0] + FRONT_D * cy, y_front = s[1] + FRONT_D * sy;
    const float x_back = s[0] + BACK_D * cy, y_back = s[1] + BACK_D * sy;
    const float tf = queryTextureTransformed(x_front, y_front);
    const float tb = queryTextureTransformed(x_back, y_back);
    float track_cost = (fabsf(tf) + fabsf(tb)) / 2.0f;
    if (fabsf(track_cost) < params_.track_slop)
      track_cost = 0;
    else
      track_cost = params_.track_coeff * track_cost;
    if (tf >= params_.boundary_threshold || tb >= params_.boundary_threshold)
      crash[0] = 1;
    return track_cost;
  }
  float getSpeedCost(const float* s) const
  {
    const float error = s[4] - params_.desired_speed;
    return params_.speed_coeff * (error * error);
  }
  float getStabilizingCost(const float* s, int* crash) const
  {
    float stabilizing_cost = 0;
    if ((double)fabsf(s[4]) > 0.001)
    {
      const float slip = -det::atan(s[5] / fabsf(s[4]));
      stabilizing_cost = params_.slip_coeff * (slip * slip);
      if (fabsf(slip) > params_.max_slip_ang)
        stabilizing_cost += params_.crash_coeff;
    }
    if ((double)fabsf(s[3]) > 1.57079632679489661923)
      crash[0] = 1;
    return stabilizing_cost;
  }
  float computeStateCost(const float* s, int timestep, int* crash) override
  {
    const float track_cost = getTrackCost(s, crash);
    const float speed_cost = getSpeedCost(s);
    const float stabilizing_cost = getStabilizingCost(s, crash);
    const float disc = params_.discount == 1.0f ? 1.0f : det::pow_pos(params_.discount, (float)timestep);
    const float crash_cost = disc * getCrashCost(crash);
    float cost = speed_cost + crash_cost + track_cost + stabilizing_cost;
    if (cost > 1e16f || cost != cost)
      cost = 1e16f;
    return cost;
  }
  float terminalCost(const float* s) override
  {
    return 0.0f;
  }
};

/* ------------------------------------------------------------------ RACER Dubins + QuadraticCost -------------------- */
/** reference: dynamics/racer_dubins/racer_dubins.cu:138-165 (device computeDynamics), :73-98 (device updateState);
 *  known answers: tests/dynamics/racer_dubins_model_test.cu:36-162 (ComputeDynamics), :321-380 (TestUpdateState) */
struct RacerDubins : Dynamics
{
  mppi_racer_dubins_params p{ { 1.3f, 2.6f, 3.9f }, { 2.5f, 3.5f, 4.5f }, { 3.7f, 4.7f, 5.7f }, 4.9f, .6f, 5, -9.1f, 0.5f, 5,
                              12.1f, 1.0f, 6.6f, 8.2f, 0.9f, 0.33f, 0.3f, 0.13f, -9.81f, 1 };
  enum
  {
    VEL_X = 0,
    YAW,
    POS_X,
    POS_Y,
    STEER_ANGLE,
    BRAKE_STATE,
    STEER_ANGLE_RATE
  };
  RacerDubins() : Dynamics(7, 2, 28)
  {
  }
  /** reference: RacerDubinsImpl::enforceLeash, dynamics/racer_dubins/racer_dubins.cu:176-240 (host; position error leashed
   *  in the body frame, yaw by the shortest angular distance) */
  void enforceLeash(const float* state_true, const float* state_nominal, const float* leash_values, float* state_output) const override
  {
    for (int i = 0; i < S; i++)
      state_output[i] = state_true[i];
    float dx = state_nominal[POS_X] - state_true[POS_X];
    float dy = state_nominal[POS_Y] - state_true[POS_Y];
    float dx_body = dx * cosf(state_true[YAW]) + dy * sinf(state_true[YAW]);
    float dy_body = -dx * sinf(state_true[YAW]) + dy * cosf(state_true[YAW]);
    float y_leash = leash_values[POS_Y];
    float x_leash = leash_values[POS_X];
    dx_body = fminf(fmaxf(dx_body, -x_leash), x_leash);
    dy_body = fminf(fmaxf(dy_body, -y_leash), y_leash);
    dx = dx_body * cosf(state_true[YAW]) + -dy_body * sinf(state_true[YAW]);
    dy = dx_body * sinf(state_true[YAW]) + dy_body * cosf(state_true[YAW]);
    state_output[POS_X] += dx;
    state_output[POS_Y] += dy;
    float diff;
    for (int i = 0; i < S; i++)
    {
      if (i == POS_X || i == POS_Y)
        continue;
      else if (i == YAW)
        diff = det::normalizeAngle(state_nominal[i] - state_true[i]); /* shortestAngularDistance, utils/angle_utils.cuh */
      else
        diff = state_nominal[i] - state_true[i];
      if (leash_values[i] < fabsf(diff))
      {
        float leash_dir = fminf(fmaxf(diff, -leash_values[i]), leash_values[i]);
        state_output[i] = state_true[i] + leash_dir;
        if (i == YAW)
          state_output[i] = det::normalizeAngle(state_output[i]);
      }
      else
      {
        state_output[i] = state_nominal[i];
      }
    }
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(p))
      return -1;
    memcpy(&p, pod, n);
    return 0;
  }
  void computeDynamics(const float* state, const float* control, float* state_der, float* theta_s) override
  {
    bool enable_brake = control[0] < 0;
    state_der[BRAKE_STATE] =
        fminf(fmaxf((enable_brake * -control[0] - state[BRAKE_STATE]) * p.brake_delay_constant, -p.max_brake_rate_neg),
              p.max_brake_rate_pos);
    state_der[VEL_X] = (!enable_brake) * p.c_t[0] * control[0] * p.gear_sign +
                       p.c_b[0] * state[BRAKE_STATE] * (state[VEL_X] >= 0 ? -1 : 1) - p.c_v[0] * state[VEL_X] + p.c_0;
    state_der[YAW] = (state[VEL_X] / p.wheel_base) * det::tan(state[STEER_ANGLE] / p.steer_angle_scale);
    float yaw, sin_yaw, cos_yaw;
    yaw = det::normalizeAngle(state[YAW]);
    det::sincos(yaw, &sin_yaw, &cos_yaw); /* reference: __sincosf */
    state_der[POS_X] = state[VEL_X] * cos_yaw;
    state_der[POS_Y] = state[VEL_X] * sin_yaw;
    state_der[STEER_ANGLE] =
        fmaxf(fminf((control[1] * p.steer_command_angle_scale - state[STEER_ANGLE]) * p.steering_constant, p.max_steer_rate),
              -p.max_steer_rate);
  }
  void updateState(const float* state, float* next_state, const float* state_der, float dt) const override
  {
    for (int i = 0; i < 6; i++)
    {
      next_state[i] = state[i] + state_der[i] * dt;
      if (i == YAW)
        next_state[i] = det::normalizeAngle(next_state[i]);
      if (i == STEER_ANGLE)
      {
        next_state[i] = fmaxf(fminf(next_state[i], p.max_steer_angle), -p.max_steer_angle);
        next_state[STEER_ANGLE_RATE] = state_der[i];
      }
      if (i == BRAKE_STATE)
        next_state[i] = fminf(fmaxf(next_state[i], 0.0f), 1.0f);
    }
  }
};

/* ------------------------------------------------------------------ RACER Dubins on an elevation map --------------- */
/**
 * reference (DEVICE flavour, paths under dynamics/racer_dubins/): racer_dubins_elevation.cu:836-874 (step), :753-798
 * (computeParametricAccelDeriv), :336-419 (computeUncertaintyJacobian), :421-506 (computeQ), :508-633 (covariance <-> state),
 * :672-738 (computeUncertaintyPropagation), :72-237 (setOutputs); racer_dubins.cu:281-305 (brake / steering lags), :69-96
 * (updateState), :358-434 (RACER::computeStaticSettling); utils/math_utils.h:457-482 (Euler2DCM_NWU), :375-391
 * (RotatePointByDCM); utils/matrix_mult_utils.cuh:82-192 (gemm1: accumulator from zero, k ascending).
 * known answers: tests/dynamics/racer_dubins_elevation_model_test.cu:305-605 (TestStep), :694-913 (TestStepReverse) — the
 * host flavour, which differs from this one only in angle wrapping (identity on the tested angles), in the math library
 * and in the side-force term of computeQ (the host branch leaves sin_roll uninitialised, :447-451); the tests do not
 * look at the covariance states.
 */
struct RacerDubinsElevation : Dynamics
{
  mppi_racer_dubins_elevation_params p{ { { 1.3f, 2.6f, 3.9f }, { 2.5f, 3.5f, 4.5f }, { 3.7f, 4.7f, 5.7f }, 4.9f, .6f, 5, -9.1f, 0.5f,
                                          5, 12.1f, 1.0f, 6.6f, 8.2f, 0.9f, 0.33f, 0.3f, 0.13f, -9.81f, 1 },
                                        5.5f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, { 41.74219f, -0.8187027f, -2.2131343f }, 0.1f, 0.001f, 0.0f };
  Texture2D map;
  std::vector<float> map_values;
  bool use_map = false;
  enum
  {
    VEL_X = 0,
    YAW,
    POS_X,
    POS_Y,
    STEER_ANGLE,
    BRAKE_STATE,
    ROLL,
    PITCH
  };
  /* the subclasses re-number what follows (racer_dubins_elevation_suspension_lstm.cuh:27-54): index of the steering rate
   * and of the first covariance entry; the ten covariance entries keep their order */
  int STEER_ANGLE_RATE = 8, UNC = 9;
  int num_euler = 6; /* states advanced by the explicit Euler step of updateState */
  enum
  {
    K_POS_X = 0,
    K_POS_Y,
    K_YAW,
    K_VEL_X,
    K_POS_X_Y,
    K_POS_X_YAW,
    K_POS_X_VEL_X,
    K_POS_Y_YAW,
    K_POS_Y_VEL_X,
    K_YAW_VEL_X
  };
  enum /* racer_dubins.cuh:37-66 */
  {
    O_BASELINK_VEL_B_X = 0,
    O_BASELINK_VEL_B_Y,
    O_BASELINK_POS_I_X,
    O_BASELINK_POS_I_Y,
    O_BASELINK_POS_I_Z,
    O_YAW,
    O_ROLL,
    O_PITCH,
    O_STEER_ANGLE,
    O_STEER_ANGLE_RATE,
    O_WHEEL_FORCE_UP_MAX,
    O_WHEEL_FORCE_FWD_MAX,
    O_WHEEL_FORCE_SIDE_MAX,
    O_ACCEL_X,
    O_ACCEL_Y,
    O_OMEGA_Z,
    O_TOTAL_VELOCITY,
    O_UNCERTAINTY_POS_X,
    O_UNCERTAINTY_POS_Y,
    O_UNCERTAINTY_YAW,
    O_UNCERTAINTY_VEL_X,
    O_UNCERTAINTY_POS_X_Y,
    O_UNCERTAINTY_POS_X_YAW,
    O_UNCERTAINTY_POS_X_VEL_X,
    O_UNCERTAINTY_POS_Y_YAW,
    O_UNCERTAINTY_POS_Y_VEL_X,
    O_UNCERTAINTY_YAW_VEL_X,
    O_FILLER_1
  };
  enum
  {
    U_VEL_X = 0,
    U_YAW,
    U_POS_X,
    U_POS_Y,
    UD
  };
  explicit RacerDubinsElevation(int num_states = 19) : Dynamics(num_states, 2, 28)
  {
  }
  /** reference: RacerDubinsImpl::enforceLeash, dynamics/racer_dubins/racer_dubins.cu:176-240 (host; position error leashed
   *  in the body frame, yaw by the shortest angular distance) */
  void enforceLeash(const float* state_true, const float* state_nominal, const float* leash_values, float* state_output) const override
  {
    for (int i = 0; i < S; i++)
      state_output[i] = state_true[i];
    float dx = state_nominal[POS_X] - state_true[POS_X];
    float dy = state_nominal[POS_Y] - state_true[POS_Y];
    float dx_body = dx * cosf(state_true[YAW]) + dy * sinf(state_true[YAW]);
    float dy_body = -dx * sinf(state_true[YAW]) + dy * cosf(state_true[YAW]);
    float y_leash = leash_values[POS_Y];
    float x_leash = leash_values[POS_X];
    dx_body = fminf(fmaxf(dx_body, -x_leash), x_leash);
    dy_body = fminf(fmaxf(dy_body, -y_leash), y_leash);
    dx = dx_body * cosf(state_true[YAW]) + -dy_body * sinf(state_true[YAW]);
    dy = dx_body * sinf(state_true[YAW]) + dy_body * cosf(state_true[YAW]);
    state_output[POS_X] += dx;
    state_output[POS_Y] += dy;
    float diff;
    for (int i = 0; i < S; i++)
    {
      if (i == POS_X || i == POS_Y)
        continue;
      else if (i == YAW)
        diff = det::normalizeAngle(state_nominal[i] - state_true[i]); /* shortestAngularDistance, utils/angle_utils.cuh */
      else
        diff = state_nominal[i] - state_true[i];
      if (leash_values[i] < fabsf(diff))
      {
        float leash_dir = fminf(fmaxf(diff, -leash_values[i]), leash_values[i]);
        state_output[i] = state_true[i] + leash_dir;
        if (i == YAW)
          state_output[i] = det::normalizeAngle(state_output[i]);
      }
      else
      {
        state_output[i] = state_nominal[i];
      }
    }
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(p))
      return -1;
    memcpy(&p, pod, n);
    return 0;
  }
  int setMap(const float* data, int height, int width)
  {
    map_values.assign(data, data + (size_t)height * width);
    map.values = map_values.data();
    map.height = height;
    map.width = width;
    use_map = true;
    return 0;
  }
  int setMapTransform(const float* t, size_t count)
  {
    if (count != 15)
      return -1;
    for (int i = 0; i < 3; i++)
      map.origin[i] = t[i];
    for (int i = 0; i < 9; i++)
      map.rot[i / 3][i % 3] = t[3 + i];
    for (int i = 0; i < 3; i++)
      map.resolution[i] = t[12 + i];
    return 0;
  }
  static int cm(int row, int col) /* matrix_mult_utils.cuh:27-30 */
  {
    return col * UD + row;
  }
  static int regime(float vx)
  {
    const float linear_brake_slope = 0.2f;
    return (fabsf(vx) > linear_brake_slope && fabsf(vx) <= 3.0f) + (fabsf(vx) > 3.0f) * 2;
  }
  void initializeDynamics(const float* x, const float* u, float* y, float* theta_s, float t0, float dt) override
  {
    Dynamics::initializeDynamics(x, u, y, theta_s, t0, dt);
    for (int i = S; i < O; i++) /* outputs step() never writes: defined (the reference leaves the buffer's content) */
      y[i] = 0.0f;
  }
  void computeDynamics(const float* x, const float* u, float* xdot, float* theta_s) override
  {
    const mppi_racer_dubins_params& b = p.base;
    /* racer_dubins.cu:281-293 */
    bool enable_brake = u[0] < 0.0f;
    const float brake_error = (enable_brake * -u[0] - x[BRAKE_STATE]);
    xdot[BRAKE_STATE] = fminf(fmaxf((brake_error > 0) * brake_error * b.brake_delay_constant +
                                        (brake_error < 0) * brake_error * b.brake_delay_constant_neg,
                                    -b.max_brake_rate_neg),
                              b.max_brake_rate_pos);
    /* racer_dubins.cu:295-305 */
    xdot[STEER_ANGLE] =
        fmaxf(fminf((u[1] * b.steer_command_angle_scale - x[STEER_ANGLE]) * b.steering_constant, b.max_steer_rate),
              -b.max_steer_rate);
    /* racer_dubins_elevation.cu:753-798 */
    float linear_brake_slope = 0.2f;
    int index = regime(x[VEL_X]);
    const float brake_state = fminf(fmaxf(x[BRAKE_STATE], 0.0f), 0.25f);
    float throttle = b.c_t[index] * u[0];
    float brake = b.c_b[index] * brake_state * (x[VEL_X] >= 0.0f ? -1.0f : 1.0f);
    if (fabsf(x[VEL_X]) <= linear_brake_slope)
    {
      throttle = b.c_t[index] * fmaxf(u[0] - b.low_min_throttle, 0.0f);
      brake = b.c_b[index] * brake_state * -x[VEL_X];
    }
    xdot[VEL_X] = (!enable_brake) * throttle * b.gear_sign + brake - b.c_v[index] * x[VEL_X] + b.c_0;
    xdot[VEL_X] = fminf(fmaxf(xdot[VEL_X], -p.clamp_ax), p.clamp_ax);
    if (fabsf(x[PITCH]) < (float)M_PI_2)
    {
      xdot[VEL_X] -= b.gravity * det::sin(det::normalizeAngle(x[PITCH])); /* __sinf */
    }
    xdot[YAW] = (x[VEL_X] / b.wheel_base) * det::tan(det::normalizeAngle(x[STEER_ANGLE] / b.steer_angle_scale)); /* __tanf */
    const float yaw_norm = det::normalizeAngle(x[YAW]);
    float sin_yaw, cos_yaw;
    det::sincos(yaw_norm, &sin_yaw, &cos_yaw); /* __cosf, __sinf */
    xdot[POS_X] = x[VEL_X] * cos_yaw;
    xdot[POS_Y] = x[VEL_X] * sin_yaw;
  }
  void updateState(const float* x, float* xn, const float* xdot, float dt) const override
  {
    const mppi_racer_dubins_params& b = p.base;
    for (int i = 0; i < num_euler; i++)
    {
      xn[i] = x[i] + xdot[i] * dt;
      if (i == YAW)
        xn[i] = det::normalizeAngle(xn[i]);
      if (i == STEER_ANGLE)
      {
        xn[i] = fmaxf(fminf(xn[i], b.max_steer_angle), -b.max_steer_angle);
        xn[STEER_ANGLE_RATE] = xdot[STEER_ANGLE];
      }
      if (i == BRAKE_STATE)
        xn[i] = fminf(fmaxf(xn[i], 0.0f), 1.0f);
    }
  }
  void uncertaintyJacobian(const float* x, float* A) const
  {
    const mppi_racer_dubins_params& b = p.base;
    float sin_yaw, cos_yaw;
    det::sincos(det::normalizeAngle(x[YAW]), &sin_yaw, &cos_yaw);
    const float delta = x[STEER_ANGLE] / b.steer_angle_scale;
    const float tan_steer_angle = det::tan(delta);
    const float cos_2_delta = ORACLE_SQ(det::cos(delta));
    const int index = regime(x[VEL_X]);
    const float brake_state = fminf(fmaxf(x[BRAKE_STATE], 0.0f), 0.25f);
    A[cm(U_VEL_X, U_VEL_X)] = -b.c_v[index] - p.K_vel_x - (index == 0 ? 1.0f : 0.0f) * b.c_b[0] * brake_state;
    A[cm(U_VEL_X, U_YAW)] = 0.0f;
    A[cm(U_VEL_X, U_POS_X)] = -p.K_x * cos_yaw;
    A[cm(U_VEL_X, U_POS_Y)] = -p.K_x * sin_yaw;
    A[cm(U_YAW, U_VEL_X)] = tan_steer_angle / (b.wheel_base);
    A[cm(U_YAW, U_YAW)] = -fabsf(x[VEL_X]) * p.K_yaw / (b.wheel_base * cos_2_delta);
    A[cm(U_YAW, U_POS_X)] = x[VEL_X] * p.K_y * sin_yaw / (b.wheel_base * cos_2_delta);
    A[cm(U_YAW, U_POS_Y)] = -x[VEL_X] * p.K_y * cos_yaw / (b.wheel_base * cos_2_delta);
    A[cm(U_POS_X, U_VEL_X)] = cos_yaw;
    A[cm(U_POS_X, U_YAW)] = -sin_yaw * x[VEL_X];
    A[cm(U_POS_X, U_POS_X)] = 0.0f;
    A[cm(U_POS_X, U_POS_Y)] = 0.0f;
    A[cm(U_POS_Y, U_VEL_X)] = sin_yaw;
    A[cm(U_POS_Y, U_YAW)] = cos_yaw * x[VEL_X];
    A[cm(U_POS_Y, U_POS_Y)] = 0.0f;
    A[cm(U_POS_Y, U_POS_X)] = 0.0f;
  }
  void computeQ(const float* x, const float* xdot, float* Q) const
  {
    const mppi_racer_dubins_params& b = p.base;
    const float abs_vx = fabsf(x[VEL_X]);
    const float abs_acc_x = fabsf(xdot[VEL_X]);
    const float delta = x[STEER_ANGLE] / b.steer_angle_scale;
    float sin_yaw, cos_yaw;
    det::sincos(det::normalizeAngle(x[YAW]), &sin_yaw, &cos_yaw);
    const float tan_steer_angle = det::tan(delta);
    const float sin_roll = det::sin(det::normalizeAngle(x[ROLL]));
    const float side_force = ORACLE_SQ(abs_vx) * tan_steer_angle / b.wheel_base + b.gravity * sin_roll;
    const float Q_11 = fabsf(p.Q_y_f * fabsf(side_force) * fmaxf(abs_vx - 2, 0.0f));
    const int index = regime(x[VEL_X]);
    for (int i = 0; i < UD * UD; i++)
      Q[i] = 0.0f;
    Q[cm(U_VEL_X, U_VEL_X)] = p.Q_x_acc * abs_acc_x + p.Q_x_v[index] * abs_vx;
    Q[cm(U_YAW, U_YAW)] = abs_vx * (p.Q_omega_steering * fabsf(delta) + p.Q_omega_v);
    Q[cm(U_POS_X, U_POS_X)] = Q_11 * sin_yaw * sin_yaw;
    Q[cm(U_POS_X, U_POS_Y)] = -Q_11 * sin_yaw * cos_yaw;
    Q[cm(U_POS_Y, U_POS_Y)] = Q_11 * cos_yaw * cos_yaw;
    Q[cm(U_POS_Y, U_POS_X)] = -Q_11 * sin_yaw * cos_yaw;
  }
  void stateToMatrix(const float* x, float* M) const
  {
    M[cm(U_VEL_X, U_VEL_X)] = x[(UNC + K_VEL_X)];
    M[cm(U_YAW, U_VEL_X)] = M[cm(U_VEL_X, U_YAW)] = x[(UNC + K_YAW_VEL_X)];
    M[cm(U_POS_X, U_VEL_X)] = M[cm(U_VEL_X, U_POS_X)] = x[(UNC + K_POS_X_VEL_X)];
    M[cm(U_POS_Y, U_VEL_X)] = M[cm(U_VEL_X, U_POS_Y)] = x[(UNC + K_POS_Y_VEL_X)];
    M[cm(U_YAW, U_YAW)] = x[(UNC + K_YAW)];
    M[cm(U_POS_X, U_YAW)] = M[cm(U_YAW, U_POS_X)] = x[(UNC + K_POS_X_YAW)];
    M[cm(U_POS_Y, U_YAW)] = M[cm(U_YAW, U_POS_Y)] = x[(UNC + K_POS_Y_YAW)];
    M[cm(U_POS_X, U_POS_X)] = x[(UNC + K_POS_X)];
    M[cm(U_POS_Y, U_POS_X)] = M[cm(U_POS_X, U_POS_Y)] = x[(UNC + K_POS_X_Y)];
    M[cm(U_POS_Y, U_POS_Y)] = x[(UNC + K_POS_Y)];
  }
  void matrixToState(const float* M, float* x) const
  {
    x[(UNC + K_VEL_X)] = M[cm(U_VEL_X, U_VEL_X)];
    x[(UNC + K_YAW_VEL_X)] = M[cm(U_YAW, U_VEL_X)];
    x[(UNC + K_POS_X_VEL_X)] = M[cm(U_POS_X, U_VEL_X)];
    x[(UNC + K_POS_Y_VEL_X)] = M[cm(U_POS_Y, U_VEL_X)];
    x[(UNC + K_YAW)] = M[cm(U_YAW, U_YAW)];
    x[(UNC + K_POS_X_YAW)] = M[cm(U_POS_X, U_YAW)];
    x[(UNC + K_POS_Y_YAW)] = M[cm(U_POS_Y, U_YAW)];
    x[(UNC + K_POS_X)] = M[cm(U_POS_X, U_POS_X)];
    x[(UNC + K_POS_X_Y)] = M[cm(U_POS_Y, U_POS_X)];
    x[(UNC + K_POS_Y)] = M[cm(U_POS_Y, U_POS_Y)];
  }
  void uncertaintyPropagation(const float* x, const float* xdot, float* xn, float dt) const
  {
    float A[UD * UD], Sigma_a[UD * UD], Sigma_b[UD * UD];
    uncertaintyJacobian(x, A);
    stateToMatrix(x, Sigma_a);
    for (int i = 0; i < UD * UD; i++)
      A[i] = (i % (UD + 1) == 0) + A[i] * dt;
    for (int pidx = 0; pidx < UD * UD; pidx++) /* gemm1(A, Sigma_a, Sigma_b) */
    {
      const int m = pidx % UD, n = pidx / UD;
      float accumulator = 0;
      for (int k = 0; k < UD; k++)
        accumulator += A[cm(m, k)] * Sigma_a[cm(k, n)];
      Sigma_b[pidx] = 1.0f * accumulator;
    }
    for (int pidx = 0; pidx < UD * UD; pidx++) /* gemm1(Sigma_b, A, Sigma_a, B transposed) */
    {
      const int m = pidx % UD, n = pidx / UD;
      float accumulator = 0;
      for (int k = 0; k < UD; k++)
        accumulator += Sigma_b[cm(m, k)] * A[k * UD + n]; /* rowMajorIndex(k, n, N) of the column-major A: A(n, k) */
      Sigma_a[pidx] = 1.0f * accumulator;
    }
    computeQ(x, xdot, Sigma_b);
    for (int i = 0; i < UD * UD; i++)
      Sigma_a[i] += Sigma_b[i] * dt;
    matrixToState(Sigma_a, xn);
  }
  void staticSettling(float yaw, float px, float py, float& roll, float& pitch, float& height) const
  {
    height = 0.0f;
    if (!use_map)
    {
      roll = 0.0f;
      pitch = 0.0f;
      return;
    }
    float sin_phi, cos_phi, sin_theta, cos_theta, sin_psi, cos_psi;
    det::sincos(det::normalizeAngle(roll), &sin_phi, &cos_phi);
    det::sincos(det::normalizeAngle(pitch), &sin_theta, &cos_theta);
    det::sincos(det::normalizeAngle(yaw), &sin_psi, &cos_psi);
    float M[3][3];
    M[0][0] = cos_theta * cos_psi;
    M[0][1] = sin_phi * sin_theta * cos_psi - cos_phi * sin_psi;
    M[0][2] = cos_phi * sin_theta * cos_psi + sin_phi * sin_psi;
    M[1][0] = cos_theta * sin_psi;
    M[1][1] = sin_phi * sin_theta * sin_psi + cos_phi * cos_psi;
    M[1][2] = cos_phi * sin_theta * sin_psi - sin_phi * cos_psi;
    M[2][0] = -sin_theta;
    M[2][1] = sin_phi * cos_theta;
    M[2][2] = cos_phi * cos_theta;
    const float body_pose[3] = { px, py, 0.0f };
    const float offsets[4][3] = { { 2.981f, 0.737f, 0.0f }, { 2.981f, -0.737f, 0.0f }, { 0.0f, 0.737f, 0.0f }, { 0.0f, -0.737f, 0.0f } };
    float h[4];
    for (int w = 0; w < 4; w++)
    {
      float world[3], mp[3], tc[3];
      for (int r = 0; r < 3; r++)
      {
        float accumulator = 0;
        for (int k = 0; k < 3; k++)
          accumulator += M[r][k] * offsets[w][k];
        world[r] = 1.0f * accumulator;
        world[r] += body_pose[r];
      }
      map.worldToMap(world, mp);
      map.mapToTex(mp, tc);
      map.query(tc, &h[w]);
    }
    const float front_left_height = h[0], front_right_height = h[1], rear_left_height = h[2], rear_right_height = h[3];
    float front_diff = front_left_height - front_right_height;
    front_diff = fmaxf(fminf(front_diff, 0.736f * 2.0f), -0.736f * 2.0f);
    float rear_diff = rear_left_height - rear_right_height;
    rear_diff = fmaxf(fminf(rear_diff, 0.736f * 2.0f), -0.736f * 2.0f);
    float front_roll = det::asin(front_diff / (0.737f * 2.0f));
    float rear_roll = det::asin(rear_diff / (0.737f * 2.0f));
    roll = (front_roll + rear_roll) / 2.0f;
    float left_diff = rear_left_height - front_left_height;
    left_diff = fmaxf(fminf(left_diff, 2.98f), -2.98f);
    float right_diff = rear_right_height - front_right_height;
    right_diff = fmaxf(fminf(right_diff, 2.98f), -2.98f);
    float left_pitch = det::asin((left_diff) / 2.981f);
    float right_pitch = det::asin((right_diff) / 2.981f);
    pitch = (left_pitch + right_pitch) / 2.0f;
    height = (rear_left_height + rear_right_height) / 2.0f;
    if (!std::isfinite(roll) || fabsf(roll) > (float)M_PI)
      roll = 2.0f * (float)M_PI;
    if (!std::isfinite(pitch) || fabsf(pitch) > (float)M_PI)
      pitch = 2.0f * (float)M_PI;
    if (!std::isfinite(height))
      height = 0.0f;
  }
  void setOutputs(const float* xdot, const float* xn, float* y) const
  {
    y[O_BASELINK_VEL_B_X] = xn[VEL_X];
    y[O_BASELINK_VEL_B_Y] = 0.0f;
    y[O_BASELINK_POS_I_X] = xn[POS_X];
    y[O_BASELINK_POS_I_Y] = xn[POS_Y];
    y[O_PITCH] = xn[PITCH];
    y[O_ROLL] = xn[ROLL];
    y[O_YAW] = xn[YAW];
    y[O_STEER_ANGLE] = xn[STEER_ANGLE];
    y[O_STEER_ANGLE_RATE] = xn[STEER_ANGLE_RATE];
    y[O_WHEEL_FORCE_UP_MAX] = NAN;
    y[O_WHEEL_FORCE_FWD_MAX] = NAN;
    y[O_WHEEL_FORCE_SIDE_MAX] = NAN;
    y[O_ACCEL_X] = xdot[VEL_X];
    y[O_ACCEL_Y] = 0.0f;
    y[O_OMEGA_Z] = xdot[YAW];
    y[O_UNCERTAINTY_VEL_X] = xn[(UNC + K_VEL_X)];
    y[O_UNCERTAINTY_YAW_VEL_X] = xn[(UNC + K_YAW_VEL_X)];
    y[O_UNCERTAINTY_POS_X_VEL_X] = xn[(UNC + K_POS_X_VEL_X)];
    y[O_UNCERTAINTY_POS_Y_VEL_X] = xn[(UNC + K_POS_Y_VEL_X)];
    y[O_UNCERTAINTY_YAW] = xn[(UNC + K_YAW)];
    y[O_UNCERTAINTY_POS_X_YAW] = xn[(UNC + K_POS_X_YAW)];
    y[O_UNCERTAINTY_POS_Y_YAW] = xn[(UNC + K_POS_Y_YAW)];
    y[O_UNCERTAINTY_POS_X] = xn[(UNC + K_POS_X)];
    y[O_UNCERTAINTY_POS_X_Y] = xn[(UNC + K_POS_X_Y)];
    y[O_UNCERTAINTY_POS_Y] = xn[(UNC + K_POS_Y)];
    y[O_TOTAL_VELOCITY] = fabsf(xn[VEL_X]);
  }
  void step(float* x, float* xn, float* xdot, const float* u, float* y, float* theta_s, int t, float dt) override
  {
    computeDynamics(x, u, xdot, theta_s);
    updateState(x, xn, xdot, dt);
    uncertaintyPropagation(x, xdot, xn, dt);
    float roll = x[ROLL], pitch = x[PITCH], height;
    staticSettling(xn[YAW], xn[POS_X], xn[POS_Y], roll, pitch, height);
    y[O_BASELINK_POS_I_Z] = height;
    xn[PITCH] = pitch;
    xn[ROLL] = roll;
    setOutputs(xdot, xn, y);
  }
};

/**
 * reference (DEVICE flavour): dynamics/racer_dubins/racer_dubins_elevation_lstm_steering.cu:131-167 (computeLSTMSteering),
 * :169-213 (step), :115-129 (initializeDynamics), :243-268 (updateState); the rest is RacerDubinsElevation above.
 * The reference's own known-answer tests for this class are disabled at this snapshot (TestStep: GTEST_SKIP,
 * tests/dynamics/racer_dubins_elevation_lstm_steering_model_test.cu:322-324; ComputeDynamics / TestUpdateState commented
 * out) and its network files are git-LFS stubs, so the pins are: the live property test compareToElevationWithoutSteering
 * (:775-945 — with a zero network every state, derivative and output except the steering ones equals the plain elevation
 * model's), the LSTMHelper known answers (tests/test_lstm_helper.py) and the steering equations checked by hand below.
 */
struct RacerDubinsElevationLSTMSteering : RacerDubinsElevation
{
  LSTM net;
  explicit RacerDubinsElevationLSTMSteering(int num_states = 19) : RacerDubinsElevation(num_states)
  {
    net.setStructure(4, 4, { 8, 20, 1 });
  }
  int setStructure(const float* desc, size_t n)
  {
    if (n < 3 || (int)desc[1] != (int)desc[0] + 4)
      return -1;
    std::vector<int> layers;
    for (size_t i = 1; i < n; i++)
      layers.push_back((int)desc[i]);
    net.setStructure(4, (int)desc[0], layers);
    return 0;
  }
  int scratchFloats() const override
  {
    return 2 * net.H;
  }
  void initializeDynamics(const float* x, const float* u, float* y, float* theta_s, float t0, float dt) override
  {
    for (int i = 0; i < net.H; i++)
    {
      theta_s[i] = net.h0()[i];
      theta_s[net.H + i] = net.c0()[i];
    }
    y[O_BASELINK_POS_I_Z] = 0.0f; /* not written by setOutputs: defined (the reference leaves the buffer's content) */
    y[O_FILLER_1] = 0.0f;
    setOutputs(x, x, y);
  }
  /** racer_dubins_elevation_lstm_steering.cu:131-167 */
  void lstmSteering(const float* x, const float* u, float* xdot, float* theta_s)
  {
    const mppi_racer_dubins_params& b = p.base;
    const float parametric_accel = (u[1] * b.steer_command_angle_scale - x[STEER_ANGLE]) * b.steering_constant;
    xdot[STEER_ANGLE_RATE] =
        fmaxf(fminf((parametric_accel - x[STEER_ANGLE_RATE]) * b.steer_accel_constant - x[STEER_ANGLE_RATE] * b.steer_accel_drag_constant,
                    b.max_steer_rate),
              -b.max_steer_rate);
    float in[4];
    in[0] = x[STEER_ANGLE] * 0.2f;
    in[1] = x[STEER_ANGLE_RATE] * 0.2f;
    in[2] = u[1];
    in[3] = xdot[STEER_ANGLE_RATE] * 0.2f;
    std::vector<float> nn_out(net.out_net.layers.back());
    net.forward(in, theta_s, theta_s + net.H, nn_out.data());
    xdot[STEER_ANGLE_RATE] += nn_out[0] * 5.0f;
    xdot[STEER_ANGLE] = x[STEER_ANGLE_RATE];
  }
  void step(float* x, float* xn, float* xdot, const float* u, float* y, float* theta_s, int t, float dt) override
  {
    computeDynamics(x, u, xdot, theta_s); /* brake lag and acceleration; the first-order steering lag is replaced below */
    lstmSteering(x, u, xdot, theta_s);
    updateState(x, xn, xdot, dt);
    xn[STEER_ANGLE_RATE] = x[STEER_ANGLE_RATE] + xdot[STEER_ANGLE_RATE] * dt;
    uncertaintyPropagation(x, xdot, xn, dt);
    float roll = x[ROLL], pitch = x[PITCH], height;
    staticSettling(xn[YAW], xn[POS_X], xn[POS_Y], roll, pitch, height);
    y[O_BASELINK_POS_I_Z] = height;
    xn[PITCH] = pitch;
    xn[ROLL] = roll;
    setOutputs(xdot, xn, y);
  }
};

/**
 * reference (DEVICE flavour): dynamics/racer_dubins/racer_dubins_elevation_suspension_lstm.cu:199-340
 * (computeSimpleSuspensionStep), :342-392 (step), :394-417 (updateState), :437-525 (setOutputs); parameters and the 24-entry
 * state layout: racer_dubins_elevation_suspension_lstm.cuh:17-66.  The wheel contributions to the three accelerations are
 * summed in the order FL, FR, BL, BR (the reference: atomicAdd, no fixed order).  The reference's tests for this class
 * compare its GPU and CPU paths with each other on random data (tests/dynamics/racer_dubins_elevation_suspension_test.cu);
 * there is no known answer to pin on — see tests/test_racer_dubins_suspension.py for what is checked instead.
 */
struct RacerDubinsElevationSuspension : RacerDubinsElevationLSTMSteering
{
  enum
  {
    CG_POS_Z = 8,
    CG_VEL_I_Z,
    ROLL_RATE,
    PITCH_RATE,
    FILLER_1 = 23
  };
  mppi_racer_dubins_suspension_params sp;
  Texture2D normals;
  std::vector<float> normals_values;
  bool use_normals = false, normals_transform_set = false;
  explicit RacerDubinsElevationSuspension(int num_states = 24) : RacerDubinsElevationLSTMSteering(num_states)
  {
    STEER_ANGLE_RATE = 12;
    UNC = 13;
    num_euler = 12;
    sp.elevation = p;
    sp.spring_k = 14000.0f;
    sp.drag_c = 1000.0f;
    sp.mass = 1447.0f;
    sp.I_xx = 1.0f / 12 * sp.mass * 2 * ORACLE_SQ(1.5f);
    sp.I_yy = 1.0f / 12 * sp.mass * (ORACLE_SQ(1.5f) + ORACLE_SQ(3.0f));
    sp.wheel_radius = 0.32f;
    sp.c_g[0] = 2.981f * 0.5f;
    sp.c_g[1] = 0.0f;
    sp.c_g[2] = 0.0f;
    normals.channels = 4;
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(sp))
      return -1;
    memcpy(&sp, pod, n);
    p = sp.elevation;
    return 0;
  }
  int setNormals(const float* data, int height, int width)
  {
    normals_values.assign(data, data + (size_t)height * width * 4);
    normals.values = normals_values.data();
    normals.height = height;
    normals.width = width;
    use_normals = true;
    return 0;
  }
  void copyFrame(const Texture2D& from, Texture2D& to)
  {
    for (int i = 0; i < 3; i++)
    {
      to.origin[i] = from.origin[i];
      to.resolution[i] = from.resolution[i];
      for (int j = 0; j < 3; j++)
        to.rot[i][j] = from.rot[i][j];
    }
  }
  int setNormalsTransform(const float* t, size_t count)
  {
    if (count != 15)
      return -1;
    for (int i = 0; i < 3; i++)
      normals.origin[i] = t[i];
    for (int i = 0; i < 9; i++)
      normals.rot[i / 3][i % 3] = t[3 + i];
    for (int i = 0; i < 3; i++)
      normals.resolution[i] = t[12 + i];
    normals_transform_set = true;
    return 0;
  }
  void suspensionStep(const float* x, float* xdot, float* y) const
  {
    xdot[ROLL] = x[ROLL_RATE];
    xdot[PITCH] = x[PITCH_RATE];
    xdot[CG_POS_Z] = x[CG_VEL_I_Z];
    const float roll = x[ROLL], pitch = x[PITCH], yaw = x[YAW];
    /* Euler2DCM_NWU, device branch (utils/math_utils.h:457-482) */
    float sin_phi, cos_phi, sin_theta, cos_theta, sin_psi, cos_psi;
    det::sincos(det::normalizeAngle(roll), &sin_phi, &cos_phi);
    det::sincos(det::normalizeAngle(pitch), &sin_theta, &cos_theta);
    det::sincos(det::normalizeAngle(yaw), &sin_psi, &cos_psi);
    float M[3][3];
    M[0][0] = cos_theta * cos_psi;
    M[0][1] = sin_phi * sin_theta * cos_psi - cos_phi * sin_psi;
    M[0][2] = cos_phi * sin_theta * cos_psi + sin_phi * sin_psi;
    M[1][0] = cos_theta * sin_psi;
    M[1][1] = sin_phi * sin_theta * sin_psi + cos_phi * cos_psi;
    M[1][2] = cos_phi * sin_theta * sin_psi - sin_phi * cos_psi;
    M[2][0] = -sin_theta;
    M[2][1] = sin_phi * cos_theta;
    M[2][2] = cos_phi * cos_theta;
    /* FL, FR, BL, BR (:249-269; rear-right at y = +0.737, rear-left at -0.737, as the reference) */
    const float body[4][3] = { { 2.981f, 0.737f, 0.0f }, { 2.981f, -0.737f, 0.0f }, { 0.0f, -0.737f, 0.0f }, { 0.0f, 0.737f, 0.0f } };
    const float body_pose[3] = { x[POS_X], x[POS_Y], 0.0f };
    float wheel_height = 0.0f;
    float n[4] = { 0.0f, 0.0f, 1.0f, 0.0f };
    float up[4], fwd[4], side[4];
    float acc_z = 0.0f, acc_roll = 0.0f, acc_pitch = 0.0f;
    for (int i = 0; i < 4; i++)
    {
      float wheel_yaw = yaw;
      if (i < 2)
        wheel_yaw += STEER_ANGLE / -9.1f; /* the state INDEX (4), :253 and :257 */
      float sin_wheel_yaw, cos_wheel_yaw;
      det::sincos(wheel_yaw, &sin_wheel_yaw, &cos_wheel_yaw); /* __sincosf */
      const float cg[3] = { body[i][0] - sp.c_g[0], body[i][1] - sp.c_g[1], body[i][2] - sp.c_g[2] };
      float world[3], mp[3], tc[3];
      for (int r = 0; r < 3; r++)
      {
        float accumulator = 0;
        for (int k = 0; k < 3; k++)
          accumulator += M[r][k] * body[i][k];
        world[r] = 1.0f * accumulator;
        world[r] += body_pose[r];
      }
      if (use_map)
      {
        map.worldToMap(world, mp);
        map.mapToTex(mp, tc);
        map.query(tc, &wheel_height);
        if (!std::isfinite(wheel_height))
          wheel_height = x[CG_POS_Z] - sp.wheel_radius;
      }
      if (use_normals)
      {
        normals.worldToMap(world, mp);
        normals.mapToTex(mp, tc);
        normals.query(tc, n);
        if (!std::isfinite(n[0]) || !std::isfinite(n[1]) || !std::isfinite(n[2]))
        {
          n[0] = 0.0f;
          n[1] = 0.0f;
          n[2] = 1.0f;
          n[3] = 0.0f;
        }
      }
      const float wheel_pos_z = x[CG_POS_Z] + roll * cg[1] - pitch * cg[0] - sp.wheel_radius;
      const float wheel_vel_z = x[CG_VEL_I_Z] + x[ROLL_RATE] * cg[1] - x[PITCH_RATE] * cg[0];
      const float h_dot = -(x[VEL_X] * cos_wheel_yaw * n[0] + x[VEL_X] * sin_wheel_yaw * n[1]);
      float wheel_force = -sp.spring_k * (wheel_pos_z - wheel_height) - sp.drag_c * (wheel_vel_z - h_dot);
      float fwd_wheel_force = wheel_force / n[2] * (n[0] * cos_wheel_yaw + n[1] * sin_wheel_yaw + n[2] * (-pitch));
      float side_wheel_force = wheel_force / n[2] * (-n[0] * sin_wheel_yaw + n[1] * cos_wheel_yaw + n[2] * roll);
      up[i] = wheel_force;
      fwd[i] = fabsf(fwd_wheel_force);
      side[i] = fabsf(side_wheel_force);
      acc_z += wheel_force / sp.mass;
      acc_roll += wheel_force * cg[1] / sp.I_xx;
      acc_pitch += -wheel_force * cg[0] / sp.I_yy;
    }
    xdot[CG_VEL_I_Z] = acc_z;
    xdot[ROLL_RATE] = acc_roll;
    xdot[PITCH_RATE] = acc_pitch;
    y[O_WHEEL_FORCE_UP_MAX] = fmaxf(up[0], fmaxf(up[1], fmaxf(up[2], up[3])));
    y[O_WHEEL_FORCE_FWD_MAX] = fmaxf(fwd[0], fmaxf(fwd[1], fmaxf(fwd[2], fwd[3])));
    y[O_WHEEL_FORCE_SIDE_MAX] = fmaxf(side[0], fmaxf(side[1], fmaxf(side[2], side[3])));
  }
  void setSuspensionOutputs(const float* xdot, const float* xn, float* y) const
  {
    const float fwd = y[O_WHEEL_FORCE_FWD_MAX], side = y[O_WHEEL_FORCE_SIDE_MAX], up = y[O_WHEEL_FORCE_UP_MAX];
    setOutputs(xdot, xn, y); /* the elevation model's list ... */
    y[O_WHEEL_FORCE_UP_MAX] = up; /* ... without its NaN wheel forces (:437-525 has no such case) */
    y[O_WHEEL_FORCE_FWD_MAX] = fwd;
    y[O_WHEEL_FORCE_SIDE_MAX] = side;
    y[O_BASELINK_POS_I_Z] = xn[CG_POS_Z] - xn[PITCH] * (-sp.c_g[0]);
  }
  void step(float* x, float* xn, float* xdot, const float* u, float* y, float* theta_s, int t, float dt) override
  {
    if (!normals_transform_set)
      copyFrame(map, normals);
    computeDynamics(x, u, xdot, theta_s); /* brake lag, acceleration, yaw and position rates */
    lstmSteering(x, u, xdot, theta_s);
    suspensionStep(x, xdot, y);
    xn[FILLER_1] = x[FILLER_1];
    updateState(x, xn, xdot, dt);
    xn[STEER_ANGLE_RATE] = x[STEER_ANGLE_RATE] + xdot[STEER_ANGLE_RATE] * dt;
    uncertaintyPropagation(x, xdot, xn, dt);
    setSuspensionOutputs(xdot, xn, y);
  }
};

/**
 * reference (DEVICE flavour): dynamics/racer_dubins/racer_dubins_elevation_lstm_unc.cu:496-605 (step), :300-494 (computeQ),
 * :607-618 (initializeDynamics); parameters and the 26-entry state layout: racer_dubins_elevation_lstm_unc.cuh:5-48.
 * Network inputs the step does not fill are zero (the reference's host path: getZeroInputVector).  The reference's tests
 * for this class compare GPU and CPU paths on random data or need a network file that is an LFS stub here
 * (TestMatchesPython): no known answer to pin on — see tests/test_racer_dubins_lstm_unc.py; the whole step is checked against
 * an independent float64 restatement written from the reference's source (tests/test_racer_complete_step_f64.py, round 3).
 * In reverse gear the reference's computeQ calls the parent's and then — without a return — overwrites every entry with the
 * network's (:305-309): the network's process noise applies in both gears, as here.
 */
struct RacerDubinsElevationLSTMUncertainty : RacerDubinsElevationSuspension
{
  enum
  {
    OMEGA_Z = 13,
    STATIC_ROLL = 14,
    STATIC_PITCH = 15
  };
  mppi_racer_dubins_uncertainty_params up;
  LSTM mean_net, unc_net;
  RacerDubinsElevationLSTMUncertainty() : RacerDubinsElevationSuspension(26)
  {
    STEER_ANGLE_RATE = 12;
    UNC = 16;
    num_euler = 12;
    mean_net.setStructure(12, 4, { 16, 20, 2 });
    unc_net.setStructure(13, 4, { 17, 20, 5 });
    up.suspension = sp;
    for (int i = 0; i < 7; i++)
      up.unc_scale[i] = 1.0f;
    const float pq[3] = { 2.0f, 0.5f, 0.3f }, nq[3] = { 5.84f, 0.15f, 1.7f };
    for (int i = 0; i < 3; i++)
    {
      up.pos_quad_brake_c[i] = pq[i];
      up.neg_quad_brake_c[i] = nq[i];
    }
    up.use_static_settling = 1;
  }
  /** another hidden size / output network for the mean (which = 1) or the uncertainty (2) network: {H, H + I, ..., outputs};
   *  the input and output sizes are the model's (12 -> 2, 13 -> 5) */
  int setNetworkStructure(int which, const float* desc, size_t n)
  {
    const int I = which == 1 ? 12 : 13, OUT = which == 1 ? 2 : 5;
    if (n < 3 || (int)desc[1] != (int)desc[0] + I || (int)desc[n - 1] != OUT)
      return -1;
    std::vector<int> layers;
    for (size_t i = 1; i < n; i++)
      layers.push_back((int)desc[i]);
    (which == 1 ? mean_net : unc_net).setStructure(I, (int)desc[0], layers);
    return 0;
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(up))
      return -1;
    memcpy(&up, pod, n);
    sp = up.suspension;
    p = sp.elevation;
    return 0;
  }
  int scratchFloats() const override
  {
    return 2 * net.H + 2 * mean_net.H + 2 * unc_net.H;
  }
  float* meanState(float* theta_s) const
  {
    return theta_s + 2 * net.H;
  }
  float* uncState(float* theta_s) const
  {
    return theta_s + 2 * net.H + 2 * mean_net.H;
  }
  void initializeDynamics(const float* x, const float* u, float* y, float* theta_s, float t0, float dt) override
  {
    RacerDubinsElevationSuspension::initializeDynamics(x, u, y, theta_s, t0, dt);
    for (int i = 0; i < mean_net.H; i++)
    {
      meanState(theta_s)[i] = mean_net.h0()[i];
      meanState(theta_s)[mean_net.H + i] = mean_net.c0()[i];
    }
    for (int i = 0; i < unc_net.H; i++)
    {
      uncState(theta_s)[i] = unc_net.h0()[i];
      uncState(theta_s)[unc_net.H + i] = unc_net.c0()[i];
    }
  }
  void networkQ(const float* x, const float* u, const float* xdot, float* theta_s, float* Q)
  {
    const mppi_racer_dubins_params& b = p.base;
    float sin_yaw, cos_yaw;
    det::sincos(det::normalizeAngle(x[YAW]), &sin_yaw, &cos_yaw);
    float in[13];
    in[0] = x[VEL_X];
    in[1] = x[OMEGA_Z];
    in[2] = x[BRAKE_STATE];
    in[3] = x[STEER_ANGLE];
    in[4] = x[STEER_ANGLE_RATE];
    in[5] = u[0] >= 0.0f ? u[0] : 0.0f;
    in[6] = u[0] <= 0.0f ? -u[0] : 0.0f;
    in[7] = u[1];
    in[8] = det::sin(x[STATIC_ROLL]);  /* __sinf */
    in[9] = det::sin(x[STATIC_PITCH]);
    in[10] = xdot[VEL_X];
    in[11] = xdot[YAW];
    in[12] = 0.0f;
    float o[5];
    unc_net.forward(in, uncState(theta_s), uncState(theta_s) + unc_net.H, o);
    for (int i = 0; i < 5; i++)
      o[i] = fabsf(det::sigmoid(o[i]) * up.unc_scale[i]);
    const int index = regime(x[VEL_X]);
    for (int i = 0; i < UD * UD; i++)
      Q[i] = 0.0f;
    Q[cm(U_VEL_X, U_VEL_X)] = o[0] + ORACLE_SQ(b.c_b[index] * (index == 0 ? x[VEL_X] : 1.0f)) * o[4];
    Q[cm(U_YAW, U_YAW)] =
        o[1] + ORACLE_SQ((x[VEL_X] / b.wheel_base) * 1.0f / (ORACLE_SQ(det::cos(x[STEER_ANGLE] / b.steer_angle_scale)) * b.steer_angle_scale)) * o[3];
    Q[cm(U_POS_X, U_POS_X)] = o[2] * sin_yaw * sin_yaw;
    Q[cm(U_POS_X, U_POS_Y)] = -o[2] * sin_yaw * cos_yaw;
    Q[cm(U_POS_Y, U_POS_Y)] = o[2] * cos_yaw * cos_yaw;
    Q[cm(U_POS_Y, U_POS_X)] = -o[2] * sin_yaw * cos_yaw;
  }
  void step(float* x, float* xn, float* xdot, const float* u, float* y, float* theta_s, int t, float dt) override
  {
    const mppi_racer_dubins_params& b = p.base;
    if (!normals_transform_set)
      copyFrame(map, normals);
    computeDynamics(x, u, xdot, theta_s); /* acceleration, yaw and position rates (the two lags are replaced below) */
    bool enable_brake = u[0] < 0.0f;
    const float brake_error = (enable_brake * -u[0] - x[BRAKE_STATE]);
    xdot[BRAKE_STATE] = fminf(
        fmaxf((brake_error > 0) * (brake_error * up.pos_quad_brake_c[0] + brake_error * fabsf(brake_error) * up.pos_quad_brake_c[1]) +
                  (brake_error < 0) * (brake_error * up.neg_quad_brake_c[0] + brake_error * fabsf(brake_error) * up.neg_quad_brake_c[1]),
              -b.max_brake_rate_neg),
        b.max_brake_rate_pos);
    lstmSteering(x, u, xdot, theta_s);
    suspensionStep(x, xdot, y);
    if (b.gear_sign == 1)
    {
      float in[12], mean_output[2];
      in[0] = x[VEL_X];
      in[1] = x[OMEGA_Z];
      in[2] = x[BRAKE_STATE];
      in[3] = x[STEER_ANGLE];
      in[4] = x[STEER_ANGLE_RATE];
      in[5] = u[0] >= 0.0f ? u[0] : 0.0f;
      in[6] = u[0] <= 0.0f ? -u[0] : 0.0f;
      in[7] = u[1];
      in[8] = det::sin(x[STATIC_PITCH]);
      in[9] = xdot[VEL_X];
      in[10] = xdot[YAW];
      in[11] = 0.0f;
      mean_net.forward(in, meanState(theta_s), meanState(theta_s) + mean_net.H, mean_output);
      xdot[VEL_X] += mean_output[0];
      xdot[YAW] += mean_output[1];
    }
    xn[OMEGA_Z] = xdot[YAW];
    updateState(x, xn, xdot, dt);
    xn[STEER_ANGLE_RATE] = x[STEER_ANGLE_RATE] + xdot[STEER_ANGLE_RATE] * dt;
    /* computeUncertaintyPropagation with this class's computeQ */
    {
      float A[UD * UD], Sigma_a[UD * UD], Sigma_b[UD * UD];
      uncertaintyJacobian(x, A);
      stateToMatrix(x, Sigma_a);
      for (int i = 0; i < UD * UD; i++)
        A[i] = (i % (UD + 1) == 0) + A[i] * dt;
      for (int pidx = 0; pidx < UD * UD; pidx++)
      {
        const int m = pidx % UD, n = pidx / UD;
        float accumulator = 0;
        for (int k = 0; k < UD; k++)
          accumulator += A[cm(m, k)] * Sigma_a[cm(k, n)];
        Sigma_b[pidx] = 1.0f * accumulator;
      }
      for (int pidx = 0; pidx < UD * UD; pidx++)
      {
        const int m = pidx % UD, n = pidx / UD;
        float accumulator = 0;
        for (int k = 0; k < UD; k++)
          accumulator += Sigma_b[cm(m, k)] * A[k * UD + n];
        Sigma_a[pidx] = 1.0f * accumulator;
      }
      networkQ(x, u, xdot, theta_s, Sigma_b);
      for (int i = 0; i < UD * UD; i++)
        Sigma_a[i] += Sigma_b[i] * dt;
      matrixToState(Sigma_a, xn);
    }
    float roll = x[STATIC_ROLL], pitch = x[STATIC_PITCH], height;
    staticSettling(xn[YAW], xn[POS_X], xn[POS_Y], roll, pitch, height);
    xn[STATIC_PITCH] = pitch;
    xn[STATIC_ROLL] = roll;
    setSuspensionOutputs(xdot, xn, y);
  }
};

/** reference: cost_functions/quadratic_cost/quadratic_cost.cu:39-60 (device), SIM_TIME_HORIZON = 1 */
struct QuadraticCost28 : Cost
{
  mppi_quadratic_cost_params_28 params_;
  bool skip_zero_coeff = false; /* the product's SKIP_ZERO_COEFF form (quadratic_cost.hpp): coefficient 0 -> exactly 0 */
  explicit QuadraticCost28(bool skip = false) : Cost(2, 28), skip_zero_coeff(skip)
  {
    memset(&params_, 0, sizeof(params_));
    params_.discount = 1.0f;
    for (int i = 0; i < 28; i++)
      params_.s_coeffs[i] = 1.0f;
  }
  int setParams(const void* pod, size_t n) override
  {
    if (n != sizeof(params_))
      return -1;
    memcpy(&params_, pod, n);
    return 0;
  }
  float computeStateCost(const float* s, int timestep, int* crash) override
  {
    float cost = 0;
    for (int i = 0; i < 28; i++)
    {
      const float term = ((s[i] - params_.s_goal[i]) * (s[i] - params_.s_goal[i])) * params_.s_coeffs[i]; /* powf(x, 2) */
      cost += (skip_zero_coeff && params_.s_coeffs[i] == 0.0f) ? 0.0f : term;
    }
    return cost;
  }
  float terminalCost(const float* s) override
  {
    return 0.0f;
  }
};

inline bool makeModel(const std::string& name, std::unique_ptr<Dynamics>& dyn, std::unique_ptr<Cost>& cost)
{
  if (name == "bicycle_slip_lstm")
  {
    dyn.reset(new BicycleSlipLSTM());
    cost.reset(new ARStandardCost());
    return true;
  }
  if (name == "autorally_nn")
  {
    dyn.reset(new ARNeuralNetModel());
    cost.reset(new ARStandardCost());
    return true;
  }
  if (name == "cartpole")
  {
    dyn.reset(new CartpoleDynamics());
    cost.reset(new CartpoleQuadraticCost());
    return true;
  }
  if (name == "double_integrator_robust")
  {
    dyn.reset(new DoubleIntegratorDynamics());
    cost.reset(new DoubleIntegratorRobustCost());
    return true;
  }
  if (name == "double_integrator")
  {
    dyn.reset(new DoubleIntegratorDynamics());
    cost.reset(new DoubleIntegratorCircleCost());
    return true;
  }
  if (name == "racer_dubins_elevation_lstm_unc")
  {
    dyn.reset(new RacerDubinsElevationLSTMUncertainty());
    cost.reset(new QuadraticCost28(true));
    return true;
  }
  if (name == "racer_dubins_elevation_suspension")
  {
    dyn.reset(new RacerDubinsElevationSuspension());
    cost.reset(new QuadraticCost28(true));
    return true;
  }
  if (name == "racer_dubins_elevation_lstm_steering")
  {
    dyn.reset(new RacerDubinsElevationLSTMSteering());
    cost.reset(new QuadraticCost28(true));
    return true;
  }
  if (name == "racer_dubins_elevation")
  {
    dyn.reset(new RacerDubinsElevation());
    cost.reset(new QuadraticCost28(true));
    return true;
  }
  if (name == "racer_dubins")
  {
    dyn.reset(new RacerDubins());
    cost.reset(new QuadraticCost28());
    return true;
  }
  return false;
}

}  // namespace oracle

#endif  // MPPI_ORACLE_MODELS_HPP_
