/**
 * lstm_wave.hpp — LSTM step + output MLP of the NN dynamics for ONE rollout on one wave (see fnn_wave.hpp for the why: the
 * re-rollout of the optimised control sequence is a single chain of T dependent steps; the MFMA form is built for 16
 * rollouts at a time).
 *
 * H = 16 hidden units x 4 gates = the 64 lanes of the wave: lane l evaluates gate l / 16 (i, f, o, c~) of unit l % 16,
 *     acc = fma(W_x[row][k], x[k], acc) over the inputs, then fma(W_h[row][k], h[k], acc) over the hidden state, then + b
 * — the reference's accumulation order, input part first (include/mppi/utils/nn_helpers/lstm_helper.cu:412-431), which is
 * also what the MFMA form's k-steps realise — with h[k] read from lane k (v_readlane).  Four cross-lane reads bring the
 * gates of a unit together; every lane group keeps a copy of (h, c).  Output MLP [h ; x] -> M (tanh) -> OUT with lane j =
 * neuron j (k order h first, then x: lstm_helper.cu:455-462).  Same fma chains, same det:: activations: the same bits as
 * lstm_mfma.hpp and lstm_helper.hpp.
 *
 * Restrictions: H == 16, M <= 64, OUT <= 64.
 */
#ifndef MPPI_AMD_LSTM_WAVE_HPP_
#define MPPI_AMD_LSTM_WAVE_HPP_

#include <hip/hip_runtime.h>
#include "mppi_amd/det_math.h"

namespace mppi
{
template <int I, int H, int M, int OUT>
struct LSTMWave
{
  static_assert(4 * H == 64 && M <= 64 && OUT <= 64, "unsupported LSTM shape for the one-rollout-per-wave forward");
  static constexpr int LSTM_NUM_PARAMS = 4 * H * H + 4 * H * I + 4 * H;

  float wx[I], wh[H], bg;    ///< this lane's gate row
  float w1h[H], w1x[I], b1;  ///< MLP layer 1, row = lane (< M)
  float w2[M], b2;           ///< MLP layer 2, row = lane (< OUT)
  float h, c;                ///< recurrent state of unit lane % H (held by the four lanes of the unit)

  /** lstm: [W_im W_fm W_om W_cm | W_ii W_fi W_oi W_ci | b_i b_f b_o b_c | h0 | c0];  fnn: [W1 | b1 | W2 | b2] */
  __device__ inline void load(const float* __restrict__ lstm, const float* __restrict__ fnn, const int lane)
  {
    const int gate = lane / H, u = lane % H;
    const float* Wm = lstm;
    const float* Wi = lstm + 4 * H * H;
    const float* B = Wi + 4 * H * I;
    const float* h0 = lstm + LSTM_NUM_PARAMS;
    const float* c0 = h0 + H;
#pragma unroll
    for (int k = 0; k < I; k++)
      wx[k] = Wi[gate * H * I + u * I + k];
#pragma unroll
    for (int k = 0; k < H; k++)
      wh[k] = Wm[gate * H * H + u * H + k];
    bg = B[gate * H + u];
    const float* W1 = fnn;
    const float* B1 = W1 + (H + I) * M;
    const float* W2 = B1 + M;
    const float* B2 = W2 + M * OUT;
    const int j = lane < M ? lane : 0;  // lanes beyond a layer repeat neuron 0: nobody reads them
    const int o = lane < OUT ? lane : 0;
#pragma unroll
    for (int k = 0; k < H; k++)
      w1h[k] = W1[j * (H + I) + k];
#pragma unroll
    for (int k = 0; k < I; k++)
      w1x[k] = W1[j * (H + I) + H + k];
    b1 = B1[j];
#pragma unroll
    for (int k = 0; k < M; k++)
      w2[k] = W2[o * M + k];
    b2 = B2[o];
    h = h0[u];
    c = c0[u];
  }

  __device__ static inline float lane_value(const float v, const int k)
  {
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), k));
  }

  /** in[I]: the network input (the same in every lane); out[OUT]: the network output, the same in every lane */
  __device__ inline void forward(const float (&in)[I], float (&out)[OUT], const int lane)
  {
    const int gate = lane / H, u = lane % H;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < I; k++)
      acc = mppi::det::fma(wx[k], in[k], acc);
#pragma unroll
    for (int k = 0; k < H; k++)
      acc = mppi::det::fma(wh[k], lane_value(h, k), acc);
    const float pre = acc + bg;
    // gates i, f, o: sigmoid(x) = fma(tanh(x / 2), 0.5, 0.5) (det::sigmoid_n_lockstep); gate c~: tanh(x) — one tanh per lane
    const bool cell_gate = gate == 3;
    const float t = mppi::det::tanh(pre * (cell_gate ? 1.0f : 0.5f));
    const float a = cell_gate ? t : mppi::det::fma(t, 0.5f, 0.5f);
    const float gi = __shfl(a, u, 64), gf = __shfl(a, H + u, 64), go = __shfl(a, 2 * H + u, 64), gc = __shfl(a, 3 * H + u, 64);
    const float in_part = gi * gc;
    const float keep_part = gf * c;
    c = in_part + keep_part;
    h = mppi::det::tanh(c) * go;
    /* ---- output MLP, layer 1: [h ; x] -> M, tanh ---- */
    acc = 0.0f;
#pragma unroll
    for (int k = 0; k < H; k++)
      acc = mppi::det::fma(w1h[k], lane_value(h, k), acc);
#pragma unroll
    for (int k = 0; k < I; k++)
      acc = mppi::det::fma(w1x[k], in[k], acc);
    const float a1 = mppi::det::tanh(acc + b1);
    /* ---- layer 2 (linear) ---- */
    // the four interleaved chains of FNNHelper::split_output_sum_ (what every form of this network evaluates)
    float c4[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int kb = 0; kb < M; kb += 16)  // (written so that every index is a constant after unrolling)
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (kb + 4 * g + i < M)
            c4[g] = mppi::det::fma(w2[kb + 4 * g + i], lane_value(a1, kb + 4 * g + i), c4[g]);
    acc = (c4[0] + c4[1]) + (c4[2] + c4[3]);
    acc = acc + b2;
#pragma unroll
    for (int i = 0; i < OUT; i++)
      out[i] = lane_value(acc, i);
  }
};
}  // namespace mppi
#endif
