/**
 * lstm_mfma.hpp — LSTM step + output MLP of the NN dynamics on the CDNA4 matrix cores, recurrent state in registers.
 *
 * One wave advances the LSTM of 16 rollouts at a time with `v_mfma_f32_16x16x4_f32` (exact fp32):
 *     rows  m = the H = 16 hidden units of ONE gate (one row block per gate: i, f, o, c),
 *     cols  n = the 16 rollouts of the wave,
 *     k       = [x (I, zero-padded to a multiple of 4) ; h (16)], 4 per instruction — the reference's accumulation
 *               order, input part first, then the recurrent part (include/mppi/utils/nn_helpers/lstm_helper.cu:412-431).
 * Lane l is (n = l & 15, g = l >> 4): rollout n, k-group g — the role of the threadIdx.y lanes in the reference.
 *
 * Because every gate is exactly one row block, the D layout gives lane (n, g) the pre-activations of units 4g..4g+3 of
 * ALL FOUR gates of its rollout: the cell update c <- sigma(i) tanh(c~) + sigma(f) c and h <- tanh(c) sigma(o) is a purely
 * per-lane affair, and the cell state never leaves its 4 registers.  The new hidden state is re-laid out as next
 * step's B operand (unit 4s + g in k-step s) with the 4-instruction cross-lane transpose of wave_ops.hpp — no LDS, no
 * barrier anywhere on the recurrent path (the reference keeps h, c in shared memory and synchronises the block 3 times
 * per step, lstm_helper.cu:407, :445, :460).
 * Output MLP [h ; x] -> M (tanh) -> OUT on the same fragments (k order h first, then x: lstm_helper.cu:455-462).
 *
 * Numerics: MFMA chains are the k-ordered fp32 fma chains of lstm_helper.hpp / the CPU oracle; zero padding adds
 * fma(0, 0, acc) = acc.  Bit-identical to the LDS variant.
 *
 * Restrictions: H == 16, I <= 8, M a multiple of 16, OUT <= 4.
 */
#ifndef MPPI_AMD_LSTM_MFMA_HPP_
#define MPPI_AMD_LSTM_MFMA_HPP_

#include <hip/hip_runtime.h>
#include "mppi_amd/det_math.h"
#include "mppi_amd/utils/wave_ops.hpp"
#include "mppi_amd/utils/nn_helpers/fnn_mfma.hpp"

namespace mppi
{
template <int I, int H, int M, int OUT>
struct LSTMMfma
{
  static_assert(H == 16 && I <= 8 && M % 16 == 0 && OUT <= 4, "unsupported LSTM shape for the MFMA forward");
  static constexpr int KS_X = (I + 3) / 4;  ///< k-steps of the input part
  static constexpr int KS_H = H / 4;        ///< k-steps of the recurrent part
  static constexpr int RB_M = M / 16;       ///< row blocks of the MLP's hidden layer
  static constexpr int KS_M = M / 4;
  static constexpr int LSTM_NUM_PARAMS = 4 * H * H + 4 * H * I + 4 * H;
  static constexpr int LSTM_BLOB = LSTM_NUM_PARAMS + 2 * H;
  static constexpr int FNN_BLOB = (H + I) * M + M + M * OUT + OUT;

  /* per-lane constants */
  float ax[4][KS_X];   ///< gate weights, input part     (A fragments)
  float ah[4][KS_H];   ///< gate weights, recurrent part
  float bg[4][4];      ///< gate biases of the units this lane owns in the D layout
  float a1h[RB_M][KS_H], a1x[RB_M][KS_X], b1[RB_M][4];  ///< MLP layer 1 ([h ; x] -> M)
  float b2[4];
  /** MLP layer 2 (M -> OUT) on the vector unit: W2[o][16 rb + 4 g + i] of the units lane group g owns, at
   *  w2_lds[(4 g + o) * RB_M * 4 + 4 rb + i] in block-shared LDS and read back with 16-byte broadcast loads.  Measured, one
   *  session (K = 65536, T = 200): weights in 32 VGPRs per lane 1437 us per launch, in LDS 1359 us — the opposite of the AutoRally
   *  MLP (fnn_mfma.hpp: registers 179 us, LDS 188 us), so each network keeps the form that is faster for it. */
  const float* w2_lds = nullptr;
  static constexpr int W2_LDS_FLOATS = 4 * 4 * RB_M * 4;
  float a2m[KS_M];  ///< the same layer as chain-masked MFMA rows for register-starved kernels (fnn_mfma.hpp: a3m)
  /* recurrent state */
  float hb[KS_H];  ///< hidden state as B fragments: unit 4s + g
  float c[4];      ///< cell state of units 4g + i

  /** lstm: [W_im W_fm W_om W_cm | W_ii W_fi W_oi W_ci | b_i b_f b_o b_c | h0 | c0];  fnn: [W1 | b1 | W2 | b2] */
  __device__ inline void load(const float* __restrict__ lstm, const float* __restrict__ fnn, const int lane,
                              float* __restrict__ w2_table_lds)
  {
    w2_lds = w2_table_lds;
    const int m = lane & 15, g = lane >> 4;
    const float* Wm = lstm;
    const float* Wi = lstm + 4 * H * H;
    const float* B = Wi + 4 * H * I;
    const float* h0 = lstm + LSTM_NUM_PARAMS;
    const float* c0 = h0 + H;
#pragma unroll
    for (int gate = 0; gate < 4; gate++)
    {
#pragma unroll
      for (int s = 0; s < KS_X; s++)
      {
        const int k = 4 * s + g;
        ax[gate][s] = (k < I) ? Wi[gate * H * I + m * I + k] : 0.0f;
      }
#pragma unroll
      for (int s = 0; s < KS_H; s++)
        ah[gate][s] = Wm[gate * H * H + m * H + 4 * s + g];
#pragma unroll
      for (int i = 0; i < 4; i++)
        bg[gate][i] = B[gate * H + 4 * g + i];
    }
    const float* W1 = fnn;
    const float* B1 = W1 + (H + I) * M;
    const float* W2 = B1 + M;
    const float* B2 = W2 + M * OUT;
#pragma unroll
    for (int rb = 0; rb < RB_M; rb++)
    {
#pragma unroll
      for (int s = 0; s < KS_H; s++)
        a1h[rb][s] = W1[(16 * rb + m) * (H + I) + 4 * s + g];
#pragma unroll
      for (int s = 0; s < KS_X; s++)
      {
        const int k = 4 * s + g;
        a1x[rb][s] = (k < I) ? W1[(16 * rb + m) * (H + I) + H + k] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
        b1[rb][i] = B1[16 * rb + 4 * g + i];
    }
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
      for (int rb = 0; rb < RB_M; rb++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          w2_table_lds[(4 * g + o) * RB_M * 4 + 4 * rb + i] = (o < OUT) ? W2[o * M + 16 * rb + 4 * g + i] : 0.0f;  // (every lane
          // of group g and every wave of the block: same words, same values; visible after the block barrier behind initializeDynamics)
#pragma unroll
    for (int s = 0; s < KS_M; s++)
      a2m[s] = ((s & 3) == (m >> 2) && (m & 3) < OUT) ? W2[(m & 3) * M + 16 * (s >> 2) + 4 * (s & 3) + g] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++)
      b2[i] = (i < OUT) ? B2[i] : 0.0f;
#pragma unroll
    for (int s = 0; s < KS_H; s++)
      hb[s] = h0[4 * s + g];
#pragma unroll
    for (int i = 0; i < 4; i++)
      c[i] = c0[4 * g + i];
  }

  /** in[I]: the network input of this lane's rollout (identical in its 4 lanes); out[OUT]: identical in the 4 lanes */
  __device__ inline void forward(const float (&in)[I], float (&out)[OUT], const int lane)
  {
    const int g = lane >> 4;
    float bx[KS_X];
#pragma unroll
    for (int s = 0; s < KS_X; s++)
    {
      float b = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (4 * s + q < I)
          b = (g == q) ? in[4 * s + q] : b;
      bx[s] = b;
    }
    /* ---- gates: 4 independent MFMA chains (one per gate) ---- */
    mfma_f32x4 acc[4];
#pragma unroll
    for (int gate = 0; gate < 4; gate++)
      acc[gate] = mfma_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int s = 0; s < KS_X; s++)
#pragma unroll
      for (int gate = 0; gate < 4; gate++)
        acc[gate] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[gate][s], bx[s], acc[gate], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS_H; s++)
#pragma unroll
      for (int gate = 0; gate < 4; gate++)
        acc[gate] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[gate][s], hb[s], acc[gate], 0, 0, 0);
    /* ---- cell / hidden update of units 4g + i, per lane; the 12 sigmoids and 8 tanh go pairwise through the packed
     *      det::tanh2 ---- */
    float sg[12], gc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
      sg[i] = acc[0][i] + bg[0][i];
      sg[4 + i] = acc[1][i] + bg[1][i];
      sg[8 + i] = acc[2][i] + bg[2][i];
      gc[i] = acc[3][i] + bg[3][i];
    }
    // lockstep evaluation (det_math.h: tanh_pairs): no dependent back-to-back packed instructions, hence none of the
    // s_nop 0 a lone wave pays an issue slot for (LSTM K=65536, T=200: 1653 -> 1531 us per launch, with the colored-noise
    // sampler 2006 -> 1882 us; A/B in one session)
    mppi::det::sigmoid_n_lockstep<12>(sg);
    mppi::det::tanh_n_lockstep<4>(gc);
    float hn[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
      const float in_part = sg[i] * gc[i];
      const float keep_part = sg[4 + i] * c[i];
      c[i] = in_part + keep_part;
      hn[i] = c[i];
    }
    mppi::det::tanh_n_lockstep<4>(hn);
#pragma unroll
    for (int i = 0; i < 4; i++)
      hn[i] = hn[i] * sg[8 + i];
    mppi::wave::transpose4x4(hn);
#pragma unroll
    for (int s = 0; s < KS_H; s++)
      hb[s] = hn[s];
    /* ---- output MLP, layer 1: [h ; x] -> M, tanh ---- */
    float act[RB_M][4];
    {
      mfma_f32x4 a[RB_M];
#pragma unroll
      for (int rb = 0; rb < RB_M; rb++)
        a[rb] = mfma_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
      for (int s = 0; s < KS_H; s++)
#pragma unroll
        for (int rb = 0; rb < RB_M; rb++)
          a[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1h[rb][s], hb[s], a[rb], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < KS_X; s++)
#pragma unroll
        for (int rb = 0; rb < RB_M; rb++)
          a[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1x[rb][s], bx[s], a[rb], 0, 0, 0);
      float v[RB_M * 4];
#pragma unroll
      for (int rb = 0; rb < RB_M; rb++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          v[4 * rb + i] = a[rb][i] + b1[rb][i];
      mppi::det::tanh_n_lockstep<RB_M * 4>(v);
#pragma unroll
      for (int rb = 0; rb < RB_M; rb++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          act[rb][i] = v[4 * rb + i];  // stays in the D layout: units 16 rb + 4 g + i
    }
    if (__builtin_amdgcn_workgroup_size_x() > 512)
    {  // blocks of more than 512 threads (128 VGPRs per wave): the chain-masked MFMA form, same bits (fnn_mfma.hpp: a3m)
      mfma_f32x4 o = mfma_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
      for (int rb = 0; rb < RB_M; rb++)
      {
        mppi::wave::transpose4x4(act[rb]);  // units 16 rb + 4 s + g, s = 0..3
#pragma unroll
        for (int s = 0; s < 4; s++)
          o = __builtin_amdgcn_mfma_f32_16x16x4f32(a2m[4 * rb + s], act[rb][s], o, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < OUT; i++)
        out[i] = mppi::wave::sumOverLaneGroups(o[i]) + b2[i];
      return;
    }
    /* ---- layer 2 (linear) on the vector unit, in the D layout (fnn_mfma.hpp, layer 3: the same step): lane group g owns
     * the inputs of chain g of FNNHelper::split_output_sum_; the four chains meet in two swap-and-add steps ---- */
    float p[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    const mfma_f32x4* __restrict__ w2q = reinterpret_cast<const mfma_f32x4*>(w2_lds + 4 * g * RB_M * 4);
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
      for (int rb = 0; rb < RB_M; rb++)
      {
        const mfma_f32x4 w = w2q[o * RB_M + rb];  // one 16-byte LDS broadcast per lane group
#pragma unroll
        for (int i = 0; i < 4; i++)
          p[o] = mppi::det::fma(w[i], act[rb][i], p[o]);
      }
#pragma unroll
    for (int o = 0; o < 4; o++)
      p[o] = mppi::wave::sumOverLaneGroups(p[o]);
#pragma unroll
    for (int i = 0; i < OUT; i++)
      out[i] = p[i] + b2[i];
  }
};
}  // namespace mppi
#endif
