/**
 * lstm_quad.hpp — a small LSTM (four hidden units) + two-layer output MLP evaluated by the FOUR replica lanes of a rollout
 * (REPLICATED_LANES = 4, lane = column + 16 * replica; include/mppi_amd/engine/rollout_kernel.hpp): replica r owns hidden
 * unit r (its four gates) and L1 / 4 neurons of the MLP's hidden layer; the new hidden state and the layer's activations
 * are exchanged with `__shfl` (ds_bpermute), the output layer's L1-term sums run on every replica with scalar-unit weights.
 * The weights a replica needs differ from lane to lane, so they cannot be scalar operands — see LSTMQuadRows below for where
 * they live.  An object of this type is a member of a Dynamics plugin that travels by value, i.e. every lane has its own
 * (its sixteenth of the replica's weights and the recurrent state: h of all four units, c of its own).
 *
 * Arithmetic per value as LSTMHelper / LSTMRegisters / the oracle: k-ordered fma chains (input part, then recurrent part,
 * then bias), det:: activations, new cell state before the new hidden state, MLP on [h ; x].
 */
#ifndef MPPI_AMD_LSTM_QUAD_HPP_
#define MPPI_AMD_LSTM_QUAD_HPP_

#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include "mppi_amd/det_math.h"
#include "mppi_amd/utils/nn_helpers/lstm_registers.hpp"

namespace mppi
{
/** compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}) */
template <class F, int... Is>
__device__ __forceinline__ void staticForImpl(F&& f, std::integer_sequence<int, Is...>)
{
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void staticFor(F&& f)
{
  staticForImpl(f, std::make_integer_sequence<int, N>{});
}

/**
 * Where the weights live.  A replica's sixteen lanes form one DPP row (lane = column + 16 * replica), and all sixteen need the
 * SAME weights: weight n of the replica is kept ONCE per row, in lane n % 16 of register n / 16, and a multiply-add fetches it
 * with the row-broadcast data-parallel primitive (`row_newbcast:n`, gfx90a+: every lane of a row reads the given lane of its
 * row; no LDS, no memory).  6 registers for the steering network (81 values per replica), 10 / 11 for the mean / uncertainty
 * networks of the complete RACER model (153 / 162).  The first version kept each lane's 81 steering weights in 81 of its own
 * registers: same speed for the LSTM-steering model, but 20 us slower per iteration for the suspension model and 35 / 245 us
 * (fused / role-pipelined kernel, whose lanes have 256 registers) for the complete model, where the registers are needed.
 */
template <int I, int L1, int OUT>
struct LSTMQuadRows
{
  static constexpr int H = 4, PER = L1 / 4;
  static_assert(L1 % 4 == 0, "the hidden layer of the output network is dealt out to four replicas");
  static constexpr int HH = H * H, HI = H * I, LSTM_NUM_PARAMS = 4 * HH + 4 * HI + 4 * H;
  static constexpr int FNN_NUM_PARAMS = L1 * (H + I) + L1 + OUT * L1 + OUT;
  // the replica's weights in the order they are used
  static constexpr int GATE0 = 0, BIAS0 = 4 * (I + H), W1_0 = BIAS0 + 4, B1_0 = W1_0 + PER * (H + I), NW = B1_0 + PER;
  static constexpr int NV = (NW + 15) / 16;

  float wv[NV];  ///< weight n: lane n % 16 of wv[n / 16]
  float h[H];    ///< hidden state of all four units
  float c;       ///< cell state of unit `replica`

  __device__ static inline float fromReplica(const float v, const int src)
  {
    return __shfl(v, (int)(threadIdx.x & 15) + 16 * src, 64);
  }
  /**
   * acc <- fma(weight N of this lane's replica, x, acc).  One instruction: `v_fmac_f32_dpp` with the row broadcast on its
   * first source.  Written as inline assembly because the compiler's DPP combiner leaves `v_mov_b32_dpp` + `v_fmac_f32`
   * (310 extra instructions per step of the complete RACER model); -DMPPI_LSTM_QUAD_ROWS_MOV restores that form.  The
   * hardware wants two wait states between a VALU write of a register and a DPP read of it, and the compiler's hazard
   * recogniser does not look into inline assembly: the weight registers are written once per rollout, but a register
   * allocator copy could land in front of a use — tools/dpp_hazard_lint.py checks the built library for that
   * (tests/test_abi.py runs it).
   */
  template <int N>
  __device__ __forceinline__ void fmaWeight(float& acc, const float x) const
  {
    static_assert(N >= 0 && N < NW, "weight index");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MPPI_LSTM_QUAD_ROWS_MOV)
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(wv[N / 16]), "v"(x), "n"(N % 16));
#else
    acc = mppi::det::fma(weight<N>(), x, acc);
#endif
  }
  /** weight N of this lane's replica */
  template <int N>
  __device__ __forceinline__ float weight() const
  {
    static_assert(N >= 0 && N < NW, "weight index");
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, wv[N / 16]), 0x150 + (N % 16), 0xF,
                                                                 0xF, true));
#else
    return 0.0f;
#endif
  }

  __device__ inline void load(const int rep, const float* __restrict__ lstm_blob, const float* __restrict__ fnn_blob)
  {
    const float* Wm = lstm_blob;
    const float* Wi = lstm_blob + 4 * HH;
    const float* B = Wi + 4 * HI;
    const float* B1 = fnn_blob + L1 * (H + I);
    const int col = (int)(threadIdx.x & 15);
#pragma unroll
    for (int v = 0; v < NV; v++)
    {
      const int n = 16 * v + col;
      float w = 0.0f;
      if (n < BIAS0)
      {
        const int gate = n / (I + H), j = n % (I + H);
        w = j < I ? Wi[gate * HI + rep * I + j] : Wm[gate * HH + rep * H + (j - I)];
      }
      else if (n < W1_0)
        w = B[(n - BIAS0) * H + rep];
      else if (n < B1_0)
      {
        const int i = (n - W1_0) / (H + I), k = (n - W1_0) % (H + I);
        w = fnn_blob[(PER * rep + i) * (H + I) + k];
      }
      else if (n < NW)
        w = B1[PER * rep + (n - B1_0)];
      wv[v] = w;
    }
#pragma unroll
    for (int j = 0; j < H; j++)
      h[j] = lstm_blob[LSTM_NUM_PARAMS + j];
    c = lstm_blob[LSTM_NUM_PARAMS + H + rep];
  }

  /** one forward pass; every replica ends with the same h and the same out */
  __device__ __forceinline__ void forward(const float* __restrict__ fnn_blob, const float (&input)[I], float (&out)[OUT])
  {
    float gate[4];
    staticFor<4>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      float acc = 0.0f;
      staticFor<I>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        fmaWeight<GATE0 + g * (I + H) + j>(acc, input[j]);
      });
      staticFor<H>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        fmaWeight<GATE0 + g * (I + H) + I + j>(acc, h[j]);
      });
      gate[g] = acc + weight<BIAS0 + g>();
    });
    float sg[3] = { gate[0], gate[1], gate[2] };
    mppi::det::sigmoid_n<3>(sg);
    const float gc = mppi::det::tanh(gate[3]);
    const float in_part = sg[0] * gc;
    const float keep_part = sg[1] * c;
    c = in_part + keep_part;
    const float h_own = mppi::det::tanh(c) * sg[2];
#pragma unroll
    for (int r = 0; r < 4; r++)
      h[r] = fromReplica(h_own, r);
    float act[H + I];
#pragma unroll
    for (int j = 0; j < H; j++)
      act[j] = h[j];
#pragma unroll
    for (int j = 0; j < I; j++)
      act[H + j] = input[j];
    float hid_own[PER];
    staticFor<PER>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      float acc = 0.0f;
      staticFor<H + I>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        fmaWeight<W1_0 + i * (H + I) + k>(acc, act[k]);
      });
      hid_own[i] = acc + weight<B1_0 + i>();
    });
    mppi::det::tanh_n<PER>(hid_own);
    float hid[L1];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
      for (int i = 0; i < PER; i++)
        hid[PER * s + i] = fromReplica(hid_own[i], s);
#if defined(__HIP_DEVICE_COMPILE__)
    lstm_const_f32* W2 = lstmScalarView(fnn_blob + L1 * (H + I) + L1);
#pragma unroll
    for (int j = 0; j < OUT; j++)
    {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < L1; k++)
        acc = mppi::det::fma(W2[j * L1 + k], hid[k], acc);
      out[j] = acc + W2[OUT * L1 + j];
      asm volatile("" ::: "memory");  // one output neuron's scalar loads at a time (lstm_registers.hpp)
    }
#endif
  }
};
}  // namespace mppi

#endif
