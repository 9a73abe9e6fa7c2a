/**
 * fnn_wave.hpp — the MLP forward pass of the NN dynamics for ONE rollout on one wave: lane j is neuron j.
 *
 * What it is for: the re-rollout of the optimised control sequence (computeStateTrajectoryHelper, controller.cuh:643-663 —
 * finalizeRepKernel here) is a single trajectory, T dependent steps of one wave.  The MFMA forward (fnn_mfma.hpp) is built
 * for 16 rollouts per wave; for one rollout its cost is the LENGTH of its dependent chains — 18 back-to-back
 * v_mfma_f32_16x16x4_f32 of 8 passes each, the packed tanh of 8 values per lane, two cross-lane transposes — 1.2 us per
 * AutoRally step.  Here every lane owns one neuron of a layer and walks its row of the weight matrix,
 *     acc = fma(W[j][k], act[k], acc),  k ascending, acc0 = 0;  then acc += b[j]
 * — literally the chain FNNHelper::forward and the CPU oracle evaluate (the output layer: their four interleaved chains,
 * FNNHelper::split_output_sum_), so the result is the same bits as the MFMA form's —
 * with act[k] read from lane k of the previous layer's register (v_readlane: a scalar operand of the fma).  One tanh per
 * lane and layer.  32 + 32 + 6 dependent fmas a step instead of 18 MFMAs: 0.5 us per AutoRally step.
 *
 * Restrictions: layers {IN, H, H, OUT} with H <= 64, OUT <= 64 (the weights of a lane: IN + 2 H registers).
 */
#ifndef MPPI_AMD_FNN_WAVE_HPP_
#define MPPI_AMD_FNN_WAVE_HPP_

#include <hip/hip_runtime.h>
#include "mppi_amd/det_math.h"

namespace mppi
{
template <int IN, int H, int OUT>
struct FNNWave
{
  static_assert(H <= 64 && OUT <= 64, "one lane per neuron");
  static constexpr int NUM_PARAMS = IN * H + H + H * H + H + H * OUT + OUT;
  float w1[IN], w2[H], w3[H];  ///< this lane's rows of W1, W2 (neuron lane) and W3 (output lane)
  float b1, b2, b3;

  /** theta: parameter blob [W1 (H x IN) | b1 | W2 (H x H) | b2 | W3 (OUT x H) | b3] (fnn_helper.cu:176-183) */
  __device__ inline void load(const float* __restrict__ theta, const int lane)
  {
    const float* W1 = theta;
    const float* B1 = W1 + IN * H;
    const float* W2 = B1 + H;
    const float* B2 = W2 + H * H;
    const float* W3 = B2 + H;
    const float* B3 = W3 + H * OUT;
    const int j = lane < H ? lane : 0;    // lanes beyond the layer repeat neuron 0: nobody reads them
    const int o = lane < OUT ? lane : 0;
#pragma unroll
    for (int k = 0; k < IN; k++)
      w1[k] = W1[j * IN + k];
#pragma unroll
    for (int k = 0; k < H; k++)
    {
      w2[k] = W2[j * H + k];
      w3[k] = W3[o * H + k];
    }
    b1 = B1[j];
    b2 = B2[j];
    b3 = B3[o];
  }

  __device__ static inline float lane_value(const float v, const int k)
  {
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), k));
  }

  /** in[IN]: the network input (the same in every lane); out[OUT]: the network output, the same in every lane */
  __device__ inline void forward(const float (&in)[IN], float (&out)[OUT]) const
  {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < IN; k++)
      acc = mppi::det::fma(w1[k], in[k], acc);
    const float h1 = mppi::det::tanh(acc + b1);
    acc = 0.0f;
#pragma unroll
    for (int k = 0; k < H; k++)
      acc = mppi::det::fma(w2[k], lane_value(h1, k), acc);
    const float h2 = mppi::det::tanh(acc + b2);
    // output layer: the four interleaved chains of FNNHelper::split_output_sum_ (what every form of this network evaluates)
    float c4[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int kb = 0; kb < H; kb += 16)  // (written so that every index is a constant after unrolling)
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (kb + 4 * g + i < H)
            c4[g] = mppi::det::fma(w3[kb + 4 * g + i], lane_value(h2, kb + 4 * g + i), c4[g]);
    acc = (c4[0] + c4[1]) + (c4[2] + c4[3]);
    acc = acc + b3;
#pragma unroll
    for (int i = 0; i < OUT; i++)
      out[i] = lane_value(acc, i);
  }
};
}  // namespace mppi
#endif
