/**
 * fnn_mfma.hpp — the MLP forward pass of the NN dynamics on the CDNA4 matrix cores (fp32-input MFMA).
 *
 * One wave evaluates the network for 16 rollouts at a time with `v_mfma_f32_16x16x4_f32`
 * (D[16x16] += A[16x4] * B[4x16], exact fp32, 32 cycles):
 *     rows  m = output neurons of the layer (16 per row block),
 *     cols  n = the 16 rollouts of the wave,
 *     k       = input neurons, 4 per instruction.
 * Lane l of the wave is (n = l & 15, g = l >> 4): it belongs to rollout n and is its k-group g — exactly the role the
 * reference gives the threadIdx.y lanes of a rollout (block shape (16, 4): "x = rollout, y = intra-rollout lane"), only
 * that the lanes now cooperate through the matrix core instead of LDS + block barriers
 * (reference: FNNHelper::forward, include/mppi/utils/nn_helpers/fnn_helper.cu:420-484).
 *
 * Fragment layouts (cdna_hip_programming.md §3): A: lane holds A[m = l & 15][k = l >> 4]; B: lane holds
 * B[k = l >> 4][n = l & 15]; D: lane holds rows 4*(l >> 4) + i, i = 0..3, of column l & 15.
 *  - The weights are A fragments and stay in VGPRs for the whole kernel (28 registers for 6-32-32-4).
 *  - A layer's output (D layout) is biased and squashed in place — 8 values per lane, i.e. the 512 tanh of a 32-neuron
 *    layer x 16 rollouts are spread evenly over the 64 lanes, evaluated pairwise as packed fp32 (det::tanh2) — and
 *    re-laid out as the next layer's B fragments with the cross-lane 4x4 transpose of wave_ops.hpp (v_permlane32_swap +
 *    v_permlane16_swap): no LDS traffic and no barrier anywhere in the forward pass.
 *  - The last (linear) layer does not go through the matrix core: 4 outputs would use 4 of an MFMA's 16 rows.  Every lane
 *    multiplies the 8 hidden values it owns in the D layout with its slices of W3 (16 packed fmas for the 4 outputs) and the
 *    four lane groups' partial chains meet in two swap-and-add steps (wave_ops.hpp: sumOverLaneGroups) — the summation
 *    order FNNHelper::split_output_sum_ defines for this network in every form and in the oracle.
 *
 * Numerics: a chain of MFMAs over the k-steps is bit-for-bit the k-ordered fp32 fma chain
 *     acc = fma(W[j][k], act[k], acc), k ascending, acc0 = 0;  then acc += b[j]
 * that FNNHelper::forward (fnn_helper.hpp) and the CPU oracle evaluate; zero padding of k adds fma(0, 0, acc) = acc.
 * Output layer: four interleaved chains + (c0 + c1) + (c2 + c3), the same in all of them (split_output_sum_).
 *
 * Restrictions: layers {IN, H, H, OUT} with IN <= 8, H a multiple of 16, OUT <= 4 (the AutoRally 6-32-32-4 network).
 */
#ifndef MPPI_AMD_FNN_MFMA_HPP_
#define MPPI_AMD_FNN_MFMA_HPP_

#include <hip/hip_runtime.h>
#include "mppi_amd/det_math.h"
#include "mppi_amd/utils/wave_ops.hpp"

namespace mppi
{
typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));

/* A/B experiment hooks (tools/, never defined in a product build): what the step loop costs without its MFMAs / tanh */
#if defined(MPPI_KNOCKOUT_MFMA)
__device__ inline mfma_f32x4 mfma16x16x4(float a, float b, mfma_f32x4 c)
{
  typedef float f32x2 __attribute__((ext_vector_type(2)));  // two packed fma instead of the MFMA: same data flow
  const f32x2 lo = __builtin_elementwise_fma(f32x2{ a, b }, f32x2{ b, a }, f32x2{ c[0], c[1] });
  const f32x2 hi = __builtin_elementwise_fma(f32x2{ b, a }, f32x2{ a, a }, f32x2{ c[2], c[3] });
  return mfma_f32x4{ lo.x, lo.y, hi.x, hi.y };
}
#else
__device__ inline mfma_f32x4 mfma16x16x4(float a, float b, mfma_f32x4 c)
{
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
#endif

template <int IN, int H, int OUT>
struct FNNMfma
{
  static_assert(IN <= 8 && H % 16 == 0 && OUT <= 4, "unsupported network shape for the MFMA forward");
  static constexpr int RB = H / 16;        ///< row blocks of a hidden layer
  static constexpr int KS_IN = (IN + 3) / 4;
  static constexpr int KS_H = H / 4;
  static constexpr int NUM_PARAMS = IN * H + H + H * H + H + H * OUT + OUT;

  /* per-lane constants: weight fragments (A operands) and the biases of the rows this lane owns in the D layout */
  float a1[RB][KS_IN];
  float a2[RB][KS_H];
  float w3s[4][RB * 4];  ///< W3[o][16 rb + 4 g + i]: the output layer's weights of the units this lane owns in the D layout
  /** the same layer as ONE chain of MFMAs for kernels that cannot afford w3s' 32 registers (blocks of more than 512 threads:
   *  128 VGPRs per wave — the Robust MPPI pipeline spilled inside its step loop with them, 388 -> 551 us per launch): row
   *  m = 4 chain + output, and A[m][k] is W3[output][k] only where input k belongs to that chain, else 0 — fma(0, x, acc) = acc,
   *  so row m accumulates exactly chain (m >> 2) of output (m & 3) in ascending k, all 16 rows busy; the D layout then hands
   *  lane group g the chains g of the four outputs, and the same two swap-and-add steps join them.  Same bits, 8 registers. */
  float a3m[KS_H];
  /** register-starved kernels (outputLayerOnMatrixCore()) also fetch the hidden layers' biases — 16 values that depend on the
   *  lane group only — from block-shared LDS at the point of use instead of holding them in VGPRs: bias_lds[((layer * 4 + g) *
   *  RB + rb) * 4 + i]; with them in registers the Robust MPPI pipeline kernel spilled 6 VGPRs inside its step loop, and the
   *  scratch stores were most of the "write amplification" its HBM counters showed (74 MB written for 39 MB of rows) */
  const float* bias_lds = nullptr;
  static constexpr int BIAS_LDS_FLOATS = 2 * 4 * RB * 4;
  float b1[RB][4];
  float b2[RB][4];
  float b3[4];

  /** theta: parameter blob [W1 (H x IN) | b1 | W2 (H x H) | b2 | W3 (OUT x H) | b3] (fnn_helper.cu:176-183) */
  __device__ inline void load(const float* __restrict__ theta, const int lane, float* __restrict__ bias_table_lds)
  {
    const int m = lane & 15, g = lane >> 4;
    bias_lds = bias_table_lds;
    const float* W1 = theta;
    const float* B1 = W1 + IN * H;
    const float* W2 = B1 + H;
    const float* B2 = W2 + H * H;
    const float* W3 = B2 + H;
    const float* B3 = W3 + H * OUT;
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
    {
#pragma unroll
      for (int s = 0; s < KS_IN; s++)
      {
        const int k = 4 * s + g;
        a1[rb][s] = (k < IN) ? W1[(16 * rb + m) * IN + k] : 0.0f;
      }
#pragma unroll
      for (int s = 0; s < KS_H; s++)
        a2[rb][s] = W2[(16 * rb + m) * H + 4 * s + g];
#pragma unroll
      for (int i = 0; i < 4; i++)
      {
        b1[rb][i] = B1[16 * rb + 4 * g + i];
        b2[rb][i] = B2[16 * rb + 4 * g + i];
        // (every lane of group g, every wave: same words, same values; visible after the barrier behind initializeDynamics)
        bias_table_lds[((0 * 4 + g) * RB + rb) * 4 + i] = b1[rb][i];
        bias_table_lds[((1 * 4 + g) * RB + rb) * 4 + i] = b2[rb][i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
      b3[i] = (i < OUT) ? B3[i] : 0.0f;
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
      for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          w3s[o][4 * rb + i] = (o < OUT) ? W3[o * H + 16 * rb + 4 * g + i] : 0.0f;
    // k-step s of the next layer's B fragments carries unit 16 (s >> 2) + 4 (s & 3) + g, which belongs to chain s & 3
#pragma unroll
    for (int s = 0; s < KS_H; s++)
      a3m[s] = ((s & 3) == (m >> 2) && (m & 3) < OUT) ? W3[(m & 3) * H + 16 * (s >> 2) + 4 * (s & 3) + g] : 0.0f;
  }
  /** which form of the output layer this kernel runs (folds at compile time: every kernel asserts its block shape) */
  __device__ static inline bool outputLayerOnMatrixCore()
  {
    return __builtin_amdgcn_workgroup_size_x() > 512;
  }

  /** hidden layer epilogue: bias + tanh (pairwise packed, det::tanh_n) of the RB x 4 values this lane owns, then the
   *  4x4 cross-lane transpose that turns each row block's D layout (units 16 rb + 4 g + i) into the next layer's B
   *  fragments (unit 16 rb + 4 s + g in k-step 4 rb + s) */
  __device__ inline void squash(const mfma_f32x4 (&acc)[RB], const float (&bias)[RB][4], float (&b_next)[KS_H],
                                const int layer, const int g) const
  {
    float v[RB * 4];
    if (outputLayerOnMatrixCore())
    {
      const mfma_f32x4* __restrict__ bq = reinterpret_cast<const mfma_f32x4*>(bias_lds + (layer * 4 + g) * RB * 4);
#pragma unroll
      for (int rb = 0; rb < RB; rb++)
      {
        const mfma_f32x4 b = bq[rb];
#pragma unroll
        for (int i = 0; i < 4; i++)
          v[4 * rb + i] = acc[rb][i] + b[i];
      }
    }
    else
    {
#pragma unroll
      for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          v[4 * rb + i] = acc[rb][i] + bias[rb][i];
    }
#if !defined(MPPI_KNOCKOUT_TANH)
    // lockstep over the four pairs: no packed instruction reads its predecessor's result, i.e. none of the s_nop 0 the
    // compiler puts behind dependent packed fp32 instructions — a lone wave pays an issue slot for each
    // (AutoRally-NN K=16384, T=150: 201.5 -> 193.4 us per launch, A/B in one session)
    mppi::det::tanh_n_lockstep<RB * 4>(v);
#endif
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
    {
      float t[4] = { v[4 * rb], v[4 * rb + 1], v[4 * rb + 2], v[4 * rb + 3] };
      mppi::wave::transpose4x4(t);
#pragma unroll
      for (int sidx = 0; sidx < 4; sidx++)
        b_next[4 * rb + sidx] = t[sidx];
    }
  }

  /** All eight B fragments of a layer pass through one empty asm statement: a data dependence the compiler cannot see
   *  through, so no MFMA of the layer is issued before the whole squash in front of it is done and the layer's MFMAs end up
   *  back to back.  Left alone, the scheduler spreads them between the tanh stages "to hide their latency" — but a wave alone
   *  on its SIMD does not issue VALU work under its own MFMA, and every MFMA <-> VALU switch costs it ~6 ns
   *  (tools/ubench/mfma_overlap.hip). */
  __device__ static inline void gather8(float (&b)[KS_H])
  {
    static_assert(KS_H == 8, "gather8 is written for H = 32");
#if !defined(MPPI_FNN_NO_CLUSTER)
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
#endif
  }

  /**
   * in[IN]: the network input of this lane's rollout (every lane of the rollout holds the same values);
   * out[OUT]: the network output, identical in the 4 lanes of the rollout.  No LDS, no barrier.
   *
   * MFMAs and the VALU work between them are left in the compiler's order, MFMAs mostly back to back: a wave that is alone
   * on its SIMD does NOT issue VALU instructions under its own MFMA (tools/ubench/mfma_overlap.hip: {MFMA + N independent
   * v_fma} costs 19.6 + 1.85 N ns against 13.3 ns for the MFMA alone — the times add, plus ~6 ns per MFMA <-> VALU switch), so
   * a hand-interleaved forward (row block 1's MFMAs between the tanh stages of row block 0) measured the same 227 us per
   * AutoRally launch.  Only ANOTHER wave of the SIMD fills the matrix pipe's shadow — the helper waves of the pipelined
   * kernel do.
   */
  __device__ inline void forward(const float (&in)[IN], float (&out)[OUT], const int lane) const
  {
    const int g = lane >> 4;
    /* ---- layer 1: B fragment of k-step s = in[4s + g] ---- */
    float bin[KS_IN];
#pragma unroll
    for (int s = 0; s < KS_IN; s++)
    {
      float b = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (4 * s + q < IN)
          b = (g == q) ? in[4 * s + q] : b;
      bin[s] = b;
    }
    mfma_f32x4 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
      acc[rb] = mfma_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    // the row blocks' chains are independent: interleave them so that consecutive MFMAs do not wait on each other
#pragma unroll
    for (int s = 0; s < KS_IN; s++)
#pragma unroll
      for (int rb = 0; rb < RB; rb++)
        acc[rb] = mfma16x16x4(a1[rb][s], bin[s], acc[rb]);
    float bh[KS_H];
    squash(acc, b1, bh, 0, g);
    gather8(bh);
    /* ---- layer 2 ---- */
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
      acc[rb] = mfma_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int s = 0; s < KS_H; s++)
#pragma unroll
      for (int rb = 0; rb < RB; rb++)
        acc[rb] = mfma16x16x4(a2[rb][s], bh[s], acc[rb]);
    if (outputLayerOnMatrixCore())
    {
      float bo[KS_H];
      squash(acc, b2, bo, 1, g);
      gather8(bo);
      mfma_f32x4 o = mfma_f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
      for (int s = 0; s < KS_H; s++)
        o = mfma16x16x4(a3m[s], bo[s], o);
#pragma unroll
      for (int i = 0; i < OUT; i++)
        out[i] = mppi::wave::sumOverLaneGroups(o[i]) + b3[i];
      return;
    }
    /* ---- layer 3 (linear) on the vector unit, in the D layout: no transpose, no matrix-core rows that compute nothing.
     * Lane group g owns the units 16 rb + 4 g + i — exactly the inputs of chain g of FNNHelper::split_output_sum_, in ascending
     * order; the chains of the four lane groups meet through v_permlane16_swap (rows 0|1, 2|3) and v_permlane32_swap (halves):
     * both lanes of a pair add the same two values in the same order, so all four lanes of a rollout hold the same bits. ---- */
    float v[RB * 4];
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
      for (int i = 0; i < 4; i++)
        v[4 * rb + i] = acc[rb][i] + b2[rb][i];
#if !defined(MPPI_KNOCKOUT_TANH)
    mppi::det::tanh_n_lockstep<RB * 4>(v);
#endif
    float p[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int j = 0; j < RB * 4; j++)
#pragma unroll
      for (int o = 0; o < 4; o++)
        p[o] = mppi::det::fma(w3s[o][j], v[j], p[o]);
#pragma unroll
    for (int o = 0; o < 4; o++)
      p[o] = mppi::wave::sumOverLaneGroups(p[o]);
#pragma unroll
    for (int i = 0; i < OUT; i++)
      out[i] = p[i] + b3[i];
  }
};
}  // namespace mppi
#endif
