/**
 * lstm_registers.hpp — a SMALL one-layer LSTM + two-layer output MLP of compile-time shape evaluated by one lane per
 * rollout with nothing but registers: the recurrent state (h, c) and every activation are VGPRs of the lane, the
 * parameters are read through the scalar unit (uniform addresses in the constant address space -> s_load_dwordx*, one
 * fetch per wave, the value is an SGPR operand of the lane's v_fma_f32).  No LDS, no barrier.
 *
 * Why next to LSTMHelper (lstm_helper.hpp, the reference's LDS contract): for the steering network of the RACER models
 * (I = 4, H = 4, MLP {8, 20, 1}: 308 multiply-adds) that contract costs far more than the arithmetic — run-time loop
 * bounds, four parameter reads from LDS per multiply-add, and the per-slot activation blocks sit 48 floats apart, i.e.
 * sixteen lanes on the same LDS bank (measured: 21 us per step and wave against 5 us for the whole elevation model).
 * Too small for the matrix cores as well (lstm_mfma.hpp tiles 16 hidden units per MFMA).
 *
 * Arithmetic contract: exactly LSTMHelper::forward's (same k-ordered fma chains, input part first, then the recurrent
 * part, then the bias; det:: activations; new cell state before the new hidden state; MLP on [h ; x]), so this form, the
 * LDS form and the CPU oracle agree bit for bit.  Blob layouts as in lstm_helper.hpp / fnn_helper.hpp.
 */
#ifndef MPPI_AMD_LSTM_REGISTERS_HPP_
#define MPPI_AMD_LSTM_REGISTERS_HPP_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mppi_amd/det_math.h"

namespace mppi
{
#if defined(__HIP_DEVICE_COMPILE__)
typedef const float __attribute__((address_space(4))) lstm_const_f32;
/** the same bytes seen through the constant address space: loads with a wave-uniform address become scalar loads */
__device__ inline lstm_const_f32* lstmScalarView(const float* p)
{
  return (lstm_const_f32*)(uintptr_t)p;
}
#endif

template <int I, int H, int L1, int OUT>
struct LSTMRegisters
{
  static constexpr int HH = H * H, HI = H * I;
  static constexpr int LSTM_NUM_PARAMS = 4 * HH + 4 * HI + 4 * H;  ///< h0, c0 follow
  static constexpr int FNN_NUM_PARAMS = L1 * (H + I) + L1 + OUT * L1 + OUT;
#ifndef MPPI_LSTM_REGISTERS_SGPR_BUDGET
#define MPPI_LSTM_REGISTERS_SGPR_BUDGET 64
#endif
  static constexpr int MLP_GROUP = (MPPI_LSTM_REGISTERS_SGPR_BUDGET / (H + I + 1)) > 0 ? (MPPI_LSTM_REGISTERS_SGPR_BUDGET / (H + I + 1)) : 1;

  /** (h, c) <- (h0, c0) of the blob */
  __device__ static inline void initialState(const float* lstm_blob, float (&h)[H], float (&c)[H])
  {
#if defined(__HIP_DEVICE_COMPILE__)
    lstm_const_f32* w = lstmScalarView(lstm_blob);
#pragma unroll
    for (int i = 0; i < H; i++)
    {
      h[i] = w[LSTM_NUM_PARAMS + i];
      c[i] = w[LSTM_NUM_PARAMS + H + i];
    }
#endif
  }

  __device__ static __forceinline__ void forward(const float* lstm_blob, const float* fnn_blob, const float (&x)[I], float (&h)[H],
                                        float (&c)[H], float (&out)[OUT])
  {
#if defined(__HIP_DEVICE_COMPILE__)
    lstm_const_f32* W_im = lstmScalarView(lstm_blob);
    lstm_const_f32* W_fm = W_im + HH;
    lstm_const_f32* W_om = W_fm + HH;
    lstm_const_f32* W_cm = W_om + HH;
    lstm_const_f32* W_ii = W_cm + HH;
    lstm_const_f32* W_fi = W_ii + HI;
    lstm_const_f32* W_oi = W_fi + HI;
    lstm_const_f32* W_ci = W_oi + HI;
    lstm_const_f32* b_i = W_ci + HI;
    lstm_const_f32* b_f = b_i + H;
    lstm_const_f32* b_o = b_f + H;
    lstm_const_f32* b_c = b_o + H;

    float sg[3 * H], gc[H];  // sigmoid arguments [i | f | o], tanh argument of the candidate
#pragma unroll
    for (int i = 0; i < H; i++)
    {
      float gi = 0.0f, gf = 0.0f, go = 0.0f, g = 0.0f;
#pragma unroll
      for (int j = 0; j < I; j++)
      {
        gi = mppi::det::fma(W_ii[i * I + j], x[j], gi);
        gf = mppi::det::fma(W_fi[i * I + j], x[j], gf);
        go = mppi::det::fma(W_oi[i * I + j], x[j], go);
        g = mppi::det::fma(W_ci[i * I + j], x[j], g);
      }
#pragma unroll
      for (int j = 0; j < H; j++)
      {
        gi = mppi::det::fma(W_im[i * H + j], h[j], gi);
        gf = mppi::det::fma(W_fm[i * H + j], h[j], gf);
        go = mppi::det::fma(W_om[i * H + j], h[j], go);
        g = mppi::det::fma(W_cm[i * H + j], h[j], g);
      }
      sg[i] = gi + b_i[i];
      sg[H + i] = gf + b_f[i];
      sg[2 * H + i] = go + b_o[i];
      gc[i] = g + b_c[i];
      // keeps the scalar loads of the next unit's weights below this point: left alone the compiler hoists every s_load of
      // the network to the top, hundreds of live SGPRs that spill into VGPRs and from there into scratch
      asm volatile("" ::: "memory");
    }
    mppi::det::sigmoid_n<3 * H>(sg);  // pairwise packed evaluation, same bits as det::sigmoid / det::tanh
    mppi::det::tanh_n<H>(gc);
    float tc[H];
#pragma unroll
    for (int i = 0; i < H; i++)
    {
      const float in_part = sg[i] * gc[i];
      const float keep_part = sg[H + i] * c[i];
      c[i] = in_part + keep_part;
      tc[i] = c[i];
    }
    mppi::det::tanh_n<H>(tc);
    float act[H + I];
#pragma unroll
    for (int i = 0; i < H; i++)
    {
      h[i] = tc[i] * sg[2 * H + i];
      act[i] = h[i];
    }
#pragma unroll
    for (int j = 0; j < I; j++)
      act[H + j] = x[j];

    lstm_const_f32* W1 = lstmScalarView(fnn_blob);
    lstm_const_f32* b1 = W1 + L1 * (H + I);
    lstm_const_f32* W2 = b1 + L1;
    lstm_const_f32* b2 = W2 + OUT * L1;
    float hid[L1];
#pragma unroll
    for (int j = 0; j < L1; j++)
    {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < H + I; k++)
        acc = mppi::det::fma(W1[j * (H + I) + k], act[k], acc);
      hid[j] = acc + b1[j];
      // a group of neurons per fence: their scalar loads are issued together and pay the scalar-memory latency once
      // (results of scalar loads return out of order, so every use waits for ALL loads in flight); the group is sized to
      // what the ~100 SGPRs of a wave hold next to the kernel's arguments
      if ((j + 1) % MLP_GROUP == 0)
        asm volatile("" ::: "memory");
    }
    mppi::det::tanh_n<L1>(hid);
#pragma unroll
    for (int j = 0; j < OUT; j++)
    {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < L1; k++)
        acc = mppi::det::fma(W2[j * L1 + k], hid[k], acc);
      out[j] = acc + b2[j];
      asm volatile("" ::: "memory");
    }
#endif
  }
};
}  // namespace mppi

#endif
