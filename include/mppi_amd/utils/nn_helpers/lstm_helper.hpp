/**
 * lstm_helper.hpp — one-layer LSTM + output MLP evaluated inside the rollout.
 *
 * Device-side counterpart of the reference's LSTMHelper<USE_SHARED = true>
 * (include/mppi/utils/nn_helpers/lstm_helper.cuh, lstm_helper.cu:197-241 initialize, :342-463 forward):
 *   - parameter blob  [W_im W_fm W_om W_cm (H x H each, row-major) | W_ii W_fi W_oi W_ci (H x I each) | b_i b_f b_o b_c |
 *                      h0 | c0]                                      (lstm_helper.cu:71-88; gate order i, f, o, c)
 *   - LDS contract    block-shared part: the 4HH + 4HI + 4H LSTM parameters, then the output network's parameters
 *                     (lstm_helper.cu:52-57); per rollout slot: [h (H) | c (H) | g_o / output activations ...]
 *                     = 2H floats + the output network's two activation buffers (lstm_helper.cu:55-57, 353-357)
 *   - work split      hidden units strided over the threadIdx.y lanes of a rollout; barriers between the phases
 *
 * Arithmetic contract (restated by the CPU oracle; the reference's `temp += W * x` loops are contracted to FMAs by nvcc):
 *   for unit i:  g = 0;  for j < I: g = fma(W_*i[i][j], x[j], g);  for j < H: g = fma(W_*m[i][j], h[j], g);  g += b_*[i]
 *   g_i, g_f, g_o = det::sigmoid(.)   (device flavour (1 + tanh(x/2)) / 2, utils/activation_functions.cuh:49-59)
 *   c~ = det::tanh(.);  c[i] <- g_i * c~ + g_f * c[i]   (two products, one add: no contraction)
 *   then, with the NEW cell state (lstm_helper.cu:448-452):  h[i] <- det::tanh(c[i]) * g_o[i]
 *   output = FNN([h ; x])                                  (lstm_helper.cu:455-462)
 * The same k-ordered fma chains are what the MFMA variant (lstm_mfma.hpp) evaluates.
 */
#ifndef MPPI_AMD_LSTM_HELPER_HPP_
#define MPPI_AMD_LSTM_HELPER_HPP_

#include <hip/hip_runtime.h>
#include "mppi_amd/det_math.h"
#include "mppi_amd/plugin/parallel_utils.hpp"
#include "mppi_amd/utils/nn_helpers/fnn_helper.hpp"

namespace mppi
{
class LSTMHelper
{
public:
  int INPUT_DIM = 0, HIDDEN_DIM = 0, OUTPUT_DIM = 0;
  int HIDDEN_HIDDEN_SIZE = 0, INPUT_HIDDEN_SIZE = 0;
  int LSTM_NUM_PARAMS = 0;   ///< 4HH + 4HI + 4H (without h0, c0)
  FNNHelper output_nn_;      ///< input layer = H + I (lstm_helper.cu:40)
  const float* weights_d_ = nullptr;  ///< LSTM blob in HBM, LSTM_NUM_PARAMS + 2H floats (owned by the engine)

  /** host: LSTM(input_dim, hidden_dim) followed by the output network `output_layers` (first entry = H + I) */
  __host__ bool setStructure(int input_dim, int hidden_dim, const int* output_layers, int num_output_layers)
  {
    if (input_dim <= 0 || hidden_dim <= 0 || !output_nn_.setStructure(output_layers, num_output_layers))
      return false;
    if (output_nn_.INPUT_DIM != input_dim + hidden_dim)
      return false;
    INPUT_DIM = input_dim;
    HIDDEN_DIM = hidden_dim;
    OUTPUT_DIM = output_nn_.OUTPUT_DIM;
    HIDDEN_HIDDEN_SIZE = hidden_dim * hidden_dim;
    INPUT_HIDDEN_SIZE = hidden_dim * input_dim;
    LSTM_NUM_PARAMS = 4 * HIDDEN_HIDDEN_SIZE + 4 * INPUT_HIDDEN_SIZE + 4 * hidden_dim;
    return true;
  }
  /** floats in the LSTM blob handed to the engine: parameters + initial hidden + initial cell (lstm_helper.cuh getNumParams) */
  __host__ __device__ int getNumParams() const
  {
    return LSTM_NUM_PARAMS + 2 * HIDDEN_DIM;
  }
  __host__ __device__ int getLSTMGrdSharedSizeBytes() const
  {
    return ((LSTM_NUM_PARAMS + 3) / 4) * 4 * (int)sizeof(float);
  }
  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return getLSTMGrdSharedSizeBytes() + output_nn_.getGrdSharedSizeBytes();
  }
  __host__ __device__ int getBlkSharedSizeBytes() const
  {
    return output_nn_.getBlkSharedSizeBytes() + ((2 * HIDDEN_DIM + 3) / 4) * 4 * (int)sizeof(float);
  }

  /** the rollout slot's [h | c | activations] block */
  __device__ inline float* getHiddenCellLocation(float* theta_s) const
  {
    const int slot = (int)(blockDim.x * threadIdx.z + threadIdx.x);
    return theta_s + getGrdSharedSizeBytes() / (int)sizeof(float) + slot * (getBlkSharedSizeBytes() / (int)sizeof(float));
  }
  /** where the caller writes the network input (reference: lstm_helper.cu:581-588, block + 3H) */
  __device__ inline float* getInputLocation(float* theta_s) const
  {
    return getHiddenCellLocation(theta_s) + ((2 * HIDDEN_DIM + 3) / 4) * 4 + HIDDEN_DIM;
  }

  /** reference: lstm_helper.cu:197-241 — parameters to LDS (all threads), then every slot's h, c <- h0, c0 */
  __device__ inline void initialize(float* theta_s) const
  {
    const int tid = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
    const int n = (int)(blockDim.x * blockDim.y * blockDim.z);
    for (int i = tid; i < LSTM_NUM_PARAMS; i += n)
      theta_s[i] = weights_d_[i];
    output_nn_.initialize(theta_s + getLSTMGrdSharedSizeBytes() / (int)sizeof(float));  // ends with a block barrier
    float* hc = getHiddenCellLocation(theta_s);
    for (int i = (int)threadIdx.y; i < HIDDEN_DIM; i += (int)blockDim.y)
    {
      hc[i] = weights_d_[LSTM_NUM_PARAMS + i];
      hc[HIDDEN_DIM + i] = weights_d_[LSTM_NUM_PARAMS + HIDDEN_DIM + i];
    }
    __syncthreads();
  }

  /** reference: lstm_helper.cu:342-463; `input` may be nullptr when the caller filled getInputLocation() itself.
   *  Returns the output network's output buffer. */
  __device__ inline float* forward(float* input, float* theta_s) const
  {
    const int H = HIDDEN_DIM, I = INPUT_DIM;
    const float* W_im = theta_s;
    const float* W_fm = W_im + HIDDEN_HIDDEN_SIZE;
    const float* W_om = W_fm + HIDDEN_HIDDEN_SIZE;
    const float* W_cm = W_om + HIDDEN_HIDDEN_SIZE;
    const float* W_ii = W_cm + HIDDEN_HIDDEN_SIZE;
    const float* W_fi = W_ii + INPUT_HIDDEN_SIZE;
    const float* W_oi = W_fi + INPUT_HIDDEN_SIZE;
    const float* W_ci = W_oi + INPUT_HIDDEN_SIZE;
    const float* b_i = W_ci + INPUT_HIDDEN_SIZE;
    const float* b_f = b_i + H;
    const float* b_o = b_f + H;
    const float* b_c = b_o + H;

    float* const h = getHiddenCellLocation(theta_s);
    float* const c = h + H;
    float* const g_o = h + ((2 * H + 3) / 4) * 4;  // output gate, then reused as the output network's input [h ; x]
    float* const x = g_o + H;

    const int tdy = (int)__builtin_amdgcn_workitem_id_y();
    const int bdy = (int)__builtin_amdgcn_workgroup_size_y();
    if (input != nullptr)
    {
      for (int i = tdy; i < I; i += bdy)
        x[i] = input[i];
      mppi::lane_sync();
    }
    for (int i = tdy; i < H; i += bdy)
    {
      float gi = 0.0f, gf = 0.0f, go = 0.0f, gc = 0.0f;
      for (int j = 0; j < I; j++)
      {
        const int index = i * I + j;
        gi = mppi::det::fma(W_ii[index], x[j], gi);
        gf = mppi::det::fma(W_fi[index], x[j], gf);
        go = mppi::det::fma(W_oi[index], x[j], go);
        gc = mppi::det::fma(W_ci[index], x[j], gc);
      }
      for (int j = 0; j < H; j++)
      {
        const int index = i * H + j;
        gi = mppi::det::fma(W_im[index], h[j], gi);
        gf = mppi::det::fma(W_fm[index], h[j], gf);
        go = mppi::det::fma(W_om[index], h[j], go);
        gc = mppi::det::fma(W_cm[index], h[j], gc);
      }
      gi += b_i[i];
      gf += b_f[i];
      go += b_o[i];
      gc += b_c[i];
      gi = mppi::nn::sigmoid(gi);
      gf = mppi::nn::sigmoid(gf);
      gc = mppi::nn::tanh(gc);
      g_o[i] = mppi::nn::sigmoid(go);
      const float in_part = gi * gc;
      const float keep_part = gf * c[i];
      c[i] = in_part + keep_part;
    }
    mppi::lane_sync();
    for (int i = tdy; i < H; i += bdy)
    {
      const float hn = mppi::nn::tanh(c[i]) * g_o[i];  // the NEW cell state, as the reference (lstm_helper.cu:450)
      h[i] = hn;
      g_o[i] = hn;  // output_act = [h ; x]
    }
    mppi::lane_sync();
    return output_nn_.forward(nullptr, theta_s + getLSTMGrdSharedSizeBytes() / (int)sizeof(float), g_o);
  }
};
}  // namespace mppi
#endif
