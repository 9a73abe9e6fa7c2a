/**
 * fnn_helper.hpp — fully connected network (tanh hidden layers, linear output) evaluated inside the rollout.
 *
 * Device-side counterpart of the reference's FNNHelper (include/mppi/utils/nn_helpers/fnn_helper.cuh,
 * fnn_helper.cu:384-502): same parameter blob layout `[W1 (out x in, row-major) | b1 | W2 | b2 | ...]`
 * (fnn_helper.cu:176-183), same LDS contract — the block-shared part (Grd request) holds the parameters, each rollout
 * slot (Blk request) holds two activation buffers of LARGEST_LAYER floats (fnn_helper.cu:187-189, 487-502) — and the
 * same work split: output neurons strided over the threadIdx.y lanes of a rollout, a barrier after every layer.
 *
 * Arithmetic contract (what the CPU oracle restates): for neuron j of a layer
 *     acc = 0;  for k ascending: acc = fma(W[j][k], act[k], acc);  acc += b[j];  hidden layers: acc = det::tanh(acc)
 * i.e. a k-ordered fp32 fma chain — exactly what `v_mfma_f32_*_f32` computes per output element, so the MFMA variant of
 * the forward pass (rollouts of a wave as the N dimension) is bit-identical to this VALU form.
 * (reference device loop: `tmp += W[j*in + k] * curr_act[k]`, fnn_helper.cu:451-456, which nvcc contracts to FMAs.)
 */
#ifndef MPPI_AMD_FNN_HELPER_HPP_
#define MPPI_AMD_FNN_HELPER_HPP_

#include <hip/hip_runtime.h>
#include "mppi_amd/det_math.h"
#include "mppi_amd/plugin/parallel_utils.hpp"

namespace mppi
{
namespace nn
{
/** reference: utils/activation_functions.cuh:17-29 */
__host__ __device__ inline float tanh(float input)
{
  return mppi::det::tanh(input);
}
/** device flavour, reference: utils/activation_functions.cuh:49-59 */
__host__ __device__ inline float sigmoid(float input)
{
  return mppi::det::sigmoid(input);
}
}  // namespace nn

constexpr int FNN_MAX_LAYERS = 8;

class FNNHelper
{
public:
  int NUM_LAYERS = 0;
  int net_structure_[FNN_MAX_LAYERS] = { 0 };
  int stride_idcs_[2 * (FNN_MAX_LAYERS - 1)] = { 0 };  ///< offsets of W_i, b_i in the blob (fnn_helper.cu:295-309)
  int NUM_PARAMS = 0;
  int LARGEST_LAYER = 0;
  int INPUT_DIM = 0, OUTPUT_DIM = 0;
  const float* theta_d_ = nullptr;  ///< parameter blob in HBM (device pointer, owned by the engine)

  /** host: derive the blob layout from the layer sizes, e.g. {6, 32, 32, 4} */
  __host__ bool setStructure(const int* layers, int num_layers)
  {
    if (num_layers < 2 || num_layers > FNN_MAX_LAYERS)
      return false;
    NUM_LAYERS = num_layers;
    NUM_PARAMS = 0;
    LARGEST_LAYER = 0;
    for (int i = 0; i < num_layers; i++)
    {
      net_structure_[i] = layers[i];
      LARGEST_LAYER = layers[i] > LARGEST_LAYER ? layers[i] : LARGEST_LAYER;
    }
    for (int i = 0; i + 1 < num_layers; i++)
    {
      stride_idcs_[2 * i] = NUM_PARAMS;
      NUM_PARAMS += layers[i] * layers[i + 1];
      stride_idcs_[2 * i + 1] = NUM_PARAMS;
      NUM_PARAMS += layers[i + 1];
    }
    INPUT_DIM = layers[0];
    OUTPUT_DIM = layers[num_layers - 1];
    return true;
  }

  /**
   * Summation order of the OUTPUT layer's dot products.  false: the reference's single chain, k ascending
   * (fnn_helper.cu:458-462).  true — set by the two networks whose rollout forward runs on the matrix cores (AutoRally
   * 6-32-32-4: NeuralNetModel; the bicycle LSTM's output net {22, 32, 4}) so that EVERY form of those models (this LDS form,
   * FNNMfma / LSTMMfma, FNNWave / LSTMWave) and the CPU oracle evaluate the same bits: four interleaved chains, chain g =
   * the inputs k with (k >> 2) & 3 == g in ascending k, out = (c0 + c1) + (c2 + c3) + b.  In the MFMA forms that is the
   * layer evaluated where the previous layer's outputs already are (lane group g of the D layout owns exactly chain g's
   * inputs) — 16 packed fmas and two swap-and-add steps instead of 8 MFMAs that use 4 of their 16 rows, behind two 4x4
   * transposes (AutoRally-NN K=16384, T=150: 188.4 / 182.5 -> 179.0 / 174.2 us per launch, A/B in one session, round 5).
   * Only the ORDER of the additions differs from the reference; tests/test_fnn_output_order.py bounds the effect.
   */
  bool split_output_sum_ = false;

  /** block-shared LDS bytes: the parameter blob, padded to 16 B (reference: SHARED_MEM_REQUEST_GRD_BYTES) */
  __host__ __device__ int getGrdSharedSizeBytes() const
  {
    return ((NUM_PARAMS + 3) / 4) * 4 * (int)sizeof(float);
  }
  /** per-rollout LDS bytes: two activation buffers (reference: SHARED_MEM_REQUEST_BLK_BYTES) */
  __host__ __device__ int getBlkSharedSizeBytes() const
  {
    return 2 * ((LARGEST_LAYER + 3) / 4) * 4 * (int)sizeof(float);
  }

  /** reference: fnn_helper.cu:384-418 — all threads of the block copy the blob into LDS, then barrier */
  __device__ inline void initialize(float* theta_s) const
  {
    const int tid = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
    const int n = (int)(blockDim.x * blockDim.y * blockDim.z);
    for (int i = tid; i < NUM_PARAMS; i += n)
      theta_s[i] = theta_d_[i];
    __syncthreads();
  }

  /** the rollout slot's activation buffers (reference: fnn_helper.cu:487-502 getInputLocation) */
  __device__ inline float* getInputLocation(float* theta_s) const
  {
    const int slot = (int)(blockDim.x * threadIdx.z + threadIdx.x);
    return theta_s + getGrdSharedSizeBytes() / (int)sizeof(float) + slot * (getBlkSharedSizeBytes() / (int)sizeof(float));
  }

  /** reference: fnn_helper.cu:420-484; returns the buffer holding the output layer */
  __device__ inline float* forward(float* input, float* theta_s) const
  {
    return forward(input, theta_s, getInputLocation(theta_s));
  }

  /** the overload with an explicit activation buffer (reference: fnn_helper.cu:425-429, used by LSTMHelper::forward,
   *  lstm_helper.cu:462): curr_act holds the input layer, the second buffer follows it */
  __device__ inline float* forward(float* input, float* theta_s, float* curr_act) const
  {
    float* next_act = curr_act + getBlkSharedSizeBytes() / (2 * (int)sizeof(float));
    const int tdy = (int)__builtin_amdgcn_workitem_id_y();
    const int bdy = (int)__builtin_amdgcn_workgroup_size_y();
    if (input != nullptr)
    {
      for (int i = tdy; i < INPUT_DIM; i += bdy)
        curr_act[i] = input[i];
      mppi::lane_sync();
    }
    for (int i = 0; i < NUM_LAYERS - 1; i++)
    {
      const float* W = theta_s + stride_idcs_[2 * i];
      const float* b = theta_s + stride_idcs_[2 * i + 1];
      const int n_in = net_structure_[i], n_out = net_structure_[i + 1];
      const bool split = split_output_sum_ && i == NUM_LAYERS - 2;
      for (int j = tdy; j < n_out; j += bdy)
      {
        float tmp = 0.0f;
        if (split)
        {  // the matrix-core networks' output layer: four interleaved chains (see split_output_sum_)
          float c[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
          for (int k = 0; k < n_in; k++)
            c[(k >> 2) & 3] = mppi::det::fma(W[j * n_in + k], curr_act[k], c[(k >> 2) & 3]);
          tmp = (c[0] + c[1]) + (c[2] + c[3]);
        }
        else
        {
          for (int k = 0; k < n_in; k++)
            tmp = mppi::det::fma(W[j * n_in + k], curr_act[k], tmp);
        }
        tmp += b[j];
        if (i < NUM_LAYERS - 2)
          tmp = mppi::nn::tanh(tmp);
        next_act[j] = tmp;
      }
      float* t = curr_act;
      curr_act = next_act;
      next_act = t;
      mppi::lane_sync();
    }
    return curr_act;
  }
};
}  // namespace mppi
#endif
