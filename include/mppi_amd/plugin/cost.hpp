/**
 * cost.hpp — CRTP base class of every Cost plugin (reference: include/mppi/cost_functions/cost.cuh:17-233, cost.cu:39-53).
 * Device contract: initializeCosts, computeRunningCost (= computeStateCost + computeControlCost on lane y == 0),
 * terminalCost.  Same names and argument meaning as the reference.
 */
#ifndef MPPI_AMD_PLUGIN_COST_HPP_
#define MPPI_AMD_PLUGIN_COST_HPP_

#include <hip/hip_runtime.h>
#include "mppi_amd/plugin/dynamics.hpp"

/** reference: cost_functions/cost.cuh:17-31 */
template <int C_DIM>
struct CostParams
{
  static const int CONTROL_DIM = C_DIM;
  float control_cost_coeff[C_DIM];
  float discount = 1.0f;
  CostParams()
  {
    for (int i = 0; i < C_DIM; ++i)
    {
      control_cost_coeff[i] = 1.0f;
    }
  }
};

template <class CLASS_T, class PARAMS_T, class DYN_PARAMS_T = DynamicsParams>
class Cost : public mppi::Managed
{
public:
  using ControlIndex = typename DYN_PARAMS_T::ControlIndex;
  using OutputIndex = typename DYN_PARAMS_T::OutputIndex;
  using TEMPLATED_DYN_PARAMS = DYN_PARAMS_T;
  static const int CONTROL_DIM = E_INDEX(ControlIndex, NUM_CONTROLS);
  static const int OUTPUT_DIM = E_INDEX(OutputIndex, NUM_OUTPUTS);
  typedef CLASS_T COST_T;
  typedef PARAMS_T COST_PARAMS_T;

  hipError_t GPUSetup()
  {
    CLASS_T* derived = static_cast<CLASS_T*>(this);
    if (!GPUMemStatus_)
    {
      hipError_t e = Managed::GPUSetup(derived, &cost_d_);
      if (e != hipSuccess)
        return e;
    }
    return derived->paramsToDevice();
  }
  hipError_t freeCudaMem()
  {
    hipError_t e = hipSuccess;
    if (GPUMemStatus_)
    {
      e = hipFree(cost_d_);
      GPUMemStatus_ = false;
      cost_d_ = nullptr;
    }
    return e;
  }
  hipError_t paramsToDevice()
  {
    if (!GPUMemStatus_)
      return hipSuccess;
    hipError_t e = hipMemcpyAsync(&cost_d_->params_, &params_, sizeof(PARAMS_T), hipMemcpyHostToDevice, stream_);
    if (e == hipSuccess)
      e = hipStreamSynchronize(stream_);
    return e;
  }
  void setParams(const PARAMS_T& params)
  {
    params_ = params;
    (void)paramsToDevice();
  }
  __host__ __device__ PARAMS_T getParams() const
  {
    return params_;
  }

  /** reference: cost.cuh:176-178 */
  __device__ inline void initializeCosts(float* output, float* control, float* theta_c, float t_0, float dt)
  {
  }
  /** reference: cost.cuh:205-208 — the control cost lives in the sampler's likelihood-ratio term */
  __device__ inline float computeControlCost(float* u, int timestep, float* theta_c, int* crash)
  {
    return 0.0f;
  }
  /** reference: cost.cuh:186-195 */
  __device__ inline float computeFeedbackCost(float* fb_u, float* std_dev, float lambda = 1.0f, float alpha = 0.0f)
  {
    float cost = 0.0f;
    for (int i = 0; i < CONTROL_DIM; i++)
    {
      cost += params_.control_cost_coeff[i] * SQ(fb_u[i] / std_dev[i]);
    }
    return 0.5f * lambda * (1.0f - alpha) * cost;
  }
  /** reference: cost.cu:39-53 */
  __device__ inline float computeRunningCost(float* y, float* u, int timestep, float* theta_c, int* crash)
  {
    if (__builtin_amdgcn_workitem_id_y() == 0)
    {
      CLASS_T* derived = static_cast<CLASS_T*>(this);
      return derived->computeStateCost(y, timestep, theta_c, crash) +
             derived->computeControlCost(u, timestep, theta_c, crash);
    }
    else
    {
      return 0.0f;
    }
  }

  CLASS_T* cost_d_ = nullptr;
  PARAMS_T params_;
};

#endif
