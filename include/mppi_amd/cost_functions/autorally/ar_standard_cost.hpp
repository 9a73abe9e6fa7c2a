/**
 * ARStandardCost — AutoRally track / speed / slip / crash cost over a costmap.
 * reference: include/mppi/cost_functions/autorally/ar_standard_cost.cuh:14-41, 215-219; ar_standard_cost.cu:224-243
 * (queryTextureTransformed), :283-413 (getSpeedCost, getStabilizingCost, getCrashCost, getTrackCost, computeStateCost).
 *
 * Costmap: the reference uploads a float4 CUDA texture (point filter, clamp, normalised coordinates,
 * ar_standard_cost.cu:137-173) and only ever consumes channel .x (:356, :360).  Here channel 0 is a plain
 * `float[height][width]` array in HBM (a 600x600 map is 1.4 MB and lives in the XCD's L2) and the texture unit's
 * point sampling is restated in code: texel = clamp(floor(u*width)), clamp(floor(v*height)).
 */
#ifndef MPPI_AMD_AR_STANDARD_COST_HPP_
#define MPPI_AMD_AR_STANDARD_COST_HPP_

#include "mppi_amd/plugin/cost.hpp"
#include "mppi_amd/dynamics/autorally/ar_nn_model.hpp"

struct ARStandardCostParams : public CostParams<2>
{
  float desired_speed = 6.0;
  float speed_coeff = 4.25;
  float track_coeff = 200.0;
  float max_slip_ang = 1.25;
  float slip_coeff = 10.0;
  float track_slop = 0;
  float crash_coeff = 10000;
  float boundary_threshold = 0.65;
  int grid_res = 10;
  /* projective transform world -> texture [0,1]^2:  (r_c1 | r_c2 | trs), columns as in the reference */
  float r_c1[3] = { 1, 0, 0 };
  float r_c2[3] = { 0, 1, 0 };
  float trs[3] = { 0, 0, 1 };

  ARStandardCostParams()
  {
    control_cost_coeff[0] = 0.0;  // steering_coeff
    control_cost_coeff[1] = 0.0;  // throttle_coeff
  }
};

template <class CLASS_T, class PARAMS_T = ARStandardCostParams, class DYN_PARAMS_T = NNDynamicsParams>
class ARStandardCostImpl : public Cost<CLASS_T, PARAMS_T, DYN_PARAMS_T>
{
public:
  using PARENT_CLASS = Cost<CLASS_T, PARAMS_T, DYN_PARAMS_T>;
  static constexpr float MAX_COST_VALUE = 1e16;

  ARStandardCostImpl(hipStream_t stream = 0)
  {
    this->bindToStream(stream);
  }

  /** channel 0 of the costmap, row-major [height][width], device pointer owned by the engine */
  const float* costmap_d_ = nullptr;
  int width_ = -1, height_ = -1;

  /** reference: ar_standard_cost.cu:211-222 */
  __device__ inline void coorTransform(float x, float y, float* u, float* v, float* w) const
  {
    u[0] = this->params_.r_c1[0] * x + this->params_.r_c2[0] * y + this->params_.trs[0];
    v[0] = this->params_.r_c1[1] * x + this->params_.r_c2[1] * y + this->params_.trs[1];
    w[0] = this->params_.r_c1[2] * x + this->params_.r_c2[2] * y + this->params_.trs[2];
  }

  /** point-sampled, clamped, normalised-coordinate fetch of channel 0 (ar_standard_cost.cu:224-243 device branch) */
  __device__ inline float queryTextureTransformed(float x, float y) const
  {
    float u, v, w;
    coorTransform(x, y, &u, &v, &w);
    const float fx = floorf(u / w * (float)width_);
    const float fy = floorf(v / w * (float)height_);
    int ix = (fx >= 0.0f) ? ((fx < (float)width_) ? (int)fx : width_ - 1) : 0;   // NaN -> 0
    int iy = (fy >= 0.0f) ? ((fy < (float)height_) ? (int)fy : height_ - 1) : 0;
    return costmap_d_[iy * width_ + ix];
  }

  __device__ inline float terminalCost(float* s, float* theta_c)
  {
    return 0.0;
  }

  /** reference: ar_standard_cost.cu:283-298 (l1_cost_ = false) */
  __device__ inline float getSpeedCost(float* s, int* crash)
  {
    float error = s[4] - this->params_.desired_speed;
    float cost = l1_cost_ ? fabsf(error) : error * error;
    return (this->params_.speed_coeff * cost);
  }

  /** reference: ar_standard_cost.cu:300-322; fabs(s[4]) > 0.001 and fabs(s[3]) > M_PI_2 are double comparisons */
  __device__ inline float getStabilizingCost(float* s, int* crash_status)
  {
    float stabilizing_cost = 0;
    if ((double)fabsf(s[4]) > 0.001)
    {
      float slip = -mppi::det::atan(s[5] / fabsf(s[4]));
      stabilizing_cost = this->params_.slip_coeff * (slip * slip);
      if (fabsf(slip) > this->params_.max_slip_ang)
      {
        // beyond max_slip_ang the rollout pays the crash cost on top of the quadratic slip term
        stabilizing_cost += this->params_.crash_coeff;
      }
    }
    // a roll angle beyond 90 degrees raises the crash flag (sticky for the rest of the rollout)
    if ((double)fabsf(s[3]) > 1.57079632679489661923)
    {
      crash_status[0] = 1;
    }
    return stabilizing_cost;
  }

  /** reference: ar_standard_cost.cu:324-336 */
  __device__ inline float getCrashCost(float* s, int* crash, int num_timestep)
  {
    float crash_cost = 0;
    if (crash[0] > 0)
    {
      crash_cost = this->params_.crash_coeff;
    }
    return crash_cost;
  }

  /** reference: ar_standard_cost.cu:338-383 (device branch: __cosf/__sinf -> det::sincos) */
  __device__ inline float getTrackCost(float* s, int* crash)
  {
    float sy, cy;
    mppi::det::sincos(s[2], &sy, &cy);
    float x_front = s[0] + FRONT_D * cy;
    float y_front = s[1] + FRONT_D * sy;
    float x_back = s[0] + BACK_D * cy;
    float y_back = s[1] + BACK_D * sy;

    float track_cost_front = queryTextureTransformed(x_front, y_front);
    float track_cost_back = queryTextureTransformed(x_back, y_back);

    float track_cost = (fabsf(track_cost_front) + fabsf(track_cost_back)) / 2.0f;
    if (fabsf(track_cost) < this->params_.track_slop)
    {
      track_cost = 0;
    }
    else
    {
      track_cost = this->params_.track_coeff * track_cost;
    }
    if (track_cost_front >= this->params_.boundary_threshold || track_cost_back >= this->params_.boundary_threshold)
    {
      crash[0] = 1;
    }
    return track_cost;
  }

  /** reference: ar_standard_cost.cu:385-413 */
  __device__ inline float computeStateCost(float* s, int timestep, float* theta_c, int* crash_status)
  {
    float track_cost = getTrackCost(s, crash_status);
    float speed_cost = getSpeedCost(s, crash_status);
    float stabilizing_cost = getStabilizingCost(s, crash_status);
    const float disc =
        this->params_.discount == 1.0f ? 1.0f : mppi::det::pow_pos(this->params_.discount, (float)timestep);
    float crash_cost = disc * getCrashCost(s, crash_status, timestep);
    float cost = speed_cost + crash_cost + track_cost + stabilizing_cost;
    if (cost > MAX_COST_VALUE || cost != cost)
    {
      cost = MAX_COST_VALUE;
    }
    return cost;
  }

  const float FRONT_D = 0.5;  ///< front query point of the track cost, metres ahead of the reference point (ar_standard_cost.cuh:75-76)
  const float BACK_D = -0.5;  ///< rear query point, metres behind it
  bool l1_cost_ = false;      ///< L1 speed cost (if false it is L2)
};

class ARStandardCost : public ARStandardCostImpl<ARStandardCost>
{
public:
  /** no block barrier in the per-step device methods: may run on the role-separated kernels (plugin/parallel_utils.hpp) */
  static constexpr bool MPPI_BARRIER_FREE_STEP = true;
  /** a pure parameter block on the device: role loops may run it straight off the kernel's argument block (engine/kernarg_view.hpp) */
  static constexpr bool MPPI_KERNARG_VIEWABLE = true;
  ARStandardCost(hipStream_t stream = 0) : ARStandardCostImpl<ARStandardCost>(stream)
  {
  }
};

#endif
