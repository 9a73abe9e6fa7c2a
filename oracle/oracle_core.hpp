/**
 * oracle_core.hpp — CPU restatement of the MPPI-Generic hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link, import or call anything in
 * oracle/.  The product (mppi-generic_amd/) never includes or links this directory.
 *
 * What it restates (all paths relative to the reference's include/mppi/ unless noted; "device flavour" means the
 * formula the reference's GPU kernels evaluate, which is the thing the HIP engine replaces):
 *   - the per-rollout loop of rolloutKernel                    core/mppi_common.cu:28-146, restated with the structure
 *     of the reference's own CPU twin                          tests/include/kernel_tests/core/rollout_kernel_test.cu:505-541
 *   - setGaussianControls                                      sampling_distributions/gaussian/gaussian.cu:17-277 (rule :99-127)
 *   - device computeLikelihoodRatioCost                        sampling_distributions/gaussian/gaussian.cu:480-569
 *   - Dynamics::enforceConstraints / step / updateState        dynamics/dynamics.cu:97-142
 *   - Cost::computeRunningCost                                 cost_functions/cost.cu:39-53
 *   - computeBaselineCost / normExpTransform / computeNormalizer / computeFreeEnergy
 *                                                              core/mppi_common.cu:858-900, 958-966, 1055-1081
 *   - weightedReductionKernel summation order                  core/mppi_common.cu:710-737, 1086-1160
 *   - smoothing / slide / history / state re-rollout           controllers/controller.cuh:557-615, 643-663
 *   - VanillaMPPIController::computeControl                    controllers/MPPI/mppi_controller.cu:151-241
 *   - TubeMPPIController::computeControl                       controllers/Tube-MPPI/tube_mppi_controller.cu:157-341
 *   - ColoredMPPIController::computeControl                    controllers/ColoredMPPI/colored_mppi_controller.cu:134-240
 *   - ColoredNoiseDistribution::generateSamples                oracle_colored.hpp
 *
 * Arithmetic: IEEE fp32 in the reference's expression order, compiled with -ffp-contract=off; transcendentals go
 * through include/mppi_amd/det_math.h (bit-reproducible on host and gfx950, see that header for why).  A rollout is
 * evaluated the way one device thread with blockDim.y == 1 evaluates it.
 *
 * Pinning: see oracle/README.md — checked against the reference's own known-answer tests (tests/test_oracle_kat.py).
 */
#ifndef MPPI_ORACLE_CORE_HPP_
#define MPPI_ORACLE_CORE_HPP_

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "mppi_amd/det_math.h"

namespace oracle
{
namespace det = mppi::det;

/* ------------------------------------------------------------------------------------------------------------------
 * Plugin interfaces (runtime polymorphic on purpose: the product is CRTP/templates)
 * ------------------------------------------------------------------------------------------------------------------ */
struct Dynamics
{
  int S = 0, C = 0, O = 0;
  std::vector<float> rng_lo, rng_hi, deadband, zero_control;

  Dynamics(int s, int c, int o) : S(s), C(c), O(o)
  {
    /* reference: dynamics/dynamics.cuh:84-93 — ranges default to +-FLT_MAX, :512 deadband 0, :121 zero control 0 */
    rng_lo.assign(c, -FLT_MAX);
    rng_hi.assign(c, FLT_MAX);
    deadband.assign(c, 0.0f);
    zero_control.assign(c, 0.0f);
  }
  virtual ~Dynamics() = default;
  virtual int setParams(const void* pod, size_t nbytes) = 0;
  /** per-rollout persistent scratch (the device's theta_s slot); size in floats */
  virtual int scratchFloats() const
  {
    return 0;
  }
  /** reference: dynamics/dynamics.cuh:429-435 — the default copies the state into the output */
  virtual void initializeDynamics(const float* x, const float* u, float* y, float* theta_s, float t0, float dt)
  {
    for (int i = 0; i < O && i < S; i++)
      y[i] = x[i];
  }
  virtual void computeKinematics(const float* x, float* xdot)
  {
  }
  virtual void computeDynamics(const float* x, const float* u, float* xdot, float* theta_s) = 0;

  /** reference: dynamics/dynamics.cu:97-116; sign(): utils/math_utils.h:744-747 (float overload: >= 0 ? 1 : -1) */
  virtual void enforceConstraints(float* x, float* u) const
  {
    for (int i = 0; i < C; i++)
    {
      if (fabsf(u[i]) < deadband[i])
      {
        u[i] = zero_control[i];
      }
      else
      {
        u[i] += deadband[i] * -(u[i] >= 0 ? 1.0f : -1.0f);
      }
      u[i] = fminf(fmaxf(rng_lo[i], u[i]), rng_hi[i]);
    }
  }
  /** reference: Dynamics::enforceLeash, dynamics/dynamics.cuh:448-466 (base rule; the RACER models override it) */
  virtual void enforceLeash(const float* x_true, const float* x_nominal, const float* leash, float* out) const
  {
    for (int i = 0; i < S; i++)
    {
      const float diff = fabsf(x_nominal[i] - x_true[i]);
      if (leash[i] < diff)
        out[i] = x_true[i] + fminf(fmaxf(x_nominal[i] - x_true[i], -leash[i]), leash[i]);
      else
        out[i] = x_nominal[i];
    }
  }
  /** reference: dynamics/dynamics.cu:118-128 */
  virtual void updateState(const float* x, float* x_next, const float* xdot, float dt) const
  {
    for (int i = 0; i < S; i++)
    {
      x_next[i] = x[i] + xdot[i] * dt;
    }
  }
  /** reference: dynamics/dynamics.cu:144-155 */
  virtual void stateToOutput(const float* x, float* y) const
  {
    for (int i = 0; i < O && i < S; i++)
    {
      y[i] = x[i];
    }
  }
  /** reference: dynamics/dynamics.cu:82-95 (computeStateDeriv) and :130-142 (step) */
  virtual void step(float* x, float* x_next, float* xdot, const float* u, float* y, float* theta_s, int t, float dt)
  {
    computeKinematics(x, xdot);
    computeDynamics(x, u, xdot, theta_s);
    updateState(x, x_next, xdot, dt);
    stateToOutput(x_next, y);
  }
};

struct Cost
{
  int C = 0, O = 0;
  Cost(int c, int o) : C(c), O(o)
  {
  }
  virtual ~Cost() = default;
  virtual int setParams(const void* pod, size_t nbytes) = 0;
  virtual void initializeCosts(const float* y, const float* u, float t0, float dt)
  {
  }
  virtual float computeStateCost(const float* y, int t, int* crash) = 0;
  /** reference: cost_functions/cost.cuh:205-208 — the base control cost is 0 (it lives in the sampler) */
  virtual float computeControlCost(const float* u, int t, int* crash)
  {
    return 0.0f;
  }
  virtual float terminalCost(const float* y) = 0;
  /** reference: cost_functions/cost.cu:39-53, the threadIdx.y == 0 lane */
  float computeRunningCost(const float* y, const float* u, int t, int* crash)
  {
    return computeStateCost(y, t, crash) + computeControlCost(u, t, crash);
  }
};

/** Parameters of the Gaussian sampler that matter on the path (reference: gaussian/gaussian.cuh:21-61). */
struct GaussianSampler
{
  int C = 0, D = 1, K = 0, T = 0;
  std::vector<float> std_dev;            /* [D][C]  (time_specific_std_dev == false) */
  bool independent_noise = false;        /* use_same_noise_for_all_distributions == false: eps is [D][K][T][C] */
  std::vector<float> std_dev_time;       /* [D][T][C] when time_specific_std_dev is on (gaussian.cuh:64-95), else empty */
  std::vector<float> control_cost_coeff; /* [C] */
  float pure_noise_trajectories_percentage = 0.01f;
  float std_dev_decay = 1.0f;
  int sum_strides = 32;

  void init(int c, int d, int k, int t)
  {
    C = c;
    D = d;
    K = k;
    T = t;
    std_dev.assign((size_t)d * c, 1.0f);
    control_cost_coeff.assign(c, 0.0f);
  }

  bool isPureNoise(int k) const
  {
    /* reference: gaussian.cu:108 / :512 — float compare of the int index against (1 - p) * K */
    return (float)k >= (1.0f - pure_noise_trajectories_percentage) * (float)K;
  }

  /**
   * reference: gaussian.cu:17-277 (rule at :99-127) with std_dev_decay^iter from :421.
   * eps: [K][T][C] shared by all distributions (use_same_noise_for_all_distributions, gaussian.cu:376-389).
   * mean: [D][T][C].  v out: [D][K][T][C].
   */
  void setGaussianControls(const float* mean, const float* eps, int optimization_stride, int iteration, float* v) const
  {
    const float decay = powf_int(std_dev_decay, iteration);
    for (int d = 0; d < D; d++)
      for (int k = 0; k < K; k++)
        for (int t = 0; t < T; t++)
          for (int c = 0; c < C; c++)
          {
            const size_t vi = (((size_t)d * K + k) * T + t) * C + c;
            /* gaussian.cu:378-394: one block of noise copied to every distribution, or one block per distribution */
            const float e = eps[(independent_noise ? (size_t)d * K * T * C : 0) + ((size_t)k * T + t) * C + c];
            const float m = mean[((size_t)d * T + t) * C + c];
            /* gaussian.cu:21-43: the std-dev index carries the time step when time_specific_std_dev is set */
            const float sd = decay * (std_dev_time.empty() ? std_dev[(size_t)d * C + c] : std_dev_time[((size_t)d * T + t) * C + c]);
            if (k == 0 || t < optimization_stride)
              v[vi] = m;
            else if (isPureNoise(k))
              v[vi] = sd * e;
            else
              v[vi] = m + sd * e;
          }
  }

  /** std_dev_decay^iter: the reference calls host powf (gaussian.cu:421); decay == 1 (default) and iter == 0 are exact.
   *  For other values the product and the oracle both use repeated multiplication (documented deviation, <= 1 ulp/step). */
  static float powf_int(float b, int n)
  {
    float r = 1.0f;
    for (int i = 0; i < n; i++)
      r *= b;
    return r;
  }

  /**
   * Device flavour of computeLikelihoodRatioCost, reference: gaussian.cu:480-569, as evaluated by ONE thread
   * (blockDim.y == 1): the CONTROL_DIM % 4 / % 2 / scalar branches accumulate per vector lane and then add the lanes.
   */
  float likelihoodRatioCost(const float* u, const float* mean_dt /* mean[d][t][:] */, int k, int d, float lambda,
                            float alpha, int t = 0) const
  {
    /* gaussian.cu:488-493: std_dev[(d * T + t) * C] when time_specific_std_dev */
    const float* sd = std_dev_time.empty() ? &std_dev[(size_t)d * C] : &std_dev_time[((size_t)d * T + t) * C];
    const bool pure = isPureNoise(k);
    float cost = 0.0f;
    const int width = (C % 4 == 0) ? 4 : ((C % 2 == 0) ? 2 : 1);
    if (width > 1)
    {
      float lane[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
      for (int i = 0; i < C / width; i++)
        for (int l = 0; l < width; l++)
        {
          const int j = i * width + l;
          const float m = pure ? 0.0f : mean_dt[j];
          lane[l] += control_cost_coeff[j] * m * (m - 2.0f * u[j]) / (sd[j] * sd[j]);
        }
      if (width == 4)
        cost += lane[0] + lane[1] + lane[2] + lane[3];
      else
        cost += lane[0] + lane[1];
    }
    else
    {
      for (int j = 0; j < C; j++)
      {
        const float m = pure ? 0.0f : mean_dt[j];
        cost += control_cost_coeff[j] * m * (m - 2.0f * u[j]) / (sd[j] * sd[j]);
      }
    }
    return 0.5f * lambda * (1.0f - alpha) * cost;
  }
};

/* ------------------------------------------------------------------------------------------------------------------
 * Rollout (device flavour of rolloutKernel; loop structure of launchCPURolloutKernel)
 * ------------------------------------------------------------------------------------------------------------------ */
/**
 * x0: [D][S], mean: [D][T][C], v: [D][K][T][C] in (samples from setGaussianControls) / out (clamped controls are
 * written back, mppi_common.cu:110-117), costs out: [D][K].
 */
inline void rolloutCosts(Dynamics& dyn, Cost& cost, const GaussianSampler& smp, float dt, float lambda, float alpha,
                         const float* x0, const float* mean, float* v, float* costs, int k_begin = 0, int k_end = -1)
{
  const int S = dyn.S, C = dyn.C, O = dyn.O, K = smp.K, T = smp.T, D = smp.D;
  if (k_end < 0)
    k_end = K;
  std::vector<float> xa(S), xb(S), xdot(S), u(C), y(O), theta(std::max(1, dyn.scratchFloats()));
  for (int d = 0; d < D; d++)
  {
    for (int k = k_begin; k < k_end; k++)
    {
      float* x = xa.data();
      float* xn = xb.data();
      /* loadGlobalToShared: x = x0[d], xdot = 0, u = 0 (mppi_common.cu:770-840) */
      for (int i = 0; i < S; i++)
      {
        x[i] = x0[(size_t)d * S + i];
        xdot[i] = 0.0f;
      }
      std::fill(u.begin(), u.end(), 0.0f);
      std::fill(y.begin(), y.end(), 0.0f);
      std::fill(theta.begin(), theta.end(), 0.0f);
      int crash = 0;
      float running = 0.0f;
      dyn.initializeDynamics(x, u.data(), y.data(), theta.data(), 0.0f, dt);
      cost.initializeCosts(y.data(), u.data(), 0.0f, dt);
      for (int t = 0; t < T; t++)
      {
        float* vkt = &v[(((size_t)d * K + k) * T + t) * C];
        for (int i = 0; i < C; i++)
          u[i] = vkt[i];
        dyn.enforceConstraints(x, u.data());
        for (int i = 0; i < C; i++)
          vkt[i] = u[i];
        dyn.step(x, xn, xdot.data(), u.data(), y.data(), theta.data(), t, dt);
        /* mppi_common.cu:122-124: running += (runningCost + likelihoodRatioCost) */
        running += cost.computeRunningCost(y.data(), u.data(), t, &crash) +
                   smp.likelihoodRatioCost(u.data(), &mean[((size_t)d * T + t) * C], k, d, lambda, alpha, t);
        std::swap(x, xn);
      }
      /* mppi_common.cu:144 and computeAndSaveCost :843-853: running/T + terminal/T */
      costs[(size_t)d * K + k] = running / (float)T + cost.terminalCost(y.data()) / (float)T;
    }
  }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Weights (host flavour — this IS what the reference's controllers run, mppi_controller.cu:187-214)
 * ------------------------------------------------------------------------------------------------------------------ */
/** reference: core/mppi_common.cu:885-900 — first-occurring minimum, linear scan */
inline int computeBestIndex(const float* costs, int K)
{
  float best = costs[0];
  int idx = 0;
  for (int i = 1; i < K; i++)
    if (costs[i] < best)
    {
      best = costs[i];
      idx = i;
    }
  return idx;
}
/** reference: core/mppi_common.cu:858-863 */
inline float computeBaselineCost(const float* costs, int K)
{
  return costs[computeBestIndex(costs, K)];
}
/** reference: core/mppi_common.cu:958-966; expf -> det::exp */
inline void normExpTransform(float* costs, int K, float lambda_inv, float baseline)
{
  for (int i = 0; i < K; i++)
  {
    const float cost_dif = costs[i] - baseline;
    costs[i] = det::exp(-lambda_inv * cost_dif);
  }
}
/** reference: core/mppi_common.cu:1055-1063 — double accumulate, returned as float */
inline float computeNormalizer(const float* w, int K)
{
  double normalizer = 0.0;
  for (int i = 0; i < K; i++)
    normalizer += w[i];
  return (float)normalizer;
}
/** reference: core/mppi_common.cu:1065-1081; logf -> det::log */
inline void computeFreeEnergy(float& fe, float& fe_var, float& fe_mod, const float* w, int K, float baseline,
                              float lambda)
{
  float var = 0, norm = 0;
  for (int i = 0; i < K; i++)
  {
    norm += w[i];
    var += w[i] * w[i];
  }
  norm /= K;
  fe = -lambda * det::log(norm) + baseline;
  fe_var = lambda * (var / K - norm * norm);
  const float weird = fe_var / (norm * det::sqrt(1.0f * K));
  fe_mod = lambda * (weird + 0.5f * (weird * weird));
}

/**
 * reference: core/mppi_common.cu:1115-1160 — thread j sums sum_stride consecutive rollouts serially with
 * weight = w/eta computed per rollout, then thread 0 sums the ceil(K/sum_stride) partials serially.
 * v: [K][T][C] (one distribution), u_out: [T][C].
 */
/** `inter += weight * v` is two roundings here by default (this file is compiled with -ffp-contract=off; it is also what
 *  the reference's own CPU statement of the kernel computes); nvcc's default -fmad=true contracts it to one fma on the
 *  reference's GPU path — g_weighted_reduction_fma selects that flavour (oracle_set_reduction_fma) */
inline bool& weightedReductionFma()
{
  static bool fma_flavour = false;
  return fma_flavour;
}
inline void weightedReduction(const float* w, const float* v, float normalizer, int K, int T, int C, int sum_stride,
                              float* u_out)
{
  const bool use_fma = weightedReductionFma();
  const int cells = (K - 1) / sum_stride + 1;
  std::vector<float> inter((size_t)cells * C);
  for (int t = 0; t < T; t++)
  {
    std::fill(inter.begin(), inter.end(), 0.0f);
    for (int j = 0; j < cells; j++)
      for (int i = 0; i < sum_stride; i++)
      {
        const int k = j * sum_stride + i;
        if (k < K)
        {
          const float weight = w[k] / normalizer;
          for (int c = 0; c < C; c++)
          {
            float& cell = inter[(size_t)j * C + c];
            const float s = v[((size_t)k * T + t) * C + c];
            if (use_fma)
              cell = det::fma(weight, s, cell);
            else
              cell += weight * s;
          }
        }
      }
    for (int c = 0; c < C; c++)
    {
      float acc = 0.0f;
      for (int j = 0; j < cells; j++)
        acc += inter[(size_t)j * C + c];
      u_out[(size_t)t * C + c] = acc;
    }
  }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Host post-processing (controllers/controller.cuh)
 * ------------------------------------------------------------------------------------------------------------------ */
/**
 * reference: controllers/controller.cuh:557-586.  u: [T][C] in/out, history: [2][C] (row 0 older).
 * Buffer = [hist0, hist1, u_0..u_{T-1}, u_{T-1}, u_{T-1}]; 5-tap [-3,12,17,12,-3]/35 (coefficients divided first,
 * in float, as Eigen's `filter_coefficients /= 35.0` does); products summed in tap order.
 */
inline void smoothControlTrajectory(float* u, const float* history, int T, int C)
{
  float coef[5] = { -3.0f, 12.0f, 17.0f, 12.0f, -3.0f };
  for (float& c : coef)
    c = (float)(c / 35.0);
  std::vector<float> buf((size_t)(T + 4) * C);
  for (int c = 0; c < C; c++)
  {
    buf[0 * C + c] = history[0 * C + c];
    buf[1 * C + c] = history[1 * C + c];
  }
  for (int t = 0; t < T; t++)
    for (int c = 0; c < C; c++)
      buf[(size_t)(t + 2) * C + c] = u[(size_t)t * C + c];
  for (int c = 0; c < C; c++)
  {
    buf[(size_t)(T + 2) * C + c] = u[(size_t)(T - 1) * C + c];
    buf[(size_t)(T + 3) * C + c] = u[(size_t)(T - 1) * C + c];
  }
  for (int t = 0; t < T; t++)
    for (int c = 0; c < C; c++)
    {
      float acc = coef[0] * buf[(size_t)(t + 0) * C + c];
      for (int j = 1; j < 5; j++)
        acc += coef[j] * buf[(size_t)(t + j) * C + c];
      u[(size_t)t * C + c] = acc;
    }
}

/** reference: controllers/controller.cuh:588-600 */
inline void slideControlSequence(float* u, int T, int C, int steps, const float* zero_control, const float* slide_scale)
{
  for (int i = 0; i < T; i++)
  {
    const int ind = std::min(i + steps, T - 1);
    for (int c = 0; c < C; c++)
    {
      u[(size_t)i * C + c] = u[(size_t)ind * C + c];
      if (i + steps > T - 1)
        u[(size_t)i * C + c] = (u[(size_t)ind * C + c] - zero_control[c]) * slide_scale[c] + zero_control[c];
    }
  }
}

/** reference: controllers/controller.cuh:602-615.  history: [2][C] */
inline void saveControlHistory(int steps, const float* u, float* history, int C)
{
  if (steps == 1)
  {
    for (int c = 0; c < C; c++)
    {
      history[c] = history[C + c];
      history[C + c] = u[c];
    }
  }
  else if (steps >= 2)
  {
    for (int c = 0; c < C; c++)
    {
      history[c] = u[(size_t)(steps - 2) * C + c];
      history[C + c] = u[(size_t)(steps - 1) * C + c];
    }
  }
}

/** reference: controllers/controller.cuh:643-663 (computeStateTrajectoryHelper) and :643-662 (computeOutputTrajectoryHelper:
 *  the same loop, also keeping the output after initializeDynamics and after every step).  result: [T][S]; u: [T][C] (not
 *  modified); output_result: [T][O] or nullptr */
inline void computeStateTrajectory(Dynamics& dyn, float dt, const float* x0, const float* u, int T, float* result,
                                   float* output_result = nullptr)
{
  const int S = dyn.S, C = dyn.C, O = dyn.O;
  std::vector<float> x(S), xn(S), xdot(S, 0.0f), ui(C), y(O, 0.0f), theta(std::max(1, dyn.scratchFloats()), 0.0f);
  for (int i = 0; i < S; i++)
    result[i] = x0[i];
  for (int c = 0; c < C; c++)
    ui[c] = u[c];
  dyn.initializeDynamics(result, ui.data(), y.data(), theta.data(), 0.0f, dt);
  if (output_result)
    for (int i = 0; i < O; i++)
      output_result[i] = y[i];
  for (int t = 0; t < T - 1; t++)
  {
    for (int i = 0; i < S; i++)
      x[i] = result[(size_t)t * S + i];
    for (int c = 0; c < C; c++)
      ui[c] = u[(size_t)t * C + c];
    dyn.enforceConstraints(x.data(), ui.data());
    dyn.step(x.data(), xn.data(), xdot.data(), ui.data(), y.data(), theta.data(), t, dt);
    for (int i = 0; i < S; i++)
      result[(size_t)(t + 1) * S + i] = xn[i];
    if (output_result)
      for (int i = 0; i < O; i++)
        output_result[(size_t)(t + 1) * O + i] = y[i];
  }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Controllers
 * ------------------------------------------------------------------------------------------------------------------ */
struct Stats
{
  float baseline[2] = { 0, 0 };
  float normalizer[2] = { 0, 0 };
  float free_energy[2] = { 0, 0 };
  float free_energy_var[2] = { 0, 0 };
  float free_energy_mod[2] = { 0, 0 };
  int nominal_state_used = 0;
};

struct Controller
{
  std::unique_ptr<Dynamics> dyn;
  std::unique_ptr<Cost> cost;
  GaussianSampler smp;
  int K = 0, T = 0, D = 1;
  float dt = 0.01f, lambda = 1.0f, alpha = 0.0f;
  int num_iters = 1;
  float nominal_threshold = 20.0f; /* tube_mppi_controller.cuh:20 */
  std::vector<float> slide_scale;  /* controller.cuh:67 slide_control_scale_, default Zero() */

  /* host state */
  std::vector<float> control;          /* [T][C] */
  std::vector<float> control_history;  /* [2][C] */
  std::vector<float> state_traj;       /* [T][S] */
  std::vector<float> nominal_control;  /* [T][C]  (tube) */
  std::vector<float> nominal_state;    /* [T][S]  (tube) */
  bool nominal_state_init = false;
  std::vector<float> costs;  /* [D][K] raw trajectory costs of the last iteration */
  std::vector<float> w;      /* [D][K] exp-transformed */
  std::vector<float> v;      /* [D][K][T][C] */
  Stats stats;

  void init(int k, int t, int d)
  {
    K = k;
    T = t;
    D = d;
    smp.init(dyn->C, d, k, t);
    control.assign((size_t)T * dyn->C, 0.0f);
    control_history.assign((size_t)2 * dyn->C, 0.0f);
    state_traj.assign((size_t)T * dyn->S, 0.0f);
    nominal_control.assign((size_t)T * dyn->C, 0.0f);
    nominal_state.assign((size_t)T * dyn->S, 0.0f);
    costs.assign((size_t)D * K, 0.0f);
    w.assign((size_t)D * K, 0.0f);
    v.assign((size_t)D * K * T * dyn->C, 0.0f);
    slide_scale.assign(dyn->C, 0.0f);
  }

  /* ColoredMPPI options (colored_mppi_controller.cuh:18-22, 159-193) */
  float tsallis_gamma = 0.0f, tsallis_r = 0.0f;
  bool leash_active = false;
  int leash_jump = 1;
  std::vector<float> leash_dist;

  /**
   * One pass of the optimisation-loop body for all D systems: sample shaping, rollout, baseline, normExp, normaliser,
   * free energy, weighted reduction.  mean: [D][T][C] in, u_new: [D][T][C] out.
   */
  void iterate(const float* x0 /*[D][S]*/, const float* mean, const float* eps, int stride, int iter, float* u_new)
  {
    const int C = dyn->C;
    smp.setGaussianControls(mean, eps, stride, iter, v.data());
    rolloutCosts(*dyn, *cost, smp, dt, lambda, alpha, x0, mean, v.data(), costs.data());
    w = costs;
    for (int d = 0; d < D; d++)
    {
      float* wd = &w[(size_t)d * K];
      stats.baseline[d] = computeBaselineCost(wd, K);
      if (tsallis_gamma != 0.0f && tsallis_r != 0.0f)
      {  // ColoredMPPI: core/mppi_common.cu:968-985 TsallisTransform (colored_mppi_controller.cu:198-206); expf/logf -> det::
        for (int i = 0; i < K; i++)
        {
          const float cost_dif = wd[i] - stats.baseline[d];
          wd[i] = cost_dif < tsallis_gamma ? det::exp(det::log(1.0f - cost_dif / tsallis_gamma) / (tsallis_r - 1.0f)) : 0.0f;
        }
      }
      else
        /* mppi_controller.cu:201: launchNormExpKernel(..., 1.0 / lambda, ...): double 1.0/lambda narrowed to float */
        normExpTransform(wd, K, (float)(1.0 / lambda), stats.baseline[d]);
      stats.normalizer[d] = computeNormalizer(wd, K);
      computeFreeEnergy(stats.free_energy[d], stats.free_energy_var[d], stats.free_energy_mod[d], wd, K,
                        stats.baseline[d], lambda);
      weightedReduction(wd, &v[(size_t)d * K * T * C], stats.normalizer[d], K, T, C, smp.sum_strides,
                        &u_new[(size_t)d * T * C]);
    }
  }

  /** reference: controllers/MPPI/mppi_controller.cu:151-241.  eps: [num_iters][K][T][C] */
  void vanillaComputeControl(const float* x0, int stride, const float* eps)
  {
    const int C = dyn->C, S = dyn->S;
    std::vector<float> u_new((size_t)T * C);
    for (int it = 0; it < num_iters; it++)
    {
      iterate(x0, control.data(), eps + (size_t)it * K * T * C, stride, it, u_new.data());
      control = u_new;
    }
    smoothControlTrajectory(control.data(), control_history.data(), T, C);
    computeStateTrajectory(*dyn, dt, x0, control.data(), T, state_traj.data());
    std::vector<float> zero_state(S, 0.0f);
    for (int t = 0; t < T; t++)
      dyn->enforceConstraints(zero_state.data(), &control[(size_t)t * C]);
  }

  /**
   * reference: controllers/ColoredMPPI/colored_mppi_controller.cu:134-240 — the vanilla loop (the sampler is the colored
   * one: eps here is its time-domain output, [num_iters][K][T][C]), then smoothing, state trajectory, and ONLY control
   * channel 1 clamped to its range (:232-237; the enforceConstraints call is commented out there).
   */
  void coloredComputeControl(const float* x0_true, int stride, const float* eps)
  {
    const int C = dyn->C, S = dyn->S;
    std::vector<float> u_new((size_t)T * C);
    // state leash (colored_mppi_controller.cu:150-156; Dynamics::enforceLeash dynamics.cuh:448-466, base rule)
    std::vector<float> local_state(x0_true, x0_true + S);
    if (leash_active)
    {
      const float* nominal = &state_traj[(size_t)leash_jump * S];
      dyn->enforceLeash(x0_true, nominal, leash_dist.data(), local_state.data());
    }
    const float* x0 = local_state.data();
    for (int it = 0; it < num_iters; it++)
    {
      iterate(x0, control.data(), eps + (size_t)it * K * T * C, stride, it, u_new.data());
      control = u_new;
    }
    smoothControlTrajectory(control.data(), control_history.data(), T, C);
    computeStateTrajectory(*dyn, dt, x0, control.data(), T, state_traj.data());
    if (C > 1)
      for (int t = 0; t < T; t++)
        control[(size_t)t * C + 1] = fminf(fmaxf(control[(size_t)t * C + 1], dyn->rng_lo[1]), dyn->rng_hi[1]);
  }

  /** reference: controllers/Tube-MPPI/tube_mppi_controller.cu:157-299.  eps: [num_iters][K][T][C] */
  void tubeComputeControl(const float* x0_actual, int stride, const float* eps)
  {
    const int C = dyn->C, S = dyn->S;
    if (!nominal_state_init)
    {
      for (int i = 0; i < S; i++)
        nominal_state[i] = x0_actual[i];
      nominal_state_init = true;
    }
    std::vector<float> x0((size_t)2 * S), mean((size_t)2 * T * C), u_new((size_t)2 * T * C);
    for (int it = 0; it < num_iters; it++)
    {
      for (int i = 0; i < S; i++)
      {
        x0[i] = x0_actual[i];
        x0[S + i] = nominal_state[i];
      }
      std::copy(control.begin(), control.end(), mean.begin());
      std::copy(nominal_control.begin(), nominal_control.end(), mean.begin() + (size_t)T * C);
      iterate(x0.data(), mean.data(), eps + (size_t)it * K * T * C * (smp.independent_noise ? D : 1), stride, it, u_new.data());
      std::copy(u_new.begin(), u_new.begin() + (size_t)T * C, control.begin());
      std::copy(u_new.begin() + (size_t)T * C, u_new.end(), nominal_control.begin());
      tubeComputeStateTrajectory(x0_actual);
      if (stats.baseline[0] < stats.baseline[1] + nominal_threshold)
      {
        stats.nominal_state_used = 0;
        nominal_state = state_traj;
        nominal_control = control;
      }
      else
      {
        stats.nominal_state_used = 1;
      }
    }
    /* :281 smoothControlTrajectory() smooths the NOMINAL control (tube_mppi_controller.cu:325-329) */
    smoothControlTrajectory(nominal_control.data(), control_history.data(), T, C);
    tubeComputeStateTrajectory(x0_actual);
  }

  /** reference: tube_mppi_controller.cu:331-341 */
  void tubeComputeStateTrajectory(const float* x0_actual)
  {
    std::vector<float> x0n(nominal_state.begin(), nominal_state.begin() + dyn->S);
    computeStateTrajectory(*dyn, dt, x0n.data(), nominal_control.data(), T, nominal_state.data());
    computeStateTrajectory(*dyn, dt, x0_actual, control.data(), T, state_traj.data());
  }

  /** reference: controllers/Tube-MPPI/tube_mppi_controller.cu:312-350 — updateNominalState(nominal control column 0): one
   *  in-place model step of nominal_state_trajectory.col(0) without constraints; the control history is taken from the NOMINAL
   *  control; both sequences slide */
  void tubeSlide(int steps)
  {
    const int S = dyn->S, C = dyn->C, O = dyn->O;
    std::vector<float> x(nominal_state.begin(), nominal_state.begin() + S), xn(S), xdot(S, 0.0f), y(O, 0.0f),
        u(nominal_control.begin(), nominal_control.begin() + C), theta(std::max(1, dyn->scratchFloats()), 0.0f);
    dyn->step(x.data(), xn.data(), xdot.data(), u.data(), y.data(), theta.data(), 0, dt);
    std::copy(xn.begin(), xn.end(), nominal_state.begin());
    saveControlHistory(steps, nominal_control.data(), control_history.data(), C);
    slideControlSequence(nominal_control.data(), T, C, steps, dyn->zero_control.data(), slide_scale.data());
    slideControlSequence(control.data(), T, C, steps, dyn->zero_control.data(), slide_scale.data());
  }

  /** reference: controllers/controller.cuh:351-356 (vanilla slide) */
  void vanillaSlide(int steps)
  {
    saveControlHistory(steps, control.data(), control_history.data(), dyn->C);
    slideControlSequence(control.data(), T, dyn->C, steps, dyn->zero_control.data(), slide_scale.data());
  }
};

}  // namespace oracle

#endif  // MPPI_ORACLE_CORE_HPP_
