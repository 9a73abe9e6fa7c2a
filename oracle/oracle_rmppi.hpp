/**
 * oracle_rmppi.hpp — CPU restatement of Robust MPPI.  TEST INFRASTRUCTURE ONLY (see oracle_core.hpp).
 *
 * Follows (paths relative to the reference's include/mppi/):
 *   DeviceDDPImpl::k                               feedback_controllers/DDP/ddp.cu:11-45 (branch structure kept literally)
 *   GaussianDistribution::computeFeedbackCost      sampling_distributions/gaussian/gaussian.cu:571-629
 *   initEvalKernel                                 core/rmppi_kernels.cu:231-356
 *   rolloutRMPPIKernel (per-rollout semantics of the split cost kernel :651-660 for the nominal-cost fix-up,
 *   see mppi-generic_amd/csrc/rmppi_kernels.hpp)   core/rmppi_kernels.cu:666-866
 *   RobustMPPIController: computeLineSearchWeights, computeImportanceSamplerStride, getInitNominalStateCandidates,
 *   computeCandidateBaseline / computeBestIndex, computeNominalStateAndStride, updateImportanceSamplingControl,
 *   computeControl                                 controllers/R-MPPI/robust_mppi_controller.cu:350-362, 480-755
 * The DDP gain producer (include/mppi/ddp/, host Eigen) is not on the path: gains are an input here as in the engine.
 */
#ifndef MPPI_ORACLE_RMPPI_HPP_
#define MPPI_ORACLE_RMPPI_HPP_

#include "oracle_core.hpp"

namespace oracle
{
struct DDPFeedback
{
  int S = 0, C = 0, T = 0;
  std::vector<float> fb_gain_traj; /* [T][S][C] */
  bool accumulate_all_states = false;

  /** reference: ddp.cu:11-45; control_output is zeroed by the caller (rmppi_kernels.cu:755-758) */
  void k(const float* x_act, const float* x_goal, int t, float* control_output) const
  {
    const float* fb_gain_t = &fb_gain_traj[(size_t)S * C * t];
    float e = 0;
    for (int i = 0; i < S; i++)
    {
      e = x_act[i] - x_goal[i];
      if (accumulate_all_states)
      {
        for (int j = 0; j < C; j++)
          control_output[j] += fb_gain_t[i * C + j] * e;
      }
      else if (C % 4 == 0)
      {
        for (int j = 0; j < C / 4; j++)
          for (int l = 0; l < 4; l++)
            control_output[4 * j + l] = fb_gain_t[i * C + 4 * j + l] * e;
      }
      else if (C % 2 == 0)
      {
        for (int j = 0; j < C / 2; j++)
          for (int l = 0; l < 2; l++)
            control_output[2 * j + l] = fb_gain_t[i * C + 2 * j + l] * e;
      }
      else
      {
        for (int j = 0; j < C; j++)
          control_output[j] += fb_gain_t[i * C + j] * e;
      }
    }
  }
};

/** reference: gaussian.cu:571-629 as one thread evaluates it */
inline float feedbackCost(const GaussianSampler& smp, const float* u_fb, int d, int t, float lambda, float alpha)
{
  const int C = smp.C;
  /* gaussian.cu:579-583: std_dev[(d * T + t) * C] when time_specific_std_dev */
  const float* sd = smp.std_dev_time.empty() ? &smp.std_dev[(size_t)d * C] : &smp.std_dev_time[((size_t)d * smp.T + t) * C];
  float cost = 0.0f;
  const int width = (C % 4 == 0) ? 4 : ((C % 2 == 0) ? 2 : 1);
  if (width > 1)
  {
    float lane[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (int i = 0; i < C / width; i++)
      for (int l = 0; l < width; l++)
      {
        const int j = i * width + l;
        lane[l] += smp.control_cost_coeff[j] * (u_fb[j] * u_fb[j]) / (sd[j] * sd[j]);
      }
    if (width == 4)
      cost += lane[0] + lane[1] + lane[2] + lane[3];
    else
      cost += lane[0] + lane[1];
  }
  else
  {
    for (int j = 0; j < C; j++)
      cost += smp.control_cost_coeff[j] * (u_fb[j] * u_fb[j]) / (sd[j] * sd[j]);
  }
  return 0.5f * lambda * (1.0f - alpha) * cost;
}

/**
 * reference: rmppi_kernels.cu:666-866.  System 0 = nominal, 1 = real.
 * x0: [2][S], mean: [2][T][C], v: [2][K][T][C] in/out (feedback-filled clamped controls written back), costs: [2][K]
 */
inline void rmppiRolloutCosts(Dynamics& dyn, Cost& cost, const GaussianSampler& smp, const DDPFeedback& fb, float dt,
                              float lambda, float alpha, float value_func_threshold, const float* x0, const float* mean,
                              float* v, float* costs)
{
  const int S = dyn.S, C = dyn.C, O = dyn.O, K = smp.K, T = smp.T;
  const int NOM = 0, REAL = 1;
  std::vector<float> xs[2][2], xdot[2], u[2], y[2], theta[2];
  for (int z = 0; z < 2; z++)
  {
    xs[z][0].resize(S);
    xs[z][1].resize(S);
    xdot[z].resize(S);
    u[z].resize(C);
    y[z].resize(O);
    theta[z].resize(std::max(1, dyn.scratchFloats()));
  }
  std::vector<float> fb_control(C);
  for (int k = 0; k < K; k++)
  {
    int cur = 0;
    int crash[2] = { 0, 0 };
    float acc_a[2] = { 0.0f, 0.0f }, acc_b[2] = { 0.0f, 0.0f };
    for (int z = 0; z < 2; z++)
    {
      for (int i = 0; i < S; i++)
      {
        xs[z][0][i] = x0[(size_t)z * S + i];
        xdot[z][i] = 0.0f;
      }
      std::fill(u[z].begin(), u[z].end(), 0.0f);
      std::fill(y[z].begin(), y[z].end(), 0.0f);
      std::fill(theta[z].begin(), theta[z].end(), 0.0f);
      dyn.initializeDynamics(xs[z][0].data(), u[z].data(), y[z].data(), theta[z].data(), 0.0f, dt);
      cost.initializeCosts(y[z].data(), u[z].data(), 0.0f, dt);
    }
    for (int t = 0; t < T; t++)
    {
      float* x_nom = xs[NOM][cur].data();
      float* x_real = xs[REAL][cur].data();
      float* vk[2] = { &v[(((size_t)NOM * K + k) * T + t) * C], &v[(((size_t)REAL * K + k) * T + t) * C] };
      for (int z = 0; z < 2; z++)
        for (int i = 0; i < C; i++)
          u[z][i] = vk[z][i];
      std::fill(fb_control.begin(), fb_control.end(), 0.0f);
      fb.k(x_real, x_nom, t, fb_control.data()); /* no feedback on the nominal system (:760-764) */
      for (int i = 0; i < C; i++)
        u[REAL][i] += fb_control[i];
      for (int z = 0; z < 2; z++)
      {
        float* x = xs[z][cur].data();
        float* xn = xs[z][1 - cur].data();
        dyn.enforceConstraints(x, u[z].data());
        for (int i = 0; i < C; i++)
          vk[z][i] = u[z][i];
        dyn.step(x, xn, xdot[z].data(), u[z].data(), y[z].data(), theta[z].data(), t, dt);
        const float curr_cost = cost.computeRunningCost(y[z].data(), u[z].data(), t, &crash[z]);
        const float lr = smp.likelihoodRatioCost(u[z].data(), &mean[((size_t)z * T + t) * C], k, z, lambda, alpha, t);
        if (z == NOM)
        {
          acc_a[z] += curr_cost;
          acc_b[z] += lr;
        }
        else
        {
          acc_a[z] += curr_cost + lr;
          acc_b[z] += curr_cost + feedbackCost(smp, fb_control.data(), z, t, lambda, alpha);
        }
      }
      cur = 1 - cur;
    }
    for (int z = 0; z < 2; z++)
    {
      const float terminal = cost.terminalCost(y[z].data());
      acc_a[z] += terminal;
      if (z != NOM)
        acc_b[z] += terminal;
      acc_a[z] /= (float)T;
      acc_b[z] /= (float)T;
    }
    float nom = 0.5f * acc_a[NOM] + 0.5f * fmaxf(fminf(acc_b[REAL], value_func_threshold), acc_a[NOM]);
    nom += acc_b[NOM];
    costs[(size_t)NOM * K + k] = nom;
    costs[(size_t)REAL * K + k] = acc_a[REAL];
  }
}

/**
 * reference: rmppi_kernels.cu:231-356.  v0: shaped samples of distribution 0, [K][T][C] (read only; sample index =
 * rollout index inside the candidate, time index shifted by the candidate's stride); costs out: [nc * ns]
 */
inline void initEvalCosts(Dynamics& dyn, Cost& cost, const GaussianSampler& smp, float dt, float lambda, float alpha,
                          int nc, int ns, const int* strides, const float* states, const float* mean0, const float* v0,
                          float* costs)
{
  const int S = dyn.S, C = dyn.C, O = dyn.O, T = smp.T;
  std::vector<float> xa(S), xb(S), xdot(S), u(C), y(O), theta(std::max(1, dyn.scratchFloats()));
  for (int g = 0; g < nc * ns; g++)
  {
    const int candidate_idx = g / ns, candidate_sample_idx = g % ns;
    float* x = xa.data();
    float* xn = xb.data();
    for (int i = 0; i < S; i++)
    {
      x[i] = states[(size_t)candidate_idx * S + i];
      xdot[i] = 0.0f;
    }
    std::fill(u.begin(), u.end(), 0.0f);
    std::fill(y.begin(), y.end(), 0.0f);
    std::fill(theta.begin(), theta.end(), 0.0f);
    const int stride = strides[candidate_idx];
    int crash = 0;
    float running = 0.0f;
    dyn.initializeDynamics(x, u.data(), y.data(), theta.data(), 0.0f, dt);
    cost.initializeCosts(y.data(), u.data(), 0.0f, dt);
    for (int t = 0; t < T; t++)
    {
      const int candidate_t = std::min(t + stride, T - 1);
      for (int i = 0; i < C; i++)
        u[i] = v0[((size_t)candidate_sample_idx * T + candidate_t) * C + i];
      dyn.enforceConstraints(x, u.data());
      dyn.step(x, xn, xdot.data(), u.data(), y.data(), theta.data(), t, dt);
      running += cost.computeRunningCost(y.data(), u.data(), t, &crash) +
                 smp.likelihoodRatioCost(u.data(), &mean0[(size_t)t * C], g, 0, lambda, alpha, t);
      std::swap(x, xn);
    }
    costs[g] = running / (float)T + cost.terminalCost(y.data()) / (float)T;
  }
}

/** host state and logic of RobustMPPIController on top of a Controller (D == 2) */
struct RobustController
{
  Controller* c = nullptr;
  DDPFeedback fb;
  float value_function_threshold = 1000.0f;
  int num_candidates = 9, samples_per_candidate = 32;
  bool nominal_state_init = false;
  int best_index = 0, nominal_stride = 0, real_stride = 0;
  std::vector<float> nominal_state, nominal_control_history, line_search_weights, candidate_states, candidate_costs,
      candidate_free_energy;
  std::vector<int> strides;

  void init(Controller* ctrl)
  {
    c = ctrl;
    nominal_state.assign(c->dyn->S, 0.0f);
    nominal_control_history.assign((size_t)2 * c->dyn->C, 0.0f);
    fb.S = c->dyn->S;
    fb.C = c->dyn->C;
    fb.T = c->T;
    fb.fb_gain_traj.assign((size_t)c->T * fb.S * fb.C, 0.0f);
  }
  /** reference: robust_mppi_controller.cu:480-500 */
  void computeLineSearchWeights()
  {
    const int nc = num_candidates;
    line_search_weights.assign((size_t)3 * nc, 0.0f);
    int num_candid_over_2 = nc / 2;
    for (int i = 0; i < num_candid_over_2 + 1; i++)
    {
      line_search_weights[0 * nc + i] = 1 - i / float(num_candid_over_2);
      line_search_weights[1 * nc + i] = i / float(num_candid_over_2);
      line_search_weights[2 * nc + i] = 0.0;
    }
    for (int i = 1; i < num_candid_over_2 + 1; i++)
    {
      line_search_weights[0 * nc + num_candid_over_2 + i] = 0.0;
      line_search_weights[1 * nc + num_candid_over_2 + i] = 1 - i / float(num_candid_over_2);
      line_search_weights[2 * nc + num_candid_over_2 + i] = i / float(num_candid_over_2);
    }
  }
  /** reference: robust_mppi_controller.cu:502-512 */
  void computeImportanceSamplerStride(int stride)
  {
    const int nc = num_candidates;
    strides.resize(nc);
    for (int i = 0; i < nc; i++)
    {
      float acc = 0.0f * line_search_weights[0 * nc + i];
      acc += (float)stride * line_search_weights[1 * nc + i];
      acc += (float)stride * line_search_weights[2 * nc + i];
      strides[i] = (int)roundf(acc);
    }
  }
  /** reference: robust_mppi_controller.cu:350-362 */
  void getInitNominalStateCandidates(const float* nominal_x_k, const float* nominal_x_kp1, const float* real_x_kp1)
  {
    const int nc = num_candidates, S = c->dyn->S;
    candidate_states.assign((size_t)nc * S, 0.0f);
    for (int k = 0; k < nc; k++)
      for (int i = 0; i < S; i++)
      {
        float acc = nominal_x_k[i] * line_search_weights[0 * nc + k];
        acc += nominal_x_kp1[i] * line_search_weights[1 * nc + k];
        acc += real_x_kp1[i] * line_search_weights[2 * nc + k];
        candidate_states[(size_t)k * S + i] = acc;
      }
  }
  /** reference: robust_mppi_controller.cu:514-545 */
  void computeBestIndex()
  {
    const int nc = num_candidates, ns = samples_per_candidate;
    float baseline = candidate_costs[0];
    for (int i = 1; i < nc * ns; i++)
      if (candidate_costs[i] < baseline)
        baseline = candidate_costs[i];
    candidate_free_energy.assign(nc, 0.0f);
    for (int i = 0; i < nc; i++)
    {
      for (int j = 0; j < ns; j++)
        candidate_free_energy[i] +=
            det::exp((float)(-1.0 / c->lambda * (candidate_costs[(size_t)i * ns + j] - baseline)));
      candidate_free_energy[i] /= (1.0 * ns);
      candidate_free_energy[i] = -c->lambda * det::log(candidate_free_energy[i]) + baseline;
      if (candidate_free_energy[i] < value_function_threshold)
        best_index = i;
    }
  }
  /** reference: robust_mppi_controller.cu:548-626.  eps: [K][T][C], consumed only when the nominal state is initialised */
  void updateImportanceSamplingControl(const float* state, int stride, const float* eps)
  {
    const int S = c->dyn->S, C = c->dyn->C, T = c->T, K = c->K;
    real_stride = stride;
    if (!nominal_state_init)
    {
      std::copy(state, state + S, nominal_state.begin());
      nominal_state_init = true;
      nominal_stride = 0;
    }
    else
    {
      computeLineSearchWeights();
      getInitNominalStateCandidates(&c->nominal_state[0], &c->nominal_state[S], state);
      computeImportanceSamplerStride(stride);
      /* copyNominalControlToDevice + generateSamples(stride, 0): distribution 0 around the nominal control */
      std::vector<float> mean((size_t)2 * T * C), v((size_t)2 * K * T * C);
      std::copy(c->nominal_control.begin(), c->nominal_control.end(), mean.begin());
      std::copy(c->nominal_control.begin(), c->nominal_control.end(), mean.begin() + (size_t)T * C);
      c->smp.setGaussianControls(mean.data(), eps, stride, 0, v.data());
      candidate_costs.assign((size_t)num_candidates * samples_per_candidate, 0.0f);
      initEvalCosts(*c->dyn, *c->cost, c->smp, c->dt, c->lambda, c->alpha, num_candidates, samples_per_candidate,
                    strides.data(), candidate_states.data(), mean.data(), v.data(), candidate_costs.data());
      computeBestIndex();
      c->stats.nominal_state_used = best_index;
      nominal_stride = strides[best_index];
      std::copy(candidate_states.begin() + (size_t)best_index * S, candidate_states.begin() + (size_t)(best_index + 1) * S,
                nominal_state.begin());
    }
    saveControlHistory(nominal_stride, c->nominal_control.data(), nominal_control_history.data(), C);
    saveControlHistory(real_stride, c->control.data(), c->control_history.data(), C);
    slideControlSequence(c->nominal_control.data(), T, C, nominal_stride, c->dyn->zero_control.data(),
                         c->slide_scale.data());
    computeStateTrajectory(*c->dyn, c->dt, nominal_state.data(), c->nominal_control.data(), T, c->nominal_state.data());
  }
  /** reference: robust_mppi_controller.cu:635-755.  eps: [num_iters][K][T][C] */
  void computeControl(const float* state, int stride, const float* eps)
  {
    const int S = c->dyn->S, C = c->dyn->C, T = c->T, K = c->K;
    std::vector<float> x0((size_t)2 * S), mean((size_t)2 * T * C), u_new((size_t)2 * T * C);
    for (int i = 0; i < S; i++)
    {
      x0[i] = nominal_state[i];
      x0[S + i] = state[i];
    }
    for (int it = 0; it < c->num_iters; it++)
    {
      std::copy(c->nominal_control.begin(), c->nominal_control.end(), mean.begin());
      std::copy(c->nominal_control.begin(), c->nominal_control.end(), mean.begin() + (size_t)T * C);
      c->smp.setGaussianControls(mean.data(), eps + (size_t)it * K * T * C, stride, it, c->v.data());
      rmppiRolloutCosts(*c->dyn, *c->cost, c->smp, fb, c->dt, c->lambda, c->alpha, value_function_threshold, x0.data(),
                        mean.data(), c->v.data(), c->costs.data());
      c->w = c->costs;
      for (int d = 0; d < 2; d++)
      {
        float* wd = &c->w[(size_t)d * K];
        c->stats.baseline[d] = computeBaselineCost(wd, K);
        normExpTransform(wd, K, (float)(1.0 / c->lambda), c->stats.baseline[d]);
        c->stats.normalizer[d] = computeNormalizer(wd, K);
        computeFreeEnergy(c->stats.free_energy[d], c->stats.free_energy_var[d], c->stats.free_energy_mod[d], wd, K,
                          c->stats.baseline[d], c->lambda);
        weightedReduction(wd, &c->v[(size_t)d * K * T * C], c->stats.normalizer[d], K, T, C, c->smp.sum_strides,
                          &u_new[(size_t)d * T * C]);
      }
      std::copy(u_new.begin(), u_new.begin() + (size_t)T * C, c->nominal_control.begin());
      std::copy(u_new.begin() + (size_t)T * C, u_new.end(), c->control.begin());
    }
    smoothControlTrajectory(c->control.data(), c->control_history.data(), T, C);
    smoothControlTrajectory(c->nominal_control.data(), nominal_control_history.data(), T, C);
    computeStateTrajectory(*c->dyn, c->dt, nominal_state.data(), c->nominal_control.data(), T, c->nominal_state.data());
  }
};
}  // namespace oracle
#endif
