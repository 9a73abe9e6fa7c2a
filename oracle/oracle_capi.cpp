/**
 * oracle_capi.cpp — C entry points of the CPU oracle for ctypes.  TEST INFRASTRUCTURE ONLY (see oracle_core.hpp).
 * Built by oracle/Makefile into oracle/_build/libmppi_oracle.so.
 */
#include "oracle_core.hpp"
#include "oracle_texture.hpp"
#include "oracle_models.hpp"
#include "oracle_rng.hpp"
#include "oracle_colored.hpp"
#include "oracle_rmppi.hpp"

#include <chrono>

#if defined(_OPENMP)
#include <omp.h>
#endif

using namespace oracle;

extern "C" {

/* ----- handles ----- */
void* oracle_create(const char* model, int K, int T, int D, float dt, float lambda, float alpha, int num_iters)
{
  auto* c = new Controller();
  if (!makeModel(model, c->dyn, c->cost))
  {
    delete c;
    return nullptr;
  }
  c->dt = dt;
  c->lambda = lambda;
  c->alpha = alpha;
  c->num_iters = num_iters;
  c->init(K, T, D);
  return c;
}
void oracle_destroy(void* h)
{
  delete (Controller*)h;
}
void oracle_dims(void* h, int* S, int* C, int* O)
{
  auto* c = (Controller*)h;
  *S = c->dyn->S;
  *C = c->dyn->C;
  *O = c->dyn->O;
}
int oracle_set_dynamics_params(void* h, const void* pod, size_t n)
{
  return ((Controller*)h)->dyn->setParams(pod, n);
}
int oracle_set_cost_params(void* h, const void* pod, size_t n)
{
  return ((Controller*)h)->cost->setParams(pod, n);
}
int oracle_set_blob(void* h, const char* name, const float* data, size_t count, const int* dims, int ndims)
{
  auto* c = (Controller*)h;
  const std::string n(name);
  if (n == "dynamics_weights")
  {
    auto* m = dynamic_cast<ARNeuralNetModel*>(c->dyn.get());
    return m ? m->setWeights(data, count) : -1;
  }
  if (n.rfind("mean_lstm", 0) == 0 || n.rfind("unc_lstm", 0) == 0)
  {
    auto* m = dynamic_cast<RacerDubinsElevationLSTMUncertainty*>(c->dyn.get());
    if (!m)
      return -1;
    if (n == "mean_lstm_structure" || n == "unc_lstm_structure")
      return m->setNetworkStructure(n.rfind("mean", 0) == 0 ? 1 : 2, data, count);
    LSTM& net = (n.rfind("mean", 0) == 0) ? m->mean_net : m->unc_net;
    if (n == "mean_lstm_state" || n == "unc_lstm_state")
    {
      if (count != (size_t)2 * net.H)
        return -1;
      std::copy(data, data + count, net.w.end() - 2 * net.H);
      return 0;
    }
    std::vector<float>& dst = (n == "mean_lstm_weights" || n == "unc_lstm_weights") ? net.w : net.out_net.theta;
    if (count != dst.size())
      return -1;
    std::copy(data, data + count, dst.begin());
    return 0;
  }
  if (n == "normals_map" || n == "normals_map_transform")
  {
    auto* m = dynamic_cast<RacerDubinsElevationSuspension*>(c->dyn.get());
    if (!m)
      return -1;
    if (n == "normals_map")
      return (ndims == 3 && dims[2] == 4) ? m->setNormals(data, dims[0], dims[1]) : -1;
    return m->setNormalsTransform(data, count);
  }
  if (n == "lstm_structure")
  {
    auto* m = dynamic_cast<RacerDubinsElevationLSTMSteering*>(c->dyn.get());
    return m ? m->setStructure(data, count) : -1;
  }
  if (n == "lstm_weights" || n == "lstm_output_weights")
  {
    LSTM* net = nullptr;
    if (auto* m = dynamic_cast<BicycleSlipLSTM*>(c->dyn.get()))
      net = &m->net;
    if (auto* m = dynamic_cast<RacerDubinsElevationLSTMSteering*>(c->dyn.get()))
      net = &m->net;
    if (!net)
      return -1;
    std::vector<float>& dst = (n == "lstm_weights") ? net->w : net->out_net.theta;
    if (count != dst.size())
      return -1;
    std::copy(data, data + count, dst.begin());
    return 0;
  }
  if (n == "elevation_map" || n == "elevation_map_transform")
  {
    auto* m = dynamic_cast<RacerDubinsElevation*>(c->dyn.get());
    if (!m)
      return -1;
    if (n == "elevation_map")
      return ndims == 2 ? m->setMap(data, dims[0], dims[1]) : -1;
    return m->setMapTransform(data, count);
  }
  if (n == "costmap")
  {
    auto* m = dynamic_cast<ARStandardCost*>(c->cost.get());
    return (m && ndims == 2) ? m->setCostmap(data, dims[0], dims[1]) : -1;
  }
  return -1;
}
/** one FNN forward on the host: layers[nl], theta blob, in -> out (known-answer tests) */
void oracle_fnn_forward2(const int* layers, int nl, const float* theta, const float* in, float* out, int split_output_sum)
{
  FNN net;
  net.setStructure(std::vector<int>(layers, layers + nl));
  net.split_output_sum = split_output_sum != 0;
  std::copy(theta, theta + net.numParams(), net.theta.begin());
  net.forward(in, out);
}
void oracle_fnn_forward(const int* layers, int nl, const float* theta, const float* in, float* out)
{
  oracle_fnn_forward2(layers, nl, theta, in, out, 0);  // the reference's order (fnn_helper.cu:458-462)
}
/** the output layer's summation order of the handle's network model (FNN::split_output_sum): 1 = what the AutoRally and
 *  bicycle-LSTM models evaluate (default for them), 0 = the reference's single chain — for the test that bounds the difference */
int oracle_set_split_output_sum(void* h, int on)
{
  auto* c = (Controller*)h;
  if (auto* m = dynamic_cast<ARNeuralNetModel*>(c->dyn.get()))
  {
    m->net.split_output_sum = on != 0;
    return 0;
  }
  if (auto* m = dynamic_cast<BicycleSlipLSTM*>(c->dyn.get()))
  {
    m->net.out_net.split_output_sum = on != 0;
    return 0;
  }
  return -1;
}
/** `steps` LSTM forwards on the host from (h0, c0) of the blob with a constant input (known-answer tests):
 *  out [steps][output dim] */
void oracle_lstm_forward2(int input_dim, int hidden_dim, const int* out_layers, int nl, const float* lstm_blob,
                          const float* fnn_blob, const float* in, int steps, float* out, int split_output_sum);
void oracle_lstm_forward(int input_dim, int hidden_dim, const int* out_layers, int nl, const float* lstm_blob,
                         const float* fnn_blob, const float* in, int steps, float* out)
{
  oracle_lstm_forward2(input_dim, hidden_dim, out_layers, nl, lstm_blob, fnn_blob, in, steps, out, 0);
}
void oracle_lstm_forward2(int input_dim, int hidden_dim, const int* out_layers, int nl, const float* lstm_blob,
                          const float* fnn_blob, const float* in, int steps, float* out, int split_output_sum)
{
  LSTM net;
  net.setStructure(input_dim, hidden_dim, std::vector<int>(out_layers, out_layers + nl));
  net.out_net.split_output_sum = split_output_sum != 0;
  std::copy(lstm_blob, lstm_blob + net.numParams(), net.w.begin());
  std::copy(fnn_blob, fnn_blob + net.out_net.numParams(), net.out_net.theta.begin());
  std::vector<float> h(net.h0(), net.h0() + hidden_dim), c(net.c0(), net.c0() + hidden_dim);
  const int od = out_layers[nl - 1];
  for (int t = 0; t < steps; t++)
    net.forward(in + (size_t)t * input_dim, h.data(), c.data(), out + (size_t)t * od);
}
/** xdot = f(x, u) of the handle's model (kinematics + dynamics), for plugin-level known-answer tests */
void oracle_state_deriv(void* h, const float* x, const float* u, float* xdot)
{
  auto* c = (Controller*)h;
  std::vector<float> th(std::max(1, c->dyn->scratchFloats()), 0.0f);
  for (int i = 0; i < c->dyn->S; i++)
    xdot[i] = 0.0f;
  std::vector<float> y0(c->dyn->O, 0.0f);
  c->dyn->initializeDynamics(x, u, y0.data(), th.data(), 0.0f, 0.0f);
  c->dyn->computeKinematics(x, xdot);
  c->dyn->computeDynamics(x, u, xdot, th.data());
}
/** Dynamics::updateState with a given derivative (the reference's TestUpdateState known answers) */
void oracle_update_state(void* h, const float* x, const float* xdot, float dt, float* x_next)
{
  auto* c = (Controller*)h;
  for (int i = 0; i < c->dyn->S; i++)
    x_next[i] = 0.0f;
  c->dyn->updateState(x, x_next, xdot, dt);
}
/** running state cost of one output vector */
float oracle_state_cost(void* h, const float* y, int t, int* crash)
{
  return ((Controller*)h)->cost->computeStateCost(y, t, crash);
}
/** the individual terms of ARStandardCost (pinned on tests/cost_functions/autorally_standard_cost_test.cu):
 *  which = 0 speed, 1 stabilizing, 2 track, 3 crash; returns NaN if the handle's cost is not ARStandardCost */
float oracle_ar_cost_term(void* h, int which, const float* s, int* crash)
{
  auto* c = dynamic_cast<ARStandardCost*>(((Controller*)h)->cost.get());
  if (!c)
    return NAN;
  switch (which)
  {
    case 0: return c->getSpeedCost(s);
    case 1: return c->getStabilizingCost(s, crash);
    case 2: return c->getTrackCost(s, crash);
    case 3: return c->getCrashCost(crash);
  }
  return NAN;
}
int oracle_ar_coor_transform(void* h, float x, float y, float* uvw)
{
  auto* c = dynamic_cast<ARStandardCost*>(((Controller*)h)->cost.get());
  if (!c)
    return -1;
  c->coorTransform(x, y, uvw, uvw + 1, uvw + 2);
  return 0;
}
void oracle_set_colored_mppi_params(void* h, float gamma, float r_exp, const float* leash_dist, int leash_active, int leash_jump)
{
  auto* c = (Controller*)h;
  c->tsallis_gamma = gamma;
  c->tsallis_r = r_exp;
  c->leash_active = leash_active != 0;
  c->leash_jump = leash_jump;
  c->leash_dist.assign(c->dyn->S, 0.0f);
  if (leash_dist)
    c->leash_dist.assign(leash_dist, leash_dist + c->dyn->S);
}
void oracle_set_control_ranges(void* h, const float* lo_hi)
{
  auto* c = (Controller*)h;
  for (int i = 0; i < c->dyn->C; i++)
  {
    c->dyn->rng_lo[i] = lo_hi[2 * i];
    c->dyn->rng_hi[i] = lo_hi[2 * i + 1];
  }
}
void oracle_set_control_deadband(void* h, const float* db)
{
  auto* c = (Controller*)h;
  for (int i = 0; i < c->dyn->C; i++)
    c->dyn->deadband[i] = db[i];
}
void oracle_set_sampler(void* h, const float* std_dev /*[D][C]*/, const float* control_cost_coeff /*[C]*/,
                        float pure_noise_pct, float std_dev_decay, int sum_strides)
{
  auto* c = (Controller*)h;
  for (size_t i = 0; i < c->smp.std_dev.size(); i++)
    c->smp.std_dev[i] = std_dev[i];
  for (int i = 0; i < c->dyn->C; i++)
    c->smp.control_cost_coeff[i] = control_cost_coeff[i];
  c->smp.pure_noise_trajectories_percentage = pure_noise_pct;
  c->smp.std_dev_decay = std_dev_decay;
  c->smp.sum_strides = sum_strides;
}
void oracle_set_independent_noise(void* h, int independent)
{
  ((Controller*)h)->smp.independent_noise = independent != 0;
}
void oracle_set_time_specific_std_dev(void* h, const float* std_dev /*[D][T][C] or NULL*/)
{
  auto* c = (Controller*)h;
  if (std_dev)
    c->smp.std_dev_time.assign(std_dev, std_dev + (size_t)c->smp.D * c->smp.T * c->smp.C);
  else
    c->smp.std_dev_time.clear();
}
void oracle_set_controller_params(void* h, float nominal_threshold, const float* slide_scale)
{
  auto* c = (Controller*)h;
  c->nominal_threshold = nominal_threshold;
  if (slide_scale)
    for (int i = 0; i < c->dyn->C; i++)
      c->slide_scale[i] = slide_scale[i];
}

/* ----- kernel-level pieces ----- */
void oracle_set_gaussian_controls(void* h, const float* mean, const float* eps, int stride, int iter, float* v)
{
  ((Controller*)h)->smp.setGaussianControls(mean, eps, stride, iter, v);
}
/** v in/out [D][K][T][C]; costs out [D][K]; OpenMP over rollouts when threads > 1 (each rollout is independent) */
void oracle_rollout_costs(void* h, const float* x0, const float* mean, float* v, float* costs, int threads)
{
  auto* c = (Controller*)h;
  if (threads <= 1)
  {
    rolloutCosts(*c->dyn, *c->cost, c->smp, c->dt, c->lambda, c->alpha, x0, mean, v, costs);
    return;
  }
#if defined(_OPENMP)
  const int K = c->K;
  const int chunk = 64;
#pragma omp parallel for schedule(dynamic) num_threads(threads)
  for (int k0 = 0; k0 < K; k0 += chunk)
  {
    rolloutCosts(*c->dyn, *c->cost, c->smp, c->dt, c->lambda, c->alpha, x0, mean, v, costs, k0, std::min(K, k0 + chunk));
  }
#else
  rolloutCosts(*c->dyn, *c->cost, c->smp, c->dt, c->lambda, c->alpha, x0, mean, v, costs);
#endif
}
int oracle_best_index(const float* costs, int K)
{
  return computeBestIndex(costs, K);
}
float oracle_baseline(const float* costs, int K)
{
  return computeBaselineCost(costs, K);
}
void oracle_norm_exp(float* costs, int K, float lambda_inv, float baseline)
{
  normExpTransform(costs, K, lambda_inv, baseline);
}
float oracle_normalizer(const float* w, int K)
{
  return computeNormalizer(w, K);
}
void oracle_free_energy(const float* w, int K, float baseline, float lambda, float* out3)
{
  computeFreeEnergy(out3[0], out3[1], out3[2], w, K, baseline, lambda);
}
void oracle_weighted_reduction(const float* w, const float* v, float normalizer, int K, int T, int C, int sum_stride,
                               float* u_out)
{
  weightedReduction(w, v, normalizer, K, T, C, sum_stride, u_out);
}
/** process-wide flavour of `inter += weight * v` in every weightedReduction: 0 = multiply, then add (default); 1 = one fma */
void oracle_set_reduction_fma(int fma)
{
  weightedReductionFma() = fma != 0;
}
void oracle_smooth(float* u, const float* history, int T, int C)
{
  smoothControlTrajectory(u, history, T, C);
}
void oracle_slide(float* u, int T, int C, int steps, const float* zero_control, const float* slide_scale)
{
  slideControlSequence(u, T, C, steps, zero_control, slide_scale);
}
void oracle_save_history(int steps, const float* u, float* history, int C)
{
  saveControlHistory(steps, u, history, C);
}
void oracle_state_trajectory(void* h, const float* x0, const float* u, float* result)
{
  auto* c = (Controller*)h;
  computeStateTrajectory(*c->dyn, c->dt, x0, u, c->T, result);
}
void oracle_output_trajectory(void* h, const float* x0, const float* u, float* state_result, float* output_result)
{
  auto* c = (Controller*)h;
  computeStateTrajectory(*c->dyn, c->dt, x0, u, c->T, state_result, output_result);
}
/** one model step on the host (used for closed-loop tests: examples/cartpole_example.cu:63-85) */
void oracle_model_step(void* h, float* x, float* u, float dt)
{
  auto* c = (Controller*)h;
  const int S = c->dyn->S, O = c->dyn->O;
  std::vector<float> xn(S), xdot(S, 0.0f), y(O, 0.0f), th(std::max(1, c->dyn->scratchFloats()), 0.0f);
  c->dyn->initializeDynamics(x, u, y.data(), th.data(), 0.0f, dt);
  c->dyn->enforceConstraints(x, u);
  c->dyn->step(x, xn.data(), xdot.data(), u, y.data(), th.data(), 0, dt);
  for (int i = 0; i < S; i++)
    x[i] = xn[i];
}

/** Dynamics::enforceLeash of the model (base rule, or the RACER body-frame rule) */
void oracle_enforce_leash(void* h, const float* x_true, const float* x_nominal, const float* leash, float* out)
{
  ((Controller*)h)->dyn->enforceLeash(x_true, x_nominal, leash, out);
}

/** one step with everything it produces: next state, the derivative entries the model writes, the output */
void oracle_model_step_full(void* h, const float* x_in, const float* u_in, float dt, float* xn, float* xdot, float* y)
{
  auto* c = (Controller*)h;
  const int S = c->dyn->S, Cd = c->dyn->C, O = c->dyn->O;
  std::vector<float> x(x_in, x_in + S), u(u_in, u_in + Cd), th(std::max(1, c->dyn->scratchFloats()), 0.0f);
  std::fill(xdot, xdot + S, 0.0f);
  std::fill(y, y + O, 0.0f);
  c->dyn->initializeDynamics(x.data(), u.data(), y, th.data(), 0.0f, dt);
  c->dyn->step(x.data(), xn, xdot, u.data(), y, th.data(), 0, dt);
}

/* ----- controller level ----- */
void oracle_set_nominal_control(void* h, const float* u)
{
  auto* c = (Controller*)h;
  std::copy(u, u + c->control.size(), c->control.begin());
  if (c->D == 2) /* Tube: nominal_control_trajectory_ too; RMPPI: nominal_control_trajectory_ = init_control_traj */
    std::copy(u, u + c->control.size(), c->nominal_control.begin());
}
void oracle_iterate(void* h, const float* x0, const float* mean, const float* eps, int stride, int iter, float* u_new)
{
  ((Controller*)h)->iterate(x0, mean, eps, stride, iter, u_new);
}
void oracle_vanilla_compute_control(void* h, const float* x0, int stride, const float* eps)
{
  ((Controller*)h)->vanillaComputeControl(x0, stride, eps);
}
void oracle_tube_compute_control(void* h, const float* x0, int stride, const float* eps)
{
  ((Controller*)h)->tubeComputeControl(x0, stride, eps);
}
void oracle_vanilla_slide(void* h, int steps)
{
  ((Controller*)h)->vanillaSlide(steps);
}
void oracle_tube_slide(void* h, int steps)
{
  ((Controller*)h)->tubeSlide(steps);
}
void oracle_get_control(void* h, float* u)
{
  auto* c = (Controller*)h;
  std::copy(c->control.begin(), c->control.end(), u);
}
void oracle_get_nominal_control(void* h, float* u)
{
  auto* c = (Controller*)h;
  std::copy(c->nominal_control.begin(), c->nominal_control.end(), u);
}
void oracle_get_state_traj(void* h, float* x)
{
  auto* c = (Controller*)h;
  std::copy(c->state_traj.begin(), c->state_traj.end(), x);
}
void oracle_get_nominal_state_traj(void* h, float* x)
{
  auto* c = (Controller*)h;
  std::copy(c->nominal_state.begin(), c->nominal_state.end(), x);
}
void oracle_get_costs(void* h, float* costs)
{
  auto* c = (Controller*)h;
  std::copy(c->costs.begin(), c->costs.end(), costs);
}
void oracle_get_weights(void* h, float* w)
{
  auto* c = (Controller*)h;
  std::copy(c->w.begin(), c->w.end(), w);
}
void oracle_get_samples(void* h, float* v)
{
  auto* c = (Controller*)h;
  std::copy(c->v.begin(), c->v.end(), v);
}
/** out: baseline[2], normalizer[2], fe[2], fe_var[2], fe_mod[2], nominal_state_used */
void oracle_get_stats(void* h, float* out11)
{
  auto* c = (Controller*)h;
  for (int d = 0; d < 2; d++)
  {
    out11[d] = c->stats.baseline[d];
    out11[2 + d] = c->stats.normalizer[d];
    out11[4 + d] = c->stats.free_energy[d];
    out11[6 + d] = c->stats.free_energy_var[d];
    out11[8 + d] = c->stats.free_energy_mod[d];
  }
  out11[10] = (float)c->stats.nominal_state_used;
}

/* ----- colored noise ----- */
static ColoredNoiseParams coloredParams(int C, const float* exponents, float decay, float fmin)
{
  ColoredNoiseParams p;
  p.exponents.assign(exponents, exponents + C);
  p.offset_decay_rate = decay;
  p.fmin = fmin;
  return p;
}
/** flavour 0: definition (double inverse DFT), 1: the engine's arithmetic for this (T, offset) — radix-4 butterfly + quarter-size
 *  GEMM where T is a multiple of 4, else the dense folded table —, 2: the dense folded table + fp32 fma chains for any T */
void oracle_colored_noise(int flavour, int K, int T, int C, const float* exponents, float decay, float fmin, int offset_t,
                          const float* z, float* eps)
{
  const ColoredNoiseParams p = coloredParams(C, exponents, decay, fmin);
  if (flavour == 0)
    coloredNoiseDefinition(K, T, C, p, offset_t, z, eps);
  else if (flavour == 2)
    coloredNoiseGemm(K, T, C, p, offset_t, z, eps);
  else
    coloredNoiseEngine(K, T, C, p, offset_t, z, eps);
}
void oracle_colored_weights(int T, int C, const float* exponents, float fmin, float* w /*[C][T+1]*/, float* sigma /*[C]*/)
{
  std::vector<float> ww, ss;
  coloredWeights(T, C, coloredParams(C, exponents, 0.0f, fmin), ww, ss);
  std::copy(ww.begin(), ww.end(), w);
  std::copy(ss.begin(), ss.end(), sigma);
}
void oracle_philox_spectrum(uint64_t seed, uint32_t generation, int T, int C, int k_begin, int k_end, float* z)
{
  philoxSpectrum(seed, generation, T, C, k_begin, k_end, z);
}
/** ColoredMPPI computeControl; z: [num_iters][K][C][T+1][2] Gaussian spectrum */
void oracle_colored_compute_control(void* h, const float* x0, int stride, const float* z, const float* exponents,
                                    float decay, float fmin)
{
  auto* c = (Controller*)h;
  const int C = c->dyn->C, K = c->K, T = c->T;
  const ColoredNoiseParams p = coloredParams(C, exponents, decay, fmin);
  std::vector<float> eps((size_t)c->num_iters * K * T * C);
  for (int it = 0; it < c->num_iters; it++)
    coloredNoiseEngine(K, T, C, p, stride, z + (size_t)it * K * C * 2 * (T + 1), &eps[(size_t)it * K * T * C]);
  c->coloredComputeControl(x0, stride, eps.data());
}

/* ----- Robust MPPI ----- */
void* oracle_rmppi_create(void* h)
{
  auto* r = new RobustController();
  r->init((Controller*)h);
  return r;
}
void oracle_rmppi_destroy(void* r)
{
  delete (RobustController*)r;
}
void oracle_rmppi_set_params(void* r, float value_function_threshold, int num_candidates, int samples_per_candidate)
{
  auto* rc = (RobustController*)r;
  rc->value_function_threshold = value_function_threshold;
  rc->num_candidates = num_candidates;
  rc->samples_per_candidate = samples_per_candidate;
}
void oracle_rmppi_set_gains(void* r, const float* gains, int accumulate_all_states)
{
  auto* rc = (RobustController*)r;
  std::copy(gains, gains + rc->fb.fb_gain_traj.size(), rc->fb.fb_gain_traj.begin());
  rc->fb.accumulate_all_states = accumulate_all_states != 0;
}
void oracle_rmppi_feedback(void* r, const float* x_act, const float* x_goal, int t, float* out)
{
  auto* rc = (RobustController*)r;
  for (int j = 0; j < rc->fb.C; j++)
    out[j] = 0.0f;
  rc->fb.k(x_act, x_goal, t, out);
}
void oracle_rmppi_line_search(void* r, int stride, float* weights /*[3][nc]*/, int* strides /*[nc]*/)
{
  auto* rc = (RobustController*)r;
  rc->computeLineSearchWeights();
  rc->computeImportanceSamplerStride(stride);
  std::copy(rc->line_search_weights.begin(), rc->line_search_weights.end(), weights);
  std::copy(rc->strides.begin(), rc->strides.end(), strides);
}
void oracle_rmppi_candidates(void* r, const float* x_k, const float* x_kp1, const float* real_kp1, float* out /*[nc][S]*/)
{
  auto* rc = (RobustController*)r;
  rc->computeLineSearchWeights();
  rc->getInitNominalStateCandidates(x_k, x_kp1, real_kp1);
  std::copy(rc->candidate_states.begin(), rc->candidate_states.end(), out);
}
int oracle_rmppi_best_index(void* r, const float* candidate_costs, float* free_energy /*[nc]*/)
{
  auto* rc = (RobustController*)r;
  rc->candidate_costs.assign(candidate_costs, candidate_costs + (size_t)rc->num_candidates * rc->samples_per_candidate);
  rc->computeBestIndex();
  std::copy(rc->candidate_free_energy.begin(), rc->candidate_free_energy.end(), free_energy);
  return rc->best_index;
}
/** x0 [2][S], mean [2][T][C], v [2][K][T][C] in/out, costs [2][K] */
void oracle_rmppi_rollout_costs(void* r, const float* x0, const float* mean, float* v, float* costs)
{
  auto* rc = (RobustController*)r;
  Controller* c = rc->c;
  rmppiRolloutCosts(*c->dyn, *c->cost, c->smp, rc->fb, c->dt, c->lambda, c->alpha, rc->value_function_threshold, x0, mean,
                    v, costs);
}
void oracle_rmppi_update_importance_sampling(void* r, const float* state, int stride, const float* eps)
{
  ((RobustController*)r)->updateImportanceSamplingControl(state, stride, eps);
}
void oracle_rmppi_compute_control(void* r, const float* state, int stride, const float* eps)
{
  ((RobustController*)r)->computeControl(state, stride, eps);
}
/** out: nominal_state [S], then best_index, nominal_stride; fe: candidate free energy [nc]; costs: [nc * ns] (may be null) */
void oracle_rmppi_get_state(void* r, float* nominal_state, int* best_and_stride, float* fe, float* cand_costs)
{
  auto* rc = (RobustController*)r;
  std::copy(rc->nominal_state.begin(), rc->nominal_state.end(), nominal_state);
  best_and_stride[0] = rc->best_index;
  best_and_stride[1] = rc->nominal_stride;
  for (size_t i = 0; i < rc->candidate_free_energy.size(); i++)
    fe[i] = rc->candidate_free_energy[i];
  if (cand_costs)
    std::copy(rc->candidate_costs.begin(), rc->candidate_costs.end(), cand_costs);
}

/* ----- RNG ----- */
void oracle_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out)
{
  philox4x32_10(ctr, key, out);
}
void oracle_philox_normal(uint64_t seed, uint32_t generation, uint32_t stream, int K, int T, int C, int k_begin,
                          int k_end, float* eps)
{
  philoxNormal(seed, generation, stream, K, T, C, k_begin, k_end, eps);
}

/* ----- det_math elementwise (for the host-vs-GPU bit-parity test) ----- */
void oracle_det_eval(int func, const float* x, float* y, int n)
{
  for (int i = 0; i < n; i++)
  {
    switch (func)
    {
      case 0: y[i] = det::sin(x[i]); break;
      case 1: y[i] = det::cos(x[i]); break;
      case 2: y[i] = det::exp(x[i]); break;
      case 3: y[i] = det::log(x[i]); break;
      case 4: y[i] = det::tanh(x[i]); break;
      case 5: y[i] = det::atan(x[i]); break;
      case 6: y[i] = det::normalizeAngle(x[i]); break;
      case 7: y[i] = det::sigmoid(x[i]); break;
      case 8: y[i] = det::sqrt(x[i]); break;
      case 9: y[i] = 1.0f / x[i]; break;
      case 10: y[i] = det::tanh(x[i]); break; /* device side: the packed tanh2() path */
      case 11:
        y[i] = det::sigmoid(x[i]) + det::sigmoid(x[i] * 0.5f) + det::sigmoid(-x[i]) + det::sigmoid(x[i] + 1.0f);
        break;
      case 12: y[i] = det::tan(x[i]); break;
      case 13: y[i] = det::asin(x[i]); break;
      default: y[i] = 0.0f;
    }
  }
}

/* ----- 2-D texture helper lookups (oracle_texture.hpp); params laid out as mppi_texture2d_params ----- */
void oracle_texture2d_query(const float* data, int width, int height, int channels, const int* address_mode, int filter_mode,
                            const float* border_color, const float* origin, const float* rotations, const float* resolution,
                            const float* points, int n, int frame, float* out)
{
  oracle::Texture2D t;
  t.values = data;
  t.width = width;
  t.height = height;
  t.channels = channels;
  t.address_mode[0] = address_mode[0];
  t.address_mode[1] = address_mode[1];
  t.filter_mode = filter_mode;
  for (int i = 0; i < 4; i++)
    t.border_color[i] = border_color[i];
  for (int i = 0; i < 3; i++)
  {
    t.origin[i] = origin[i];
    t.resolution[i] = resolution[i];
    for (int j = 0; j < 3; j++)
      t.rot[i][j] = rotations[3 * i + j];
  }
  for (int i = 0; i < n; i++)
  {
    float map[3], tex[3];
    const float* pt = points + 3 * i;
    if (frame == 2)
    {
      t.worldToMap(pt, map);
      t.mapToTex(map, tex);
    }
    else if (frame == 1)
      t.mapToTex(pt, tex);
    else
      for (int k = 0; k < 3; k++)
        tex[k] = pt[k];
    t.query(tex, out + (size_t)i * channels);
  }
}

/* ----- CPU baseline timing: one optimisation-loop body (bench.py cpu_baseline leg) ----- */
/** returns seconds for `iters` passes of iterate() on the given inputs using `threads` host threads for the rollout */
double oracle_time_iterations(void* h, const float* x0, const float* mean, const float* eps, int iters, int threads)
{
  auto* c = (Controller*)h;
  const int C = c->dyn->C;
  std::vector<float> u_new((size_t)c->D * c->T * C);
  auto t0 = std::chrono::steady_clock::now();
  for (int it = 0; it < iters; it++)
  {
    c->smp.setGaussianControls(mean, eps, 1, 0, c->v.data());
    oracle_rollout_costs(h, x0, mean, c->v.data(), c->costs.data(), threads);
    c->w = c->costs;
    for (int d = 0; d < c->D; d++)
    {
      float* wd = &c->w[(size_t)d * c->K];
      const float b = computeBaselineCost(wd, c->K);
      normExpTransform(wd, c->K, (float)(1.0 / c->lambda), b);
      const float eta = computeNormalizer(wd, c->K);
      weightedReduction(wd, &c->v[(size_t)d * c->K * c->T * C], eta, c->K, c->T, C, c->smp.sum_strides,
                        &u_new[(size_t)d * c->T * C]);
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

int oracle_max_threads()
{
#if defined(_OPENMP)
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
