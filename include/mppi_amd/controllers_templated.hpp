/**
 * controllers_templated.hpp — the reference's TEMPLATED controller classes over this engine.
 *
 * A user of ACDSLab/MPPI-Generic writes (examples/cartpole_example.cu:46-52)
 *
 *     auto controller = new VanillaMPPIController<CartpoleDynamics, CartpoleQuadraticCost, DDPFeedback<CartpoleDynamics, 100>,
 *                                                 100, 2048>(model, cost, fb_controller, sampler, dt, max_iter, lambda, alpha);
 *
 * (controllers/MPPI/mppi_controller.cuh:14-40, Tube-MPPI/tube_mppi_controller.cuh:20-60, R-MPPI/robust_mppi_controller.cuh:
 * 60-110, ColoredMPPI/colored_mppi_controller.cuh:20-60).  The classes below keep that spelling: same template parameters,
 * same constructor argument lists, same method names.  What happens underneath is this engine's way of doing it:
 *   - the class INSTANTIATES mppi::engine::ModelT<DYN_T, COST_T, SAMPLING_T> — i.e. every rollout / post-processing kernel
 *     for the user's plugin types — in the user's translation unit (compile it with hipcc, as the reference's is compiled
 *     with nvcc) and registers it with libmppi_amd.so under a name derived from the types;
 *   - the controller itself is a handle of the C ABI (include/mppi_amd.h) created for that name, on first use, so that
 *     setParams() with another rollout block shape can still be honoured;
 *   - the user's model / cost / sampler OBJECTS stay the source of truth: their parameter structs, control ranges and
 *     dead bands are pushed to the engine whenever they differ from what was pushed last (the reference's objects copy
 *     themselves to the device on setParams; here the engine's copy travels with every launch as a kernel argument).
 * Host vectors are mppi::host::Array / Matrix (plugin/host_arrays.hpp) where the reference uses Eigen types of the same
 * layout.  Errors throw mppi_amd::Error (controllers.hpp).
 */
#ifndef MPPI_AMD_CONTROLLERS_TEMPLATED_HPP_
#define MPPI_AMD_CONTROLLERS_TEMPLATED_HPP_

#include <cstring>
#include <memory>
#include <string>
#include <typeinfo>
#include <vector>

#include "mppi_amd.h"
#include "mppi_amd/controllers.hpp"
#include "mppi_amd/engine/model_registry.hpp"
#include "mppi_amd/feedback_controllers/ddp_feedback.hpp"
#include "mppi_amd/plugin/dynamics_host.hpp"
#include "mppi_amd/sampling_distributions/colored_noise.hpp"
#include "mppi_amd/sampling_distributions/gaussian.hpp"

/** reference: controllers/controller.cuh:30-60 (ControllerParams): the fields a caller sets before the first computeControl */
template <int S_DIM, int C_DIM, int MAX_TIMESTEPS>
struct ControllerParams
{
  float dt_ = 0.0f;
  float lambda_ = 1.0f;
  float alpha_ = 0.0f;
  int num_iters_ = 1;
  int num_timesteps_ = MAX_TIMESTEPS;
  int seed_ = 42;
  /** (rollouts per block, lanes per rollout, systems per launch): the first two select the engine's block shape */
  dim3 dynamics_rollout_dim_ = dim3(0, 0, 1);
  dim3 cost_rollout_dim_ = dim3(0, 0, 1);  ///< accepted, unused: sampling, dynamics and cost share one kernel here
  mppi::host::Array<C_DIM> slide_control_scale_ = mppi::host::Array<C_DIM>::Zero();
};

namespace mppi_amd
{
namespace templated
{
/**
 * The rollout block shapes a controller type is instantiated for — (rollouts per block, lanes per rollout, systems per launch),
 * the reference's dynamics_rollout_dim_ with z chosen by the controller kind (Tube / Robust: 2).  Default: one lane per rollout
 * (64 or 32 rollouts per block: what the analytic in-tree models run on) and the reference's own example shape (64, 4, .)
 * (examples/cartpole_example.cu:50-51) on the LDS + barrier form of the plugin contract.  A caller that sets another
 * dynamics_rollout_dim_ specialises this for its types (every shape is one more set of kernels in the caller's translation
 * unit); a dynamics_rollout_dim_ that is not in the list makes the controller THROW MPPI_ERR_LAUNCH_SHAPE naming the list — it
 * is never replaced by another shape silently.
 */
template <class DYN_T, class COST_T, class SAMPLING_T>
struct RolloutShapes
{
  using type = mppi::engine::Shapes<mppi::engine::Shape<64, 1, 1>, mppi::engine::Shape<64, 1, 2>, mppi::engine::Shape<32, 1, 1>,
                                    mppi::engine::Shape<32, 1, 2>, mppi::engine::Shape<64, 4, 1>, mppi::engine::Shape<64, 4, 2>>;
};

/**
 * May the user's plugin classes run on the ROLE-SEPARATED rollout kernels?  Only when all three say so
 * (`static constexpr bool MPPI_BARRIER_FREE_STEP = true;`, plugin/parallel_utils.hpp) — the in-tree models do.  A model written
 * like the reference's, with the two block barriers of Dynamics::step (dynamics/dynamics.cu:138,140) or a __syncthreads() of
 * its own in computeDynamics, says nothing, is taken to have barriers and runs on the FUSED kernel, where every thread of the
 * block reaches every plugin call as in the reference.  (Round 5 instantiated every user model with PIPELINE = true: a
 * reference-style model hung the GPU.)
 */
template <class DYN_T, class COST_T, class SAMPLING_T>
constexpr bool rolePipelineAllowed()
{
  return mppi::barrier_free_step<DYN_T>::value && mppi::barrier_free_step<COST_T>::value &&
         mppi::barrier_free_step<SAMPLING_T>::value;
}

/** registers ModelT<DYN_T, COST_T, SAMPLING_T> once per process and returns the name it is registered under */
template <class DYN_T, class COST_T, class SAMPLING_T>
inline const std::string& registeredModelName()
{
  using namespace mppi::engine;
  using MODEL = ModelT<DYN_T, COST_T, SAMPLING_T, typename RolloutShapes<DYN_T, COST_T, SAMPLING_T>::type,
                       /*FIN_BY=*/1, void, Shapes<>, /*PIPELINE=*/rolePipelineAllowed<DYN_T, COST_T, SAMPLING_T>(),
                       /*RMPPI=*/!SAMPLING_T::COLORED>;
  static const std::string name = [] {
    // the sampler type is part of the instantiation (two controllers that differ only in their Gaussian sampler class must
    // not share a registration: the parameter layouts differ)
    std::string n = std::string("tpl:") + typeid(DYN_T).name() + ":" + typeid(COST_T).name() + ":" + typeid(SAMPLING_T).name();
    const mppi_status s =
        mppi_register_model_checked(n.c_str(), SAMPLING_T::COLORED ? MPPI_SAMPLER_COLORED : MPPI_SAMPLER_GAUSSIAN,
                                    &modelFactory<MODEL, 64, 1>, engineAbiFingerprint(), modelFlags<MODEL>());
    if (s != MPPI_OK)
      throw Error(s, "mppi_register_model failed for " + n + ": " + mppi_last_error(nullptr));
    return n;
  }();
  return name;
}

/** what the four controller classes share (reference: Controller<...>, controllers/controller.cuh) */
template <class DYN_T, class COST_T, class FB_T, int MAX_TIMESTEPS, int NUM_ROLLOUTS, class SAMPLING_T>
class ControllerBase
{
public:
  static const int STATE_DIM = DYN_T::STATE_DIM, CONTROL_DIM = DYN_T::CONTROL_DIM, OUTPUT_DIM = DYN_T::OUTPUT_DIM;
  typedef mppi::host::Array<STATE_DIM> state_array;
  typedef mppi::host::Array<CONTROL_DIM> control_array;
  typedef mppi::host::Array<OUTPUT_DIM> output_array;
  typedef mppi::host::Matrix<CONTROL_DIM, MAX_TIMESTEPS> control_trajectory;
  typedef mppi::host::Matrix<STATE_DIM, MAX_TIMESTEPS> state_trajectory;
  typedef mppi::host::Matrix<OUTPUT_DIM, MAX_TIMESTEPS> output_trajectory;
  typedef ControllerParams<STATE_DIM, CONTROL_DIM, MAX_TIMESTEPS> PARAMS_T;
  typedef DYN_T TEMPLATED_DYNAMICS;
  typedef COST_T TEMPLATED_COSTS;
  typedef FB_T TEMPLATED_FEEDBACK;
  typedef SAMPLING_T TEMPLATED_SAMPLING;

  ControllerBase(int kind, DYN_T* model, COST_T* cost, FB_T* fb_controller, SAMPLING_T* sampler, float dt, int max_iter,
                 float lambda, float alpha, int num_timesteps, const control_trajectory& init_control_traj, hipStream_t stream)
    : model_(model), cost_(cost), fb_controller_(fb_controller), sampler_(sampler), kind_(kind), stream_(stream),
      init_control_traj_(init_control_traj)
  {
    params_.dt_ = dt;
    params_.num_iters_ = max_iter;
    params_.lambda_ = lambda;
    params_.alpha_ = alpha;
    params_.num_timesteps_ = num_timesteps;
    if (num_timesteps <= 0 || num_timesteps > MAX_TIMESTEPS)
      throw Error(MPPI_ERR_INVALID_ARG, "num_timesteps must be in [1, MAX_TIMESTEPS]");
    (void)registeredModelName<DYN_T, COST_T, SAMPLING_T>();  // fail early if the library does not take the model
  }
  virtual ~ControllerBase()
  {
    if (h_)
      mppi_destroy(h_);
  }
  ControllerBase(const ControllerBase&) = delete;
  ControllerBase& operator=(const ControllerBase&) = delete;

  virtual std::string getControllerName() const = 0;

  /* ---- parameters ---- */
  PARAMS_T getParams() const
  {
    return params_;
  }
  /** before the first computeControl everything takes effect; afterwards lambda / alpha / num_iters / the slide scale do
   *  (the block shape and the horizon are properties of the engine handle) */
  void setParams(const PARAMS_T& p)
  {
    const bool shape_changed = p.dynamics_rollout_dim_.x != params_.dynamics_rollout_dim_.x ||
                               p.dynamics_rollout_dim_.y != params_.dynamics_rollout_dim_.y ||
                               p.num_timesteps_ != params_.num_timesteps_ || p.dt_ != params_.dt_ || p.seed_ != params_.seed_;
    if (h_ && shape_changed)
      throw Error(MPPI_ERR_STATE, "rollout block shape, horizon, dt and seed are fixed once the controller has run");
    params_ = p;
    if (h_)
      pushControllerParams();
  }
  void setLambda(float lambda)
  {
    params_.lambda_ = lambda;
    if (h_)
      pushControllerParams();
  }
  void setAlpha(float alpha)
  {
    params_.alpha_ = alpha;
    if (h_)
      pushControllerParams();
  }
  void setNumIters(int n)
  {
    params_.num_iters_ = n;
    if (h_)
      pushControllerParams();
  }
  /** mppi_reduction_mode (mppi_amd.h): the reference's own summation order for the last stage of an iteration, for parity runs */
  void setReductionMode(int mode)
  {
    ensureHandle();
    check(mppi_set_reduction_mode(h_, mode));
  }
  float getLambda() const
  {
    return params_.lambda_;
  }
  float getAlpha() const
  {
    return params_.alpha_;
  }
  float getDt() const
  {
    return params_.dt_;
  }
  int getNumIters() const
  {
    return params_.num_iters_;
  }
  int getNumTimesteps() const
  {
    return params_.num_timesteps_;
  }
  void setSeedCUDARandomNumberGen(unsigned seed)
  {
    params_.seed_ = (int)seed;
    if (h_)
      check(mppi_set_seed(h_, seed));
  }

  /* ---- control loop (controllers/controller.cuh:300-400) ---- */
  virtual void computeControl(const state_array& state, int optimization_stride = 1)
  {
    ensureHandle();
    syncPlugins();
    check(mppi_compute_control(h_, state.data(), optimization_stride));
  }
  void updateImportanceSampler(const control_trajectory& nominal_control)
  {
    ensureHandle();
    check(mppi_set_nominal_control(h_, nominal_control.data()));
  }
  control_trajectory getControlSeq() const
  {
    control_trajectory u = control_trajectory::Zero();
    if (h_)
      check(mppi_get_control_seq(h_, u.data()));
    return u;
  }
  state_trajectory getTargetStateSeq() const
  {
    state_trajectory x = state_trajectory::Zero();
    if (h_)
      check(mppi_get_state_seq(h_, x.data()));
    return x;
  }
  output_trajectory getTargetOutputSeq() const
  {
    output_trajectory y = output_trajectory::Zero();
    if (h_)
      check(mppi_get_output_seq(h_, y.data()));
    return y;
  }
  void slideControlSequence(int steps)
  {
    ensureHandle();
    check(mppi_slide(h_, steps));
  }
  mppi_stats getFreeEnergyStatistics() const
  {
    mppi_stats s{};
    if (h_)
      check(mppi_get_stats(h_, &s));
    return s;
  }
  float getBaselineCost() const
  {
    return getFreeEnergyStatistics().real_sys.baseline;
  }
  float getNormalizerCost() const
  {
    return getFreeEnergyStatistics().real_sys.normalizer;
  }
  std::vector<float> getSampledCostSeq() const
  {
    std::vector<float> c((size_t)(kind_ == MPPI_CONTROLLER_TUBE || kind_ == MPPI_CONTROLLER_ROBUST ? 2 : 1) * NUM_ROLLOUTS);
    if (h_)
      check(mppi_get_costs(h_, c.data()));
    return c;
  }
  /** reference: controller.cuh:329-345 — interpolated feed-forward control, clamped by the model */
  control_array getCurrentControl(const state_array& state, double rel_time) const
  {
    const control_trajectory u = getControlSeq();
    // a relative time before the sequence or at / beyond its end holds the first / last control (no read outside it)
    int lower = (int)(rel_time / params_.dt_);
    lower = lower < 0 ? 0 : (lower > params_.num_timesteps_ - 1 ? params_.num_timesteps_ - 1 : lower);
    double a = (rel_time - lower * (double)params_.dt_) / params_.dt_;
    a = a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a);
    control_array out;
    for (int i = 0; i < CONTROL_DIM; i++)
      out[i] = (float)((1.0 - a) * u(i, lower) + a * u(i, lower + 1 < params_.num_timesteps_ ? lower + 1 : lower));
    if (h_)
      check(mppi_enforce_constraints(h_, state.data(), out.data()));
    return out;
  }
  virtual void chooseAppropriateKernel()
  {
    ensureHandle();
    syncPlugins();
    int v = 0;
    check(mppi_choose_kernel(h_, 10, &v, nullptr, nullptr));
  }
  /** the engine handle (blobs, noise injection, sharding: everything of include/mppi_amd.h the classes do not wrap) */
  mppi_handle handle()
  {
    ensureHandle();
    return h_;
  }

  DYN_T* model_;
  COST_T* cost_;
  FB_T* fb_controller_;
  SAMPLING_T* sampler_;

protected:
  void check(mppi_status s) const
  {
    if (s != MPPI_OK)
      throw Error(s, std::string(mppi_status_string(s)) + ": " + mppi_last_error(h_));
  }
  virtual void afterCreate()
  {
  }
  void ensureHandle()
  {
    if (h_)
      return;
    const std::string& name = registeredModelName<DYN_T, COST_T, SAMPLING_T>();
    mppi_config cfg{};
    cfg.model = name.c_str();
    cfg.controller = kind_;
    cfg.num_rollouts = NUM_ROLLOUTS;
    cfg.num_timesteps = params_.num_timesteps_;
    cfg.dt = params_.dt_;
    cfg.lambda = params_.lambda_;
    cfg.alpha = params_.alpha_;
    cfg.num_iters = params_.num_iters_;
    cfg.seed = (unsigned long long)params_.seed_;
    cfg.noise_source = MPPI_NOISE_PHILOX_FUSED;
    cfg.stream = (void*)stream_;
    // dynamics_rollout_dim_ = (rollouts per block, lanes per rollout, .) is honoured or refused, never replaced: (0, 0, .) asks
    // for the engine's default, anything else has to be one of RolloutShapes<...>::type (the reference's (64, 4, 1) is)
    cfg.block_x = (int)params_.dynamics_rollout_dim_.x;
    cfg.block_y = (int)params_.dynamics_rollout_dim_.y;
    const mppi_status s = mppi_create(&cfg, &h_);
    if (s == MPPI_ERR_LAUNCH_SHAPE)
      throw Error(s, std::string("mppi_create: ") + mppi_last_error(nullptr) +
                         " — dynamics_rollout_dim_ has to be (0, 0, .) or one of mppi_amd::templated::RolloutShapes<DYN_T, COST_T, "
                         "SAMPLING_T>::type (specialise it to instantiate the kernels for another block shape)");
    if (s != MPPI_OK)
      throw Error(s, std::string("mppi_create: ") + mppi_last_error(nullptr));
    pushControllerParams();
    check(mppi_set_nominal_control(h_, init_control_traj_.data()));
    afterCreate();
  }
  void pushControllerParams()
  {
    check(mppi_set_lambda_alpha(h_, params_.lambda_, params_.alpha_));
    check(mppi_set_num_iters(h_, params_.num_iters_));
    bool any = false;
    for (int i = 0; i < CONTROL_DIM; i++)
      any = any || params_.slide_control_scale_[i] != 0.0f;
    if (any)
      check(mppi_set_slide_control_scale(h_, params_.slide_control_scale_.data()));
  }
  /** the user's plugin objects are the source of truth: push what changed since the last launch */
  virtual void syncPlugins()
  {
    typedef typename DYN_T::DYN_PARAMS_T DP;
    typedef typename COST_T::COST_PARAMS_T CP;
    const DP dp = model_->getParams();
    if (!pushed_ || std::memcmp(&dp, &dyn_params_, sizeof(DP)) != 0)
    {
      if (!std::is_empty<DP>::value)
        check(mppi_set_dynamics_params(h_, &dp, sizeof(DP)));
      std::memcpy((void*)&dyn_params_, &dp, sizeof(DP));
    }
    const CP cp = cost_->getParams();
    if (!pushed_ || std::memcmp(&cp, &cost_params_, sizeof(CP)) != 0)
    {
      check(mppi_set_cost_params(h_, &cp, sizeof(CP)));
      std::memcpy((void*)&cost_params_, &cp, sizeof(CP));
    }
    float rng[2 * CONTROL_DIM], db[CONTROL_DIM];
    for (int i = 0; i < CONTROL_DIM; i++)
    {
      rng[2 * i] = model_->control_rngs_[i].x;
      rng[2 * i + 1] = model_->control_rngs_[i].y;
      db[i] = model_->control_deadband_[i];
    }
    if (!pushed_ || std::memcmp(rng, rng_, sizeof(rng)) != 0)
    {
      check(mppi_set_control_ranges(h_, rng));
      std::memcpy(rng_, rng, sizeof(rng));
    }
    if (!pushed_ || std::memcmp(db, db_, sizeof(db)) != 0)
    {
      check(mppi_set_control_deadband(h_, db));
      std::memcpy(db_, db, sizeof(db));
    }
    const typename SAMPLING_T::SAMPLING_PARAMS_T sp = sampler_->getParams();
    if (!pushed_ || std::memcmp(&sp, &smp_params_, sizeof(sp)) != 0)
    {
      float sd[2 * CONTROL_DIM], cc[CONTROL_DIM];
      // the sampler's own table may hold ONE distribution (MAX_DISTRIBUTIONS of its parameter type): never read past it
      constexpr int SD_HELD = (int)(sizeof(sp.std_dev) / sizeof(sp.std_dev[0]));
      for (int i = 0; i < 2 * CONTROL_DIM; i++)
        sd[i] = sp.std_dev[i < SD_HELD ? i : i % CONTROL_DIM];
      if (sp.num_distributions <= 1 || SD_HELD < 2 * CONTROL_DIM)  // one set of standard deviations serves both systems of Tube / Robust MPPI
        for (int i = 0; i < CONTROL_DIM; i++)
          sd[CONTROL_DIM + i] = sd[i];
      for (int i = 0; i < CONTROL_DIM; i++)
        cc[i] = sp.control_cost_coeff[i];
      mppi_gaussian_params gp{ sd, cc, sp.pure_noise_trajectories_percentage, sp.std_dev_decay, sp.sum_strides };
      check(mppi_set_sampler_params(h_, &gp));
      check(mppi_set_independent_noise(h_, sp.use_same_noise_for_all_distributions ? 0 : 1));
      std::memcpy((void*)&smp_params_, &sp, sizeof(sp));
    }
    pushed_ = true;
  }

  PARAMS_T params_;
  mppi_handle h_ = nullptr;
  int kind_;
  hipStream_t stream_;
  control_trajectory init_control_traj_;
  bool pushed_ = false;
  alignas(8) unsigned char dyn_params_[sizeof(typename DYN_T::DYN_PARAMS_T)];
  alignas(8) unsigned char cost_params_[sizeof(typename COST_T::COST_PARAMS_T)];
  alignas(8) unsigned char smp_params_[sizeof(typename SAMPLING_T::SAMPLING_PARAMS_T)];
  float rng_[2 * DYN_T::CONTROL_DIM], db_[DYN_T::CONTROL_DIM];
};
}  // namespace templated
}  // namespace mppi_amd

#define MPPI_AMD_TPL_ARGS DYN_T, COST_T, FB_T, MAX_TIMESTEPS, NUM_ROLLOUTS, SAMPLING_T
#define MPPI_AMD_TPL_HEAD                                                                                             \
  template <class DYN_T, class COST_T, class FB_T, int MAX_TIMESTEPS, int NUM_ROLLOUTS,                                \
            class SAMPLING_T = ::mppi::sampling_distributions::GaussianDistribution<typename DYN_T::DYN_PARAMS_T>>

/** reference: controllers/MPPI/mppi_controller.cuh:14-60 */
MPPI_AMD_TPL_HEAD class VanillaMPPIController : public mppi_amd::templated::ControllerBase<MPPI_AMD_TPL_ARGS>
{
public:
  typedef mppi_amd::templated::ControllerBase<MPPI_AMD_TPL_ARGS> PARENT_CLASS;
  using control_trajectory = typename PARENT_CLASS::control_trajectory;
  VanillaMPPIController(DYN_T* model, COST_T* cost, FB_T* fb_controller, SAMPLING_T* sampler, float dt, int max_iter,
                        float lambda, float alpha, int num_timesteps = MAX_TIMESTEPS,
                        const control_trajectory& init_control_traj = control_trajectory::Zero(), hipStream_t stream = nullptr)
    : PARENT_CLASS(MPPI_CONTROLLER_VANILLA, model, cost, fb_controller, sampler, dt, max_iter, lambda, alpha, num_timesteps,
                   init_control_traj, stream)
  {
  }
  std::string getControllerName() const override
  {
    return "Vanilla MPPI";
  }
};

/** reference: controllers/Tube-MPPI/tube_mppi_controller.cuh:20-110 */
MPPI_AMD_TPL_HEAD class TubeMPPIController : public mppi_amd::templated::ControllerBase<MPPI_AMD_TPL_ARGS>
{
public:
  typedef mppi_amd::templated::ControllerBase<MPPI_AMD_TPL_ARGS> PARENT_CLASS;
  using control_trajectory = typename PARENT_CLASS::control_trajectory;
  using state_trajectory = typename PARENT_CLASS::state_trajectory;
  TubeMPPIController(DYN_T* model, COST_T* cost, FB_T* fb_controller, SAMPLING_T* sampler, float dt, int max_iter,
                     float lambda, float alpha, int num_timesteps = MAX_TIMESTEPS,
                     const control_trajectory& init_control_traj = control_trajectory::Zero(), hipStream_t stream = nullptr)
    : PARENT_CLASS(MPPI_CONTROLLER_TUBE, model, cost, fb_controller, sampler, dt, max_iter, lambda, alpha, num_timesteps,
                   init_control_traj, stream)
  {
  }
  std::string getControllerName() const override
  {
    return "Tube MPPI";
  }
  void setNominalThreshold(float threshold)
  {
    nominal_threshold_ = threshold;
    if (this->h_)
      this->check(mppi_set_nominal_threshold(this->h_, threshold));
  }
  float getNominalThreshold() const
  {
    return nominal_threshold_;
  }
  control_trajectory getNominalControlSeq() const
  {
    control_trajectory u = control_trajectory::Zero();
    if (this->h_)
      this->check(mppi_get_nominal_control_seq(this->h_, u.data()));
    return u;
  }
  state_trajectory getNominalStateSeq() const
  {
    state_trajectory x = state_trajectory::Zero();
    if (this->h_)
      this->check(mppi_get_nominal_state_seq(this->h_, x.data()));
    return x;
  }
  /** reference: the actual system's trajectories under the Tube names (tube_mppi_controller.cuh getActual...) */
  control_trajectory getActualControlSeq() const
  {
    return this->getControlSeq();
  }
  state_trajectory getActualStateSeq() const
  {
    return this->getTargetStateSeq();
  }

protected:
  void afterCreate() override
  {
    this->check(mppi_set_nominal_threshold(this->h_, nominal_threshold_));
  }
  float nominal_threshold_ = 20.0f;  ///< tube_mppi_controller.cuh: nominal_threshold_ default
};

/** reference: controllers/R-MPPI/robust_mppi_controller.cuh:60-150 */
MPPI_AMD_TPL_HEAD class RobustMPPIController : public mppi_amd::templated::ControllerBase<MPPI_AMD_TPL_ARGS>
{
public:
  typedef mppi_amd::templated::ControllerBase<MPPI_AMD_TPL_ARGS> PARENT_CLASS;
  using control_trajectory = typename PARENT_CLASS::control_trajectory;
  using state_array = typename PARENT_CLASS::state_array;
  RobustMPPIController(DYN_T* model, COST_T* cost, FB_T* fb_controller, SAMPLING_T* sampler, float dt, int max_iter,
                       float lambda, float alpha, float value_function_threshold, int num_timesteps = MAX_TIMESTEPS,
                       const control_trajectory& init_control_traj = control_trajectory::Zero(),
                       int num_candidate_nominal_states = 9, int optimization_stride = 1, hipStream_t stream = nullptr)
    : PARENT_CLASS(MPPI_CONTROLLER_ROBUST, model, cost, fb_controller, sampler, dt, max_iter, lambda, alpha, num_timesteps,
                   init_control_traj, stream)
    , value_function_threshold_(value_function_threshold)
    , num_candidates_(num_candidate_nominal_states)
    , optimization_stride_(optimization_stride)
  {
  }
  std::string getControllerName() const override
  {
    return "Robust MPPI";
  }
  /** robust_mppi_controller.cu:509-560: candidate nominal states, init-eval kernel, best candidate, slide */
  void updateImportanceSamplingControl(const state_array& state, int stride)
  {
    this->ensureHandle();
    this->syncPlugins();
    this->check(mppi_update_importance_sampling_control(this->h_, state.data(), stride));
  }
  control_trajectory getNominalControlSeq() const
  {
    control_trajectory u = control_trajectory::Zero();
    if (this->h_)
      this->check(mppi_get_nominal_control_seq(this->h_, u.data()));
    return u;
  }
  state_array getNominalState() const
  {
    state_array x = state_array::Zero();
    if (this->h_)
      this->check(mppi_get_rmppi_state(this->h_, x.data(), nullptr, nullptr, nullptr));
    return x;
  }
  void setValueFunctionThreshold(float t)
  {
    value_function_threshold_ = t;
    if (this->h_)
      this->check(mppi_set_rmppi_params(this->h_, value_function_threshold_, num_candidates_, samples_per_candidate_));
  }

protected:
  void afterCreate() override
  {
    this->check(mppi_set_rmppi_params(this->h_, value_function_threshold_, num_candidates_, samples_per_candidate_));
  }
  /** the DDP gains live in the caller's feedback object; upload them when they changed */
  void syncPlugins() override
  {
    PARENT_CLASS::syncPlugins();
    if (this->fb_controller_ && (!gains_pushed_ || this->fb_controller_->version() != gains_version_))
    {
      const std::vector<float>& g = this->fb_controller_->getFeedbackGains();
      if (g.size() == (size_t)this->params_.num_timesteps_ * DYN_T::STATE_DIM * DYN_T::CONTROL_DIM)
      {
        this->check(mppi_set_feedback_gains(this->h_, g.data(), this->fb_controller_->accumulateAllStates() ? 1 : 0));
        gains_pushed_ = true;
        gains_version_ = this->fb_controller_->version();
      }
    }
  }
  float value_function_threshold_;
  int num_candidates_;
  int samples_per_candidate_ = 32;  ///< robust_mppi_controller.cuh SAMPLES_PER_CONDITION_MULTIPLIER
  int optimization_stride_;
  bool gains_pushed_ = false;
  unsigned gains_version_ = 0;
};

/** reference: controllers/ColoredMPPI/colored_mppi_controller.cuh:20-80 with sampling_distributions/colored_noise */
template <class DYN_T, class COST_T, class FB_T, int MAX_TIMESTEPS, int NUM_ROLLOUTS,
          class SAMPLING_T = ::mppi::sampling_distributions::ColoredNoiseDistribution<typename DYN_T::DYN_PARAMS_T>>
class ColoredMPPIController : public mppi_amd::templated::ControllerBase<MPPI_AMD_TPL_ARGS>
{
public:
  typedef mppi_amd::templated::ControllerBase<MPPI_AMD_TPL_ARGS> PARENT_CLASS;
  using control_trajectory = typename PARENT_CLASS::control_trajectory;
  ColoredMPPIController(DYN_T* model, COST_T* cost, FB_T* fb_controller, SAMPLING_T* sampler, float dt, int max_iter,
                        float lambda, float alpha, int num_timesteps = MAX_TIMESTEPS,
                        const control_trajectory& init_control_traj = control_trajectory::Zero(), hipStream_t stream = nullptr)
    : PARENT_CLASS(MPPI_CONTROLLER_COLORED, model, cost, fb_controller, sampler, dt, max_iter, lambda, alpha, num_timesteps,
                   init_control_traj, stream)
  {
  }
  std::string getControllerName() const override
  {
    return "Colored MPPI";
  }

protected:
  void syncPlugins() override
  {
    PARENT_CLASS::syncPlugins();
    float ex[DYN_T::CONTROL_DIM];
    bool changed = !colored_pushed_ || decay_ != this->sampler_->offset_decay_rate_ || fmin_ != this->sampler_->fmin_;
    for (int i = 0; i < DYN_T::CONTROL_DIM; i++)
    {
      ex[i] = this->sampler_->exponents_[i];
      changed = changed || ex[i] != ex_[i];
    }
    if (changed)
    {
      this->check(mppi_set_colored_noise_params(this->h_, ex, this->sampler_->offset_decay_rate_, this->sampler_->fmin_));
      for (int i = 0; i < DYN_T::CONTROL_DIM; i++)
        ex_[i] = ex[i];
      decay_ = this->sampler_->offset_decay_rate_;
      fmin_ = this->sampler_->fmin_;
      colored_pushed_ = true;
    }
  }
  bool colored_pushed_ = false;
  float ex_[DYN_T::CONTROL_DIM] = {};
  float decay_ = 0.0f, fmin_ = 0.0f;
};

#undef MPPI_AMD_TPL_ARGS
#undef MPPI_AMD_TPL_HEAD
#endif
