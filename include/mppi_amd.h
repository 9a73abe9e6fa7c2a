/**
 * mppi_amd.h — C ABI of the MI355X-native MPPI rollout-and-reduce engine (libmppi_amd.so).
 *
 * The reference (ACDSLab/MPPI-Generic) has no C ABI: its boundary is a compile-time template contract, and its only
 * binary form is explicit instantiation of controller templates into shared libraries
 * (reference: src/controllers/cartpole/cartpole_mppi.cu:30-42, src/controllers/autorally/autorally_mppi.cu:10-11,
 * include/mppi/instantiations/).  This header is the C-level equivalent of those instantiations: every entry point
 * names the reference interface it stands in for (paths relative to the reference's include/mppi/).  The templated
 * Dynamics / Cost / SamplingDistribution plugin contract itself is in include/mppi_amd/plugin/ and
 * include/mppi_amd/sampling_distributions/; a model is added by writing the plugin and registering one instantiation
 * (see INTEGRATION.md).
 *
 * Conventions
 *  - Plain pointers and sizes only.  Host arrays are caller-owned, row-major, last index fastest:
 *      control sequence u[T][C] (== the reference's Eigen control_trajectory, C x T column-major, controller.cuh:96),
 *      state sequence x[T][S], initial state x0[S] (Tube / RMPPI: see each call), noise eps[K][T][C], costs[D][K].
 *  - Every call returns an mppi_status; nothing exit()s or throws across the boundary
 *    (reference: HANDLE_ERROR -> exit, utils/gpu_err_chk.cuh:32-40; launch-shape checks -> exit, core/mppi_common.cu:
 *    1266-1277, 1305-1310; shared-memory overflow -> std::runtime_error, controllers/MPPI/mppi_controller.cu:64-76).
 *    mppi_last_error() returns the message of the last failing call on that handle.
 *  - One call in flight per handle (the reference's controllers are single-caller too, core/base_plant.hpp:398-428);
 *    handles are independent.  A handle owns its device buffers and, unless given one, its HIP stream.
 *  - The library never falls back to a CPU path: without a usable HIP device mppi_create fails with MPPI_ERR_NO_DEVICE.
 */
#ifndef MPPI_AMD_H_
#define MPPI_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPPI_AMD_VERSION_MAJOR 0
#define MPPI_AMD_VERSION_MINOR 1

typedef struct mppi_handle_s* mppi_handle;

typedef enum mppi_status
{
  MPPI_OK = 0,
  MPPI_ERR_INVALID_ARG = 1,
  MPPI_ERR_UNKNOWN_MODEL = 2,
  MPPI_ERR_NO_DEVICE = 3,
  MPPI_ERR_HIP = 4,           /* a HIP runtime call failed; message in mppi_last_error */
  MPPI_ERR_LAUNCH_SHAPE = 5,  /* unsupported block shape for this instantiation (mppi_common.cu:1266-1277, 1305-1310) */
  MPPI_ERR_LDS_OVERFLOW = 6,  /* LDS request exceeds 160 KiB/CU (mppi_controller.cu:64-76 runtime_error) */
  MPPI_ERR_STATE = 7,         /* call not valid in the handle's current state / configuration */
  MPPI_ERR_NAN = 8,           /* non-finite value in the resulting control (base_plant.hpp:515-535 exit(-1)) */
  MPPI_ERR_COMM = 9,          /* RCCL failure */
  MPPI_ERR_UNSUPPORTED = 10
} mppi_status;

/** which controller loop runs on top of the kernels */
typedef enum mppi_controller_kind
{
  MPPI_CONTROLLER_VANILLA = 0, /* controllers/MPPI/mppi_controller.cu:151-241 */
  MPPI_CONTROLLER_TUBE = 1,    /* controllers/Tube-MPPI/tube_mppi_controller.cu:157-299 (two systems per launch) */
  MPPI_CONTROLLER_ROBUST = 2,  /* controllers/R-MPPI/robust_mppi_controller.cu:548-755 (nominal + real system, tracking
                                  feedback, candidate nominal states) */
  MPPI_CONTROLLER_COLORED = 3  /* controllers/ColoredMPPI/colored_mppi_controller.cu:134-240: the vanilla loop with the
                                  colored-noise sampler; after smoothing only control channel 1 is clamped (:232-237) */
} mppi_controller_kind;

/** where eps ~ N(0,1) comes from (reference: curandGenerateNormal, sampling_distributions/gaussian/gaussian.cu:380-394) */
typedef enum mppi_noise_source
{
  MPPI_NOISE_PHILOX_FUSED = 0, /* Philox4x32-10 drawn inside the rollout kernel; nothing materialised in HBM */
  MPPI_NOISE_INJECTED = 1,     /* caller-supplied eps (mppi_inject_noise): parity / replay mode */
  MPPI_NOISE_ROCRAND_HOST = 2  /* rocrand_generate_normal (Philox) into an HBM eps buffer, the reference's structure */
} mppi_noise_source;

/**
 * Structure of the rollout kernel.  The reference chooses between its "single" and "split" kernels by timing both at
 * construction (chooseAppropriateKernel, controllers/MPPI/mppi_controller.cu:44-143); here the choice is explicit, with an
 * automatic default.  All variants produce bit-identical trajectory costs.
 */
typedef enum mppi_kernel_variant
{
  MPPI_KERNEL_AUTO = 0,     /* pipeline where the model is registered for it and the block shape is (64, 1), else fused */
  MPPI_KERNEL_FUSED = 1,    /* one wave carries sampling, dynamics and cost of its rollouts (engine/rollout_kernel.hpp) */
  MPPI_KERNEL_PIPELINE = 2  /* sampler / dynamics / cost waves decoupled through LDS (engine/rollout_pipeline_kernel.hpp);
                             * Robust MPPI: engine/rmppi_pipeline_kernel.hpp (rollout and candidate evaluation), for models
                             * registered for it (replicated-lane or one-lane PIPELINE dynamics) with the Gaussian sampler —
                             * what AUTO picks there as well */
} mppi_kernel_variant;

/**
 * Arithmetic of the last stage of an iteration (baseline -> weights -> normaliser -> weighted mean).  Every mode is far inside
 * the 1e-5 parity bar for ONE iteration; the reference-order modes exist so that a free-running closed loop — which amplifies
 * any difference — can be compared with the reference bit for bit (tests/test_closed_loop_parity.py).
 */
typedef enum mppi_reduction_mode
{
  MPPI_REDUCTION_FUSED = 0,               /* default: block-local softmin records in the rollout kernel's epilogue + one merge
                                             kernel (rescale-merge: exact in real arithmetic, ~6e-8 from the reference's order) */
  MPPI_REDUCTION_REFERENCE_ORDER = 1,     /* the reference's own order, operation for operation: global first-minimum baseline,
                                             w_k against it, eta = float(sum of double(w_k)) in index order, weight = w_k / eta per
                                             rollout, sum_strides consecutive rollouts serially, then the cells serially
                                             (core/mppi_common.cu:885-900, 958-966, 1055-1063, 1115-1160).  Samples go through HBM */
  MPPI_REDUCTION_REFERENCE_ORDER_FMA = 2  /* the same with inter = fma(weight, v, inter): what nvcc's default -fmad=true makes of
                                             `inter += weight * v` on the reference's GPU path */
} mppi_reduction_mode;

/**
 * Construction parameters == the template arguments + ControllerParams of the reference
 * (controllers/controller.cuh:46-68; template <DYN, COST, FB, SAMPLING, MAX_TIMESTEPS, NUM_ROLLOUTS>, :70-75).
 */
typedef struct mppi_config
{
  const char* model;     /* name of a registered instantiation: "cartpole", "double_integrator", ... (mppi_list_models) */
  int controller;        /* mppi_controller_kind */
  int num_rollouts;      /* K over ALL ranks (NUM_ROLLOUTS) */
  int num_timesteps;     /* T (MAX_TIMESTEPS / num_timesteps_) */
  float dt;              /* dt_ */
  float lambda;          /* lambda_ */
  float alpha;           /* alpha_ */
  int num_iters;         /* num_iters_ : optimisation iterations per mppi_compute_control */
  uint64_t seed;         /* seed_ (controller.cuh:59; the reference defaults to wall-clock time, here the caller decides) */
  int noise_source;      /* mppi_noise_source */
  int block_x, block_y;  /* dynamics_rollout_dim_.{x,y} hint; 0 = instantiation default */
  int device;            /* HIP device ordinal */
  void* stream;          /* hipStream_t to run on, or NULL: the handle creates its own (Managed::stream_, utils/managed.cuh:57) */
  int rank, world_size;  /* K-sharding over GPUs: this handle owns rollouts [rank*K/world, (rank+1)*K/world) */
  int save_samples;      /* != 0: keep the clamped samples v[D][K_local][T][C] in HBM (control_samples_d_) */
  int kernel_variant;    /* mppi_kernel_variant: which rollout kernel structure to use */
  int force_exchange;    /* != 0 with world_size == 1: still run local merge -> all-gather -> global merge (exercises the
                            RCCL path on a single GPU; needs mppi_comm_init_rccl or an external exchange) */
} mppi_config;

/** reference: GaussianParamsImpl, sampling_distributions/gaussian/gaussian.cuh:21-61 */
typedef struct mppi_gaussian_params
{
  const float* std_dev;            /* [D][C] */
  const float* control_cost_coeff; /* [C] */
  float pure_noise_trajectories_percentage;
  float std_dev_decay;
  int sum_strides; /* rollouts per cell of the weighted reduction (gaussian.cuh:30, default 32): only the reference-order
                      reduction modes use it; the block-local reduction does not need it */
} mppi_gaussian_params;

/** per-system statistics (reference: MPPIFreeEnergyStatistics controllers/controller.cuh:22-38, getBaselineCost/getNormalizerCost) */
typedef struct mppi_system_stats
{
  float baseline;   /* rho  = min_k S_k              (computeBaselineCost, core/mppi_common.cu:858-900) */
  float normalizer; /* eta  = sum_k exp(-(S_k-rho)/lambda) (computeNormalizer, core/mppi_common.cu:1055-1063) */
  float free_energy_mean;
  float free_energy_variance;
  float free_energy_modified_variance; /* computeFreeEnergy, core/mppi_common.cu:1065-1081 */
} mppi_system_stats;

typedef struct mppi_stats
{
  mppi_system_stats real_sys;
  mppi_system_stats nominal_sys; /* Tube / RMPPI */
  int nominal_state_used;        /* tube_mppi_controller.cu:268-280 */
} mppi_stats;

/* ---------------------------------------------------------------- library ---------------------------------------- */
const char* mppi_version(void);
const char* mppi_status_string(mppi_status s);
/** number of visible HIP devices (0 => mppi_create will fail with MPPI_ERR_NO_DEVICE) */
int mppi_device_count(void);
/** names of the registered (DYN, COST, SAMPLER) instantiations, '\n'-separated (reference: include/mppi/instantiations/) */
const char* mppi_list_models(void);
/** hex digest of the sources libmppi_amd.so was built from (buildlib.py; __graft_entry__.smoke() compares it with the tree) */
const char* mppi_source_hash(void);

/* ---------------------------------------------------------------- model registration ------------------------------ */
/**
 * The reference's user instantiates the controller templates with their own Dynamics / Cost classes in their own
 * translation unit (src/controllers/cartpole/cartpole_mppi.cu:30-42).  Here such a translation unit registers a factory
 * for its ModelT<...> under a model name (include/mppi_amd/engine/model_registry.hpp: MPPI_REGISTER_MODEL) — the in-tree
 * models do it from static initialisers of libmppi_amd.so, an out-of-tree model from a library of its own that is linked
 * next to libmppi_amd.so or loaded with mppi_load_plugin.
 */
typedef enum mppi_sampler_kind
{
  MPPI_SAMPLER_GAUSSIAN = 0, /* used by the Vanilla / Tube / Robust controllers */
  MPPI_SAMPLER_COLORED = 1   /* used by MPPI_CONTROLLER_COLORED */
} mppi_sampler_kind;
/** returns a new mppi::engine::ModelBase* (owned by the handle that asked for it) */
typedef void* (*mppi_model_factory)(void);
/** abi_fingerprint = mppi::engine::engineAbiFingerprint() in the caller's build (sizes of ModelBase and of the kernel argument
 *  blocks): a mismatch (header / library skew) is refused */
mppi_status mppi_register_model(const char* name, int sampler_kind, mppi_model_factory factory, int abi_fingerprint);
/** what MPPI_REGISTER_MODEL and the templated controllers call: the same, plus what the instantiation says about itself.
 *  MPPI_MODEL_ROLE_SEPARATED: it carries role-separated rollout kernels (waves of a block with different jobs: a block barrier
 *  inside a plugin's per-step method would never complete there); MPPI_MODEL_BARRIER_FREE_DECLARED: every plugin class those
 *  kernels would run declares `static constexpr bool MPPI_BARRIER_FREE_STEP = true` (mppi_amd/plugin/parallel_utils.hpp).
 *  ROLE_SEPARATED without BARRIER_FREE_DECLARED is refused with MPPI_ERR_INVALID_ARG — the reference's own Dynamics::step has
 *  two block barriers (dynamics/dynamics.cu:138,140), so a model that says nothing is taken to have them and belongs on the
 *  fused kernel (PIPELINE = false).  (mppi_create checks the model object again, whichever way it was registered.) */
#define MPPI_MODEL_ROLE_SEPARATED 1u
#define MPPI_MODEL_BARRIER_FREE_DECLARED 2u
mppi_status mppi_register_model_checked(const char* name, int sampler_kind, mppi_model_factory factory, int abi_fingerprint,
                                        unsigned flags);
/** dlopen()s a library whose static initialisers call mppi_register_model; the library stays loaded */
mppi_status mppi_load_plugin(const char* path);

/* ---------------------------------------------------------------- lifecycle -------------------------------------- */
/** Controller constructor + GPUSetup + allocateCUDAMemory (controllers/controller.cuh:160-216, 269-277, 931-992) */
mppi_status mppi_create(const mppi_config* cfg, mppi_handle* out);
/** Controller destructor + freeCudaMem (controllers/controller.cuh:194-216) */
void mppi_destroy(mppi_handle h);
/** message of the last failing call; h == NULL: last mppi_create failure of this thread */
const char* mppi_last_error(mppi_handle h);
/** STATE_DIM, CONTROL_DIM, OUTPUT_DIM of the model (dynamics/dynamics.cuh:74-76) and the number of systems D per launch */
mppi_status mppi_get_dims(mppi_handle h, int* state_dim, int* control_dim, int* output_dim, int* num_systems);
/** rollouts owned by this handle (K / world_size) */
mppi_status mppi_get_local_rollouts(mppi_handle h, int* k_local, int* k_offset);
/**
 * Kernel launches this handle has made since mppi_create: rollout launches (rolloutKernel and its pipelined forms — the
 * reference's launchRolloutKernel / launchFastRolloutKernel, core/mppi_common.cu:520-630) and launches of the reduction stage
 * (the merge of the per-block records, or the reference-order kernels counted as one).  A Vanilla handle whose rollout kernel
 * merges the previous iteration's records in its sampler waves makes n rollout launches and ONE merge launch for n iterations.
 * Either pointer may be NULL.
 */
mppi_status mppi_get_launch_counts(mppi_handle h, unsigned long long* rollout_launches, unsigned long long* merge_launches);

/* ---------------------------------------------------------------- parameters ------------------------------------- */
/** Dynamics::setParams + paramsToDevice (dynamics/dynamics.cu:3-17); pod = the model's *_dynamics_params (mppi_amd/model_params.h) */
mppi_status mppi_set_dynamics_params(mppi_handle h, const void* pod, size_t nbytes);
/** Cost::setParams + paramsToDevice (cost_functions/cost.cu:5-13) */
mppi_status mppi_set_cost_params(mppi_handle h, const void* pod, size_t nbytes);
/** SamplingDistribution::setParams (sampling_distributions/sampling_distribution.cuh:93-116) */
mppi_status mppi_set_sampler_params(mppi_handle h, const mppi_gaussian_params* p);
/** ColoredNoiseParamsImpl (sampling_distributions/colored_noise/colored_noise.cuh:45-73): exponents[C] (0 = white),
 *  offset_decay_rate, fmin.  Only for handles created with MPPI_CONTROLLER_COLORED; std_dev etc. come from
 *  mppi_set_sampler_params as for the Gaussian sampler (ColoredNoiseParams extends GaussianParams). */
/**
 * use_same_noise_for_all_distributions (sampling_distributions/sampling_distribution.cuh:20; gaussian.cu:378-394).  Default
 * (0): the systems of a Tube / Robust controller see the SAME noise, as the reference's default.  independent != 0: every
 * distribution draws its own — Philox stream d in the generator modes, slab d of the buffer for injected noise, which then is
 * eps[n_iters][D][K_local][T][C].
 */
mppi_status mppi_set_independent_noise(mppi_handle h, int independent);
/**
 * time_specific_std_dev (GaussianTimeVaryingStdDevParams, sampling_distributions/gaussian/gaussian.cuh:64-95; used by
 * setGaussianControls, gaussian.cu:21-43, and by the likelihood-ratio cost, gaussian.cu:488-493): std_dev[D][T][C], one sigma per
 * distribution, time step and control instead of mppi_gaussian_params.std_dev.  NULL switches back.
 */
mppi_status mppi_set_time_specific_std_dev(mppi_handle h, const float* std_dev);
mppi_status mppi_set_colored_noise_params(mppi_handle h, const float* exponents, float offset_decay_rate, float fmin);
/** Dynamics::setControlRanges (dynamics/dynamics.cu:19-36); lo_hi = [C][2] */
/**
 * ColoredMPPI options (controllers/ColoredMPPI/colored_mppi_controller.cuh:18-22, 95-193):
 *  - gamma, r_exp: Tsallis weights w = (S - rho < gamma) ? exp(log(1 - (S - rho) / gamma) / (r - 1)) : 0 instead of the exponential
 *    ones when BOTH are non-zero (core/mppi_common.cu:968-985; colored_mppi_controller.cu:198-206);
 *  - state leash (setStateLeashLength / setLeashActive, colored_mppi_controller.cu:150-156): with the leash active every
 *    mppi_compute_control starts from Dynamics::enforceLeash(measured state, previous state trajectory[leash_jump],
 *    state_leash_dist[S]) instead of the measured state.  state_leash_dist may be NULL (keeps the previous / zero leash).
 */
mppi_status mppi_set_colored_mppi_params(mppi_handle h, float gamma, float r_exp, const float* state_leash_dist,
                                         int leash_active, int leash_jump);
mppi_status mppi_set_control_ranges(mppi_handle h, const float* lo_hi);
/** Dynamics::setControlDeadbands (dynamics/dynamics.cu:38-55) */
mppi_status mppi_set_control_deadband(mppi_handle h, const float* deadband);
/** setLambda / setAlpha / setNumIters (controllers/controller.cuh:700-760) */
mppi_status mppi_set_lambda_alpha(mppi_handle h, float lambda, float alpha);
mppi_status mppi_set_num_iters(mppi_handle h, int num_iters);
/** mppi_reduction_mode; MPPI_ERR_UNSUPPORTED on a K-sharded handle (the reference order runs over all rollouts).  The
 *  environment variable MPPI_AMD_REDUCTION=reference | reference_fma selects the mode for every handle created afterwards. */
mppi_status mppi_set_reduction_mode(mppi_handle h, int mode);
/** slide_control_scale_ (controller.cuh:67) [C]; Tube: nominal_threshold_ (Tube-MPPI/tube_mppi_controller.cuh:20) */
mppi_status mppi_set_slide_control_scale(mppi_handle h, const float* scale);
mppi_status mppi_set_nominal_threshold(mppi_handle h, float threshold);
/**
 * Bulk model data, the role of the reference's .npz loaders (NeuralNetModel::loadParams -> FNNHelper::loadParams,
 * utils/nn_helpers/fnn_helper.cu:96-174; ARStandardCost::loadTrackData, cost_functions/autorally/ar_standard_cost.cu:84-142).
 * The caller parses the file (numpy / cnpy) and hands over plain arrays:
 *   "dynamics_weights"  FNN parameters in the reference's blob order [W1 (out x in row-major) | b1 | W2 | b2 | ...]
 *                       (fnn_helper.cu:176-183); dims = {count}
 *   "costmap"           channel 0 of the track costmap, row-major [height][width]; dims = {height, width}.  The
 *                       world -> texture transform goes in the cost parameter block (r_c1, r_c2, trs).
 */
mppi_status mppi_set_model_blob(mppi_handle h, const char* name, const float* data, size_t count, const int* dims,
                                int ndims);
/**
 * Models with an LSTM in the rollout (bicycle_slip_lstm, racer_dubins_elevation_lstm_steering): new initial hidden / cell
 * state [H] each for every rollout of the following iterations — LSTMHelper::setHiddenState / setCellState +
 * copyHiddenCellToDevice (utils/nn_helpers/lstm_helper.cu:476-500) — without re-uploading the weights.
 */
mppi_status mppi_set_lstm_initial_state(mppi_handle h, const float* hidden, const float* cell);
/**
 * LSTMLSTMHelper::initializeLSTM (utils/nn_helpers/lstm_lstm_helper.cu:50-76) on the host, as in the reference: the
 * initialiser LSTM(init_input_dim, init_hidden_dim) with output network init_output_layers[num_init_output_layers]
 * (first entry init_hidden_dim + init_input_dim, last entry 2 * hidden_dim) and parameter blobs in the device helpers'
 * layouts reads the last init_len samples of buffer[cols][init_input_dim] and emits hidden_cell_out[2 * hidden_dim] =
 * [hidden | cell], the argument of mppi_set_lstm_initial_state().  No handle and no device involved.
 */
mppi_status mppi_lstm_lstm_initialize(int init_input_dim, int init_hidden_dim, const int* init_output_layers,
                                      int num_init_output_layers, const float* init_lstm_blob, const float* init_output_blob,
                                      int hidden_dim, int init_len, const float* buffer, int cols, float* hidden_cell_out);
/**
 * The same data straight from the reference's .npz files (read with cnpy there; a ZIP + NPY reader with zlib here):
 *   kind "dynamics"  FNNHelper::loadParams (utils/nn_helpers/fnn_helper.cu:96-174): keys {prefix}dynamics_W{i},
 *                    {prefix}dynamics_b{i}, i = 1.., float64
 *   kind "lstm"      LSTMHelper::loadParams (utils/nn_helpers/lstm_helper.cu:514-585): [model/]{prefix}lstm/weight_hh_l0,
 *                    weight_ih_l0, bias_hh_l0, bias_ih_l0 (PyTorch gate order, re-ordered and summed as there) and
 *                    {prefix}output/dynamics_W{i}, _b{i}; optional {prefix}lstm/h0, c0 (default zeros).  kinds "mean_lstm" and
 *                    "unc_lstm": the same layout into the mean / uncertainty network of racer_dubins_elevation_lstm_unc
 *                    (the reference reads them from one archive with the prefixes "terra/mean_network/" and
 *                    "terra/uncertainty_network/", dynamics/racer_dubins/racer_dubins_elevation_lstm_unc.cu:30-33)
 *   kind "costmap"   ARStandardCost::loadTrackData (cost_functions/autorally/ar_standard_cost.cu:84-142): xBounds, yBounds,
 *                    pixelsPerMeter, channel0; also sets the world -> texture transform of the cost
 * prefix may be NULL.  A git-LFS pointer file in place of the archive is reported as such.
 */
mppi_status mppi_load_npz(mppi_handle h, const char* kind, const char* path, const char* prefix);
/** handle-free access to one array of an .npz (as double, C order): count / dims[<= 8] / ndims are always filled, out only
 *  when non-NULL (capacity in elements).  Host-only; message of a failure: mppi_last_error(NULL). */
mppi_status mppi_npz_read_array(const char* path, const char* key, double* out, size_t capacity, size_t* count, int* dims,
                                int* ndims);
/** reseed the noise generator and reset its offset (controllers/controller.cu:200-207) */
mppi_status mppi_set_seed(mppi_handle h, uint64_t seed);

/* ---------------------------------------------------------------- control loop ----------------------------------- */
/** updateImportanceSampler / init_control_traj_ (controllers/controller.cuh:330-349): u[T][C] */
mppi_status mppi_set_nominal_control(mppi_handle h, const float* u);
/**
 * The raw noise eps[K_local][T][C] (before the mean / std-dev rule of setGaussianControls) that the next iteration
 * would use with this optimization_stride — for colored noise the output of powerlaw_psd_gaussian's pipeline
 * (colored_noise.cu:58-191: spectrum shaping, inverse real DFT, offset removal, normalisation).  Generator tests only;
 * the rollout never materialises this tensor.  Not available for the in-loop Philox draw of the Gaussian sampler
 * (use mppi_philox_normal for that stream).
 */
mppi_status mppi_sample_noise(mppi_handle h, int optimization_stride, float* eps_out);
/**
 * Parity / replay mode: eps[n_iters][K_local][T][C] replaces the generator for the next mppi_compute_control calls
 * (iteration i of a call uses slab i % n_iters).  MPPI_CONTROLLER_COLORED handles take the Gaussian SPECTRUM instead, in
 * the layout of the reference's samples_in_freq_complex_d_: z[n_iters][K_local][C][T+1][2] (colored_noise.cu:343).
 * Switches the handle to MPPI_NOISE_INJECTED.  n_iters == 0 switches
 * back to the configured generator.  (No reference equivalent: the reference never pins its noise)
 */
mppi_status mppi_inject_noise(mppi_handle h, const float* eps, int n_iters);
/**
 * Controller::computeControl(state, optimization_stride):
 *   Vanilla  controllers/MPPI/mppi_controller.cu:151-241       x0 = [S]
 *   Tube     controllers/Tube-MPPI/tube_mppi_controller.cu:157-299   x0 = actual state [S]
 * Runs num_iters optimisation iterations on the device, then smoothing, state-trajectory propagation and constraint
 * enforcement (controllers/controller.cuh:557-586, 643-663; mppi_controller.cu:225-231).
 * Vanilla / Colored: inputs and results travel through host memory mapped into the device (no copy command, no stream
 * synchronisation: the host spins on a flag the last kernel raises) and the call returns as soon as the CONTROL sequence and
 * the statistics are on the host — the state / output trajectories of u* (a T-step serial re-rollout) land a little later;
 * mppi_get_state_seq / mppi_get_output_seq wait for them and report a non-finite state trajectory (MPPI_ERR_NAN) then.
 * Tube / Robust, or MPPI_AMD_NO_SPIN=1 in the environment: returns when all results are on the host.
 */
mppi_status mppi_compute_control(mppi_handle h, const float* x0, int optimization_stride);
/** getControlSeq (controllers/controller.cuh:433-436): u_out[T][C]; Tube: the actual system's sequence */
mppi_status mppi_get_control_seq(mppi_handle h, float* u_out);
/** getTargetStateSeq (controllers/controller.cuh:438-446): x_out[T][S] */
mppi_status mppi_get_state_seq(mppi_handle h, float* x_out);
/** getTargetOutputSeq(): the outputs y[T][O] along the same trajectory (computeOutputTrajectoryHelper,
 *  controllers/controller.cuh:643-662: output after initializeDynamics, then after every step) */
mppi_status mppi_get_output_seq(mppi_handle h, float* y_out);
/** Tube: getNominalControlSeq / getNominalStateSeq equivalents (Tube-MPPI/tube_mppi_controller.cuh:86-106) */
mppi_status mppi_get_nominal_control_seq(mppi_handle h, float* u_out);
mppi_status mppi_get_nominal_state_seq(mppi_handle h, float* x_out);
/** slideControlSequence(steps) (controllers/controller.cuh:351-356, 588-615; Tube: tube_mppi_controller.cu:312-323) */
mppi_status mppi_slide(mppi_handle h, int steps);
/** trajectory costs S[D][K_local] of the last iteration (trajectory_costs_d_, core/mppi_common.cu:850) */
mppi_status mppi_get_costs(mppi_handle h, float* costs);
/** baseline, normaliser, free-energy statistics of the last iteration (controllers/controller.cuh:22-38, 455-472) */
mppi_status mppi_get_stats(mppi_handle h, mppi_stats* out);
/** clamped samples v[D][K_local][T][C] of the last iteration (control_samples_d_); needs cfg.save_samples */
mppi_status mppi_get_sampled_controls(mppi_handle h, float* v);

/* ---------------------------------------------------------------- Robust MPPI (MPPI_CONTROLLER_ROBUST) ----------- */
/* Systems of a Robust-MPPI handle: 0 = nominal, 1 = real (robust_mppi_controller.cu:637-640); x0 of
 * mppi_compute_control is the REAL state, mppi_get_control_seq the real system's control (getControlSeq),
 * mppi_get_state_seq the NOMINAL state trajectory (getTargetStateSeq, robust_mppi_controller.cuh:131-134),
 * mppi_get_nominal_control_seq the importance-sampler (nominal) control.  mppi_slide is a no-op there (:178). */
/** value_function_threshold_, num_candidate_nominal_states_ (odd, >= 3), samples per candidate = eval_dyn_kernel_dim_.x
 *  (controllers/R-MPPI/robust_mppi_controller.cuh:46-53; checks of updateNumCandidates, robust_mppi_controller.cu:414-448
 *  become MPPI_ERR_INVALID_ARG) */
mppi_status mppi_set_rmppi_params(mppi_handle h, float value_function_threshold, int num_candidates,
                                  int samples_per_candidate);
/**
 * DDP tracking-controller gains, the reference's DDPFeedbackState::fb_gain_traj_ (feedback_controllers/DDP/ddp.cuh:18-60):
 * gains[T][S][C], entry [t][i][j] = K_t(j, i).  The producer (host DDP / iLQR, include/mppi/ddp/) is the caller's.
 * accumulate_all_states == 0 reproduces DeviceDDPImpl::k (ddp.cu:11-45) including its behaviour for an even
 * CONTROL_DIM (only the last state's gain row takes effect); != 0 sums over all states.
 */
mppi_status mppi_set_feedback_gains(mppi_handle h, const float* gains, int accumulate_all_states);
/**
 * RobustMPPIController::updateImportanceSamplingControl(state, stride) (robust_mppi_controller.cu:548-568): candidate
 * nominal states by line search, init-eval kernel (core/rmppi_kernels.cu:231-356), best candidate by free energy,
 * control histories, slide of the nominal control, nominal state trajectory.  The DDP gain update that ends the
 * reference's function is the caller's (mppi_set_feedback_gains).
 */
mppi_status mppi_update_importance_sampling_control(mppi_handle h, const float* state, int stride);
/** getNominalState, getBestIndex, nominal stride, getCandidateFreeEnergy [num_candidates]; any pointer may be NULL */
mppi_status mppi_get_rmppi_state(mppi_handle h, float* nominal_state, int* best_index, int* nominal_stride,
                                 float* candidate_free_energy);

/* ---------------------------------------------------------------- device-resident iteration loop ----------------- */
/**
 * Runs `num_iterations` passes of the optimisation-loop body (sample -> rollout -> baseline/normExp -> weighted
 * reduction, mean <- u*) back to back on the handle's stream from the CURRENT device mean and initial state, without
 * host round trips (the three blocking D2H copies per iteration of mppi_controller.cu:187-219 do not exist here).
 * synchronize != 0 waits for completion.  This is the unit bench.py times.
 */
mppi_status mppi_optimize(mppi_handle h, int num_iterations, int synchronize);
/** SamplingDistribution::setHostOptimalControlSequence (sampling_distributions/gaussian/gaussian.cu:459-478): copies the
 *  device control means — the raw u* of the last iteration, before smoothing — to u_out[D][T][C] */
mppi_status mppi_get_optimal_control(mppi_handle h, float* u_out);
/** uploads x0 (Vanilla [S]; Tube [2][S]) and the nominal control without running anything */
mppi_status mppi_upload_state(mppi_handle h, const float* x0);
/**
 * HIP-event timing on the handle's own stream: total ms for `num_iterations` iterations and, separately, the summed
 * duration of the rollout kernel launches alone (events recorded around each rollout launch).
 */
mppi_status mppi_time_iterations(mppi_handle h, int num_iterations, float* ms_total, float* ms_rollout_kernels);
/**
 * chooseAppropriateKernel (controllers/MPPI/mppi_controller.cu:44-143): times the rollout kernel structures that fit this
 * configuration — the fused kernel and the role-pipelined one, the counterparts of the reference's "single" and "split"
 * kernels — num_evaluations launches each on the handle's current state, mean and parameters (model blobs must be loaded),
 * and keeps the faster one for all later launches.  The trial launches do not advance the noise stream.  chosen_variant
 * (mppi_kernel_variant), fused_ms, pipeline_ms (time per launch; +inf where a structure does not apply) may be NULL.
 * Without this call the handle runs cfg.kernel_variant (MPPI_KERNEL_AUTO: the pipelined kernel where it exists).
 */
mppi_status mppi_choose_kernel(mppi_handle h, int num_evaluations, int* chosen_variant, float* fused_ms, float* pipeline_ms);
/** waits for everything enqueued on the handle's stream */
mppi_status mppi_synchronize(mppi_handle h);

/* ---------------------------------------------------------------- multi-GPU (K-sharding) ---------- */
/**
 * Device pointers for an external exchange (torch.distributed / RCCL owned by the caller):
 *   *send = this rank's merged record [D][T*C+4]; *recv = buffer for all ranks' records [world][D][T*C+4].
 * Per iteration the caller all-gathers send -> recv on the handle's stream; mppi_optimize does this itself when the
 * handle has an RCCL communicator (mppi_comm_init_rccl).
 */
mppi_status mppi_get_exchange_buffers(mppi_handle h, void** send, void** recv, size_t* floats_per_rank);
/** host-staged variant of the external exchange, for callers whose collective library cannot touch this library's device
 *  memory (e.g. a second ROCm runtime in the process): this rank's record [D][T*C+4] to the host / all ranks' records
 *  [world][D][T*C+4] from the host.  Both synchronise the handle's stream. */
mppi_status mppi_read_send_record(mppi_handle h, float* out);
mppi_status mppi_write_recv_records(mppi_handle h, const float* in);
/** one iteration split around the exchange: local rollout + local merge | (caller's all-gather) | global merge */
mppi_status mppi_iteration_local(mppi_handle h);
mppi_status mppi_iteration_merge(mppi_handle h);
/**
 * P2P mailbox exchange over xGMI — the second stage of SURVEY.md §8e: instead of an ncclAllGather of ~1 KB (10-20 us on a
 * ~30 us iteration) every rank writes its merged record straight into a mailbox in each peer's memory (one kernel, one flag
 * store per peer) and every rank's merge kernel spins on its own flags.  One process per GPU: every rank calls
 * mppi_p2p_mailbox_handle (a 64-byte hipIpcMemHandle_t), the handles are exchanged over any control plane (bench.py: a gloo
 * all-gather), then every rank calls mppi_p2p_connect(handles[world]).  Ranks living in ONE process (several handles, one or
 * more devices) call mppi_p2p_connect_local(peers[world]) instead — hipIpc does not open a handle of its own process.
 * From then on mppi_optimize / mppi_compute_control use the mailbox; RCCL (mppi_comm_init_rccl) stays the fallback.
 * A merge kernel gives up after 2 s without a peer's record; mppi_get_stats then returns MPPI_ERR_COMM.
 * Reconnecting: every rank of the new session calls mppi_p2p_reset (ends the session: flags and records cleared, sequence
 * numbers restart at 1), then exports / connects again before any of them runs its first exchange; mppi_p2p_connect closes the
 * mappings it opened before.  mppi_p2p_mailbox_handle itself has NO side effect on a connected session — it may be called
 * again (a peer that maps late, a caller that simply asks twice) without disturbing records in flight.  The rule is ENFORCED:
 * mppi_p2p_connect on a handle whose session is live or used (sequence numbers != 0) returns MPPI_ERR_STATE instead of
 * restarting the sequence numbers over the old session's flags (the mailbox cannot be cleared at connect time — a peer that
 * connected first may already have posted).  The exchange-failure mark is sticky on the device until a session starts.
 */
mppi_status mppi_p2p_mailbox_handle(mppi_handle h, void* out_bytes, size_t capacity, size_t* nbytes);
mppi_status mppi_p2p_reset(mppi_handle h);
mppi_status mppi_p2p_connect(mppi_handle h, const void* handles, size_t stride_bytes);
mppi_status mppi_p2p_connect_local(mppi_handle h, const mppi_handle* peers);
/** native RCCL path: unique id created on rank 0 and shipped by the caller to every rank (ncclGetUniqueId / ncclCommInitRank) */
mppi_status mppi_rccl_unique_id(void* out_bytes, size_t capacity, size_t* nbytes);
mppi_status mppi_comm_init_rccl(mppi_handle h, const void* unique_id, size_t nbytes);

/** diagnostics: microseconds since the first statement of the last low-latency Vanilla mppi_compute_control at which the host had
 *  [0] written the inputs, [1] enqueued the ingest, [2] the iterations, [3] the merge, [4] the finalize kernel, [5] seen the
 *  control-ready flag, [6] copied the results out; out8[7] is unused (tools/compute_control_host_timing.py) */
mppi_status mppi_debug_host_stamps(mppi_handle h, double* out8);

/* ---------------------------------------------------------------- kernel-level operators ------------------------- */
/* Host-buffer wrappers around single kernels, mirroring the reference's launch wrappers; used by the kernel-level
 * parity tests the way the reference's tests/include/kernel_tests/core harnesses use theirs. */
/** one rollout launch from x0 ([D][S]) and the current nominal control, no mean update: costs via mppi_get_costs
 *  (launchRolloutKernel, core/mppi_common.cu:1299-1325) */
mppi_status mppi_rollout_costs(mppi_handle h, const float* x0, int optimization_stride);
/**
 * Dynamics::enforceConstraints on one control vector u[C] (state[S] may be NULL; only plugins with state-dependent
 * constraints read it): what Controller::getCurrentControl applies before a control is published
 * (controllers/controller.cuh:329-345).  For plugins that keep the base rule (deadband, clamp to the control ranges) this runs
 * on the host and takes no handle lock, so it can be called from a second thread while mppi_compute_control is in flight;
 * every other entry point of a handle is serialised by the handle's own mutex.
 */
mppi_status mppi_enforce_constraints(mppi_handle h, const float* state, float* u);
/** one model step on the device plugin, the simulation step of the reference's examples (examples/cartpole_example.cu:
 *  76-80: model->enforceConstraints(x, u); model->step(x, x_next, xdot, u, y, t, dt)): u is clamped in place when
 *  enforce_constraints != 0, then x <- x_next */
mppi_status mppi_model_step(mppi_handle h, float* x_inout, float* u_inout, float dt, int enforce_constraints);
/** launchNormExpKernel (core/mppi_common.cu:1327-1337): costs[K] -> exp(-lambda_inv (S - baseline)) in place */
mppi_status mppi_norm_exp(float* costs, int num_rollouts, float lambda_inv, float baseline, int device);
/** device two-pass baseline + normaliser (the role of fullGPUcomputeWeights, core/mppi_common.cu:1031-1053);
 *  costs[K] -> weights in place, out2 = {baseline, normalizer} */
mppi_status mppi_compute_weights(float* costs, int num_rollouts, float lambda_inv, float* out2, int device);
/** launchWeightedReductionKernel (core/mppi_common.cu:1366-1388): u_out[T][C] = sum_k (w_k/normalizer) v[k][t][c] */
mppi_status mppi_weighted_reduction(const float* weights, const float* v, float normalizer, int num_rollouts,
                                    int num_timesteps, int control_dim, float* u_out, int device);
/** the reference-order forms of the two (exact_reduce_kernels.hpp; what MPPI_REDUCTION_REFERENCE_ORDER runs): costs[K] ->
 *  w_k = exp(-(S_k - rho)/lambda) in place with the GLOBAL first-minimum rho; stats8 = {rho, eta = float(sum of double(w)) in
 *  index order, free energy mean, variance, modified variance (computeFreeEnergy's serial fp32 sums), sum w^2, 0, 0} */
mppi_status mppi_compute_weights_reference_order(float* costs, int num_rollouts, float lambda, float* stats8, int device);
/** weightedReductionKernel in the reference's summation order (core/mppi_common.cu:1115-1160): cells of sum_stride
 *  consecutive rollouts summed serially with weight = w_k / normalizer, then the cells serially; fma != 0 contracts
 *  inter += weight * v into one fma as nvcc's default does */
mppi_status mppi_weighted_reduction_reference_order(const float* weights, const float* v, float normalizer, int num_rollouts,
                                                    int num_timesteps, int control_dim, int sum_stride, int fma, float* u_out,
                                                    int device);
/** eps[k_begin..k_end)[T][C] exactly as the fused generator draws it (for generator parity tests) */
mppi_status mppi_philox_normal(uint64_t seed, uint32_t generation, int num_rollouts, int num_timesteps,
                               int control_dim, int k_begin, int k_end, float* eps_out, int device);
/**
 * Measurement aid (bench.py's latency model): average time per launch of n dependent launches of a trivial 256-workgroup kernel
 * on a stream of its own, i.e. what a kernel boundary costs on this device (MI355X_MICROARCH.md "boundary": ~1.45 us).
 */
mppi_status mppi_measure_launch_boundary(int device, int n, float* us_per_launch);
/**
 * Measurement aid (bench.py's issue floor): nanoseconds per instruction of a wave that is ALONE on its SIMD and issues 4-byte
 * v_fmac_f32 on eight independent accumulators — the cheapest VALU stream; 256 one-wave workgroups, two lengths differenced.
 * A rollout step cannot cost less than its instruction count on the dynamics wave times this interval.
 */
mppi_status mppi_measure_issue_interval(int device, float* ns_per_instruction);
/** elementwise det_math on the device (func ids as oracle_det_eval): host/device bit-parity test hook */
mppi_status mppi_det_eval(int func, const float* x, float* y, int n, int device);

/** geometry / sampling state of one 2-D map (reference: TextureParams, utils/texture_helpers/texture_helper.cuh:17-63) */
typedef struct mppi_texture2d_params
{
  int address_mode[2];   /* 0 clamp (default), 1 border */
  int filter_mode;       /* 0 linear (default), 1 point */
  float border_color[4];
  float origin[3];
  float rotations[9];    /* row-major 3x3, world -> map */
  float resolution[3];   /* metres per texel */
} mppi_texture2d_params;
/** TwoDTextureHelper lookups on the device (utils/texture_helpers/two_d_texture_helper.hpp; reference:
 *  texture_helper.cu:270-289): data[height][width][channels] (channels 1 or 4), points[n][3], frame 0 = normalised texture
 *  coordinate (queryTexture), 1 = map pose (queryTextureAtMapPose), 2 = world pose (queryTextureAtWorldPose);
 *  out[n][channels] */
mppi_status mppi_texture2d_query(const float* data, int width, int height, int channels, const mppi_texture2d_params* p,
                                 const float* points, int n, int frame, float* out, int device);

#ifdef __cplusplus
}
#endif

#endif /* MPPI_AMD_H_ */
