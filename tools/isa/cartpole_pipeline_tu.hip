// Minimal translation unit: only the cart-pole role-pipelined rollout kernel, for quick `hipcc -S` iterations on its ISA
// (the full engine takes two minutes to compile).  tools/isa/build_tu.sh compiles it to /tmp/isa/cp.s.
#include "rollout_pipeline_kernel.hpp"
#include "mppi_amd/sampling_distributions/gaussian.hpp"
#include "mppi_amd/dynamics/cartpole/cartpole_dynamics.hpp"
#include "mppi_amd/cost_functions/cartpole/cartpole_quadratic_cost.hpp"

using Sampler = mppi::sampling_distributions::GaussianDistribution<CartpoleDynamicsParams>;
template __global__ void mppi::kernels::rolloutPipelineKernel<CartpoleDynamics, CartpoleQuadraticCost, Sampler, 1, true, false>(
    CartpoleDynamics, CartpoleQuadraticCost, Sampler, const mppi::kernels::RolloutArgs);
