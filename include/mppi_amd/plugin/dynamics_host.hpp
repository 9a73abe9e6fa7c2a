/**
 * dynamics_host.hpp — the host overloads of Dynamics::enforceConstraints / Dynamics::step (declared in plugin/dynamics.hpp).
 *
 * The reference's models carry a second, Eigen-based implementation of computeDynamics / step for the host
 * (dynamics/dynamics.cuh:300-340; e.g. dynamics/cartpole/cartpole_dynamics.cu:32-60), and its examples integrate their
 * "plant" with it (examples/cartpole_example.cu:76-80).  A model written for this engine has ONE implementation, the device
 * one; these overloads run it for a single rollout — the model object travels to the kernel by value, exactly as it does
 * in the rollout kernels — so the caller's loop keeps its two lines and integrates with the arithmetic the rollouts use.
 * Cost: one tiny kernel and a synchronisation per call (~10 us) — a simulation convenience, never part of the rollout path
 * (a real plant has its own state source; core/base_plant.hpp).  Needs hipcc (it launches a kernel).
 */
#ifndef MPPI_AMD_PLUGIN_DYNAMICS_HOST_HPP_
#define MPPI_AMD_PLUGIN_DYNAMICS_HOST_HPP_

#include <stdexcept>
#include <string>

#include "mppi_amd/plugin/dynamics.hpp"

namespace mppi
{
namespace host
{
/** buf: in [x (S) | u (C)], out [x_next (S) | xdot (S) | y (O) | u after the constraints (C)] */
template <class DYN_T>
__global__ void __launch_bounds__(1) dynamicsHostStepKernel(DYN_T dyn, float* buf, float t, float dt, int mode)
{
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* theta_s = reinterpret_cast<float*>(smem_raw);
  float x[S], xn[S], xdot[S], u[C], y[O];
  for (int i = 0; i < S; i++)
  {
    x[i] = buf[i];
    xn[i] = 0.0f;
    xdot[i] = 0.0f;
  }
  for (int i = 0; i < C; i++)
    u[i] = buf[S + i];
  for (int i = 0; i < O; i++)
    y[i] = 0.0f;
  dyn.initializeDynamics(x, u, y, theta_s, 0.0f, dt);
  if (mode & 1)
    dyn.enforceConstraints(x, u);
  if (mode & 2)
    dyn.step(x, xn, xdot, u, y, theta_s, t, dt);
  for (int i = 0; i < S; i++)
  {
    buf[i] = xn[i];
    buf[S + i] = xdot[i];
  }
  for (int i = 0; i < O; i++)
    buf[2 * S + i] = y[i];
  for (int i = 0; i < C; i++)
    buf[2 * S + O + i] = u[i];
}

/** mode: bit 0 = enforceConstraints, bit 1 = step */
template <class DYN_T>
inline void runOnDevice(DYN_T& dyn, const float* x, float* u, float* xn, float* xdot, float* y, float t, float dt, int mode)
{
  constexpr int S = DYN_T::STATE_DIM, C = DYN_T::CONTROL_DIM, O = DYN_T::OUTPUT_DIM, N = 2 * S + O + C;
  static float* buf_d = nullptr;  // one small buffer per model type (a caller's simulation loop is single-threaded)
  auto ok = [](hipError_t e, const char* what) {
    if (e != hipSuccess)
      throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
  };
  if (!buf_d)
    ok(hipMalloc((void**)&buf_d, sizeof(float) * N), "hipMalloc");
  float host[N];
  for (int i = 0; i < S; i++)
    host[i] = x[i];
  for (int i = 0; i < C; i++)
    host[S + i] = u[i];
  ok(hipMemcpyAsync(buf_d, host, sizeof(float) * (S + C), hipMemcpyHostToDevice, dyn.stream_), "hipMemcpyAsync");
  const size_t smem = (size_t)calcClassSharedMemSize(&dyn, 1);
  hipLaunchKernelGGL((dynamicsHostStepKernel<DYN_T>), dim3(1), dim3(1, 1, 1), smem, dyn.stream_, dyn, buf_d, t, dt, mode);
  ok(hipGetLastError(), "dynamicsHostStepKernel");
  ok(hipMemcpyAsync(host, buf_d, sizeof(float) * N, hipMemcpyDeviceToHost, dyn.stream_), "hipMemcpyAsync");
  ok(hipStreamSynchronize(dyn.stream_), "hipStreamSynchronize");
  for (int i = 0; i < S; i++)
  {
    if (xn)
      xn[i] = host[i];
    if (xdot)
      xdot[i] = host[S + i];
  }
  for (int i = 0; i < O && y; i++)
    y[i] = host[2 * S + i];
  for (int i = 0; i < C; i++)
    u[i] = host[2 * S + O + i];
}
}  // namespace host
}  // namespace mppi

namespace MPPI_internal
{
template <class CLASS_T, class PARAMS_T>
void Dynamics<CLASS_T, PARAMS_T>::enforceConstraints(state_array& state, control_array& control)
{
  mppi::host::runOnDevice(*static_cast<CLASS_T*>(this), state.data(), control.data(), nullptr, nullptr, nullptr, 0.0f, 0.0f, 1);
}

template <class CLASS_T, class PARAMS_T>
void Dynamics<CLASS_T, PARAMS_T>::step(state_array& state, state_array& next_state, state_array& state_der,
                                       control_array& control, output_array& output, const float t, const float dt)
{
  mppi::host::runOnDevice(*static_cast<CLASS_T*>(this), state.data(), control.data(), next_state.data(), state_der.data(),
                          output.data(), t, dt, 2);
}
}  // namespace MPPI_internal

#endif
