/**
 * host_arrays.hpp — the small fixed-size host containers the templated controller classes hand around.
 *
 * The reference's host API speaks Eigen (state_array = Eigen::Matrix<float, STATE_DIM, 1>, control_trajectory =
 * Eigen::Matrix<float, CONTROL_DIM, MAX_TIMESTEPS>, column-major: dynamics/dynamics.cuh:80-90, controllers/controller.cuh:
 * 90-100).  Eigen is a host-side convenience there, not part of the rollout path, and it is not a dependency here: these
 * two classes offer the handful of members the reference's examples and plant code use on those types — Zero(), data(),
 * operator[], operator()(i) / (i, j), col(j), block(i, j, rows, 1), setZero(), size() — with the SAME memory layout
 * (column-major, a column = one time step), so `Eigen::Map<...>(x.data())` on a caller's side sees the reference's matrix.
 */
#ifndef MPPI_AMD_PLUGIN_HOST_ARRAYS_HPP_
#define MPPI_AMD_PLUGIN_HOST_ARRAYS_HPP_

#include <cstddef>
#include <initializer_list>
#include <ostream>

namespace mppi
{
namespace host
{
/** N x 1 column (Eigen::Matrix<float, N, 1>) */
template <int N>
struct Array
{
  float v[N > 0 ? N : 1];
  static Array Zero()
  {
    Array a;
    a.setZero();
    return a;
  }
  static Array Constant(float c)
  {
    Array a;
    for (int i = 0; i < N; i++)
      a.v[i] = c;
    return a;
  }
  Array()
  {
  }
  Array(std::initializer_list<float> init)
  {
    setZero();
    int i = 0;
    for (float f : init)
      if (i < N)
        v[i++] = f;
  }
  void setZero()
  {
    for (int i = 0; i < N; i++)
      v[i] = 0.0f;
  }
  float& operator[](int i)
  {
    return v[i];
  }
  const float& operator[](int i) const
  {
    return v[i];
  }
  float& operator()(int i)
  {
    return v[i];
  }
  const float& operator()(int i) const
  {
    return v[i];
  }
  float* data()
  {
    return v;
  }
  const float* data() const
  {
    return v;
  }
  static constexpr int size()
  {
    return N;
  }
  static constexpr int rows()
  {
    return N;
  }
  static constexpr int cols()
  {
    return 1;
  }
};

template <int N>
inline std::ostream& operator<<(std::ostream& os, const Array<N>& a)
{
  for (int i = 0; i < N; i++)
    os << a.v[i] << (i + 1 < N ? "\n" : "");
  return os;
}

/** R x C column-major matrix (Eigen::Matrix<float, R, C>): column j = time step j of a trajectory */
template <int R, int C>
struct Matrix
{
  float v[(size_t)(R > 0 ? R : 1) * (C > 0 ? C : 1)];
  static Matrix Zero()
  {
    Matrix m;
    m.setZero();
    return m;
  }
  void setZero()
  {
    for (size_t i = 0; i < (size_t)R * C; i++)
      v[i] = 0.0f;
  }
  float& operator()(int i, int j)
  {
    return v[(size_t)j * R + i];
  }
  const float& operator()(int i, int j) const
  {
    return v[(size_t)j * R + i];
  }
  Array<R> col(int j) const
  {
    Array<R> a;
    for (int i = 0; i < R; i++)
      a.v[i] = v[(size_t)j * R + i];
    return a;
  }
  void setCol(int j, const Array<R>& a)
  {
    for (int i = 0; i < R; i++)
      v[(size_t)j * R + i] = a.v[i];
  }
  /** the one block shape the reference's callers take from a trajectory: `p` rows of one column (p == R: the time step) */
  Array<R> block(int i, int j, int p, int q) const
  {
    Array<R> a = Array<R>::Zero();
    for (int r = 0; r < p && r + i < R; r++)
      a.v[r] = v[(size_t)j * R + i + r];
    (void)q;
    return a;
  }
  float* data()
  {
    return v;
  }
  const float* data() const
  {
    return v;
  }
  static constexpr int rows()
  {
    return R;
  }
  static constexpr int cols()
  {
    return C;
  }
  static constexpr int size()
  {
    return R * C;
  }
};
}  // namespace host
}  // namespace mppi

#endif
