/**
 * det_math.h — bit-reproducible fp32 elementary functions for the MPPI hot path.
 *
 * WHY THIS EXISTS
 * The softmin weights of MPPI are w_k = exp(-(S_k - rho)/lambda).  A 1-ulp difference in a trajectory cost
 * S_k ~ O(1e3) is ~1e-4 absolute, i.e. a ~4e-4 *relative* change of w_k at lambda = 0.25.  The parity bar of this
 * project (control sequence L-inf <= 1e-5 against the CPU oracle, BASELINE.json) can therefore only be met if the
 * trajectory costs of the HIP engine and of the CPU oracle agree essentially bit for bit.  IEEE-754 add / mul / fma /
 * div / sqrt are correctly rounded on both x86-64 and gfx950 (hipcc keeps -fhip-fp32-correctly-rounded-divide-sqrt
 * on by default and keeps f32 denormals), but libm's sinf/cosf/expf/tanhf (glibc) and the device library's (ocml)
 * differ in the last bit on a few per cent of inputs, and the reference's own device path uses the even looser
 * __sinf/__cosf fast intrinsics (reference: include/mppi/dynamics/cartpole/cartpole_dynamics.cu:91-93 versus the
 * host overload :52-55; include/mppi/utils/activation_functions.cuh:17-29, 49-59).
 *
 * So every transcendental on the hot path goes through the functions in this header.  They are written with
 * nothing but IEEE fp32 +,-,*,/ , sqrt, explicit fmaf and integer bit operations, in ONE fixed operation order.
 * Both translation environments must be compiled with -ffp-contract=off so that the only fused operations are the
 * explicit det::fma() calls (hipcc: v_fma_f32; gcc -mfma: vfmadd, or glibc's correctly rounded fmaf()).
 * Result: identical bits on the MI355X and on the host, which tests/test_det_math.py checks on the GPU.
 *
 * Accuracy (checked against float64 libm in tests/test_det_math.py): sin/cos/exp/log <= 2 ulp, atan <= 3 ulp, tanh
 * <= 7 ulp (4e-7 absolute; branch-free rational form) on the ranges the models use.  That is tighter than the reference's own device intrinsics, and well inside the
 * reference's GPU-vs-CPU test tolerances (tests/mppi_core/rollout_kernel_tests.cu:258: 1e-4 relative).
 *
 * Polynomial coefficients are the classic single-precision Cephes minimax sets (S. Moshier, netlib cephes/single).
 *
 * This header is part of the public boundary (include/) and is used by the HIP engine AND by the CPU oracle; it is
 * pure arithmetic with no device or host dependency.
 */
#ifndef MPPI_AMD_DET_MATH_H_
#define MPPI_AMD_DET_MATH_H_

#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define MPPI_HD __host__ __device__
#else
#define MPPI_HD
#endif

namespace mppi
{
namespace det
{
/** The one fused operation both sides share: round(a*b + c) with a single rounding. */
MPPI_HD static inline float fma(float a, float b, float c)
{
  return __builtin_fmaf(a, b, c);
}

MPPI_HD static inline uint32_t f2u(float f)
{
  uint32_t u;
  memcpy(&u, &f, sizeof(u));
  return u;
}
MPPI_HD static inline float u2f(uint32_t u)
{
  float f;
  memcpy(&f, &u, sizeof(f));
  return f;
}

/** IEEE correctly rounded on both sides. */
MPPI_HD static inline float sqrt(float x)
{
  return __builtin_sqrtf(x);
}
/** Round to nearest even; exact on both sides (v_rndne_f32 / roundss). */
MPPI_HD static inline float rint(float x)
{
  return __builtin_rintf(x);
}
MPPI_HD static inline float trunc(float x)
{
  return __builtin_truncf(x);
}
MPPI_HD static inline float fabs(float x)
{
  return __builtin_fabsf(x);
}
MPPI_HD static inline float copysign(float mag, float sgn)
{
  return __builtin_copysignf(mag, sgn);
}

/**
 * Exact fmodf(a, b) for b > 0 (C99 semantics: result has the sign of a, |result| < b), given rb = (float)(1/b).
 * q = trunc(a * rb) is within one of the true quotient; fma(-q, b, a) is then exact because the true remainder is
 * representable, and one conditional +-b fixes the off-by-one in either direction.  No division on the fast path.
 * Falls back to the (also exact) library fmodf when the quotient is too large for that argument.
 * Used by normalizeAngle (reference: utils/angle_utils.cuh:21-27).
 */
MPPI_HD static inline float fmod_rb(float a, float b, float rb)
{
  /* fmod(a, b) = copysign(fmod(|a|, b), a): work on |a| so that the off-by-one fix-up is two selects */
  const float aa = fabs(a);
  const float q = trunc(aa * rb);
  if (!(q < 2097152.0f))
  {
    return ::fmodf(a, b);
  }
  float r = fma(-q, b, aa);
  const float rp = r + b, rm = r - b;
  r = (r < 0.0f) ? rp : ((r >= b) ? rm : r);
  return copysign(r, a);
}
MPPI_HD static inline float fmod(float a, float b)
{
  return fmod_rb(a, b, 1.0f / b);
}

#define MPPI_DET_PI 3.14159274101257324219f     /* (float)pi, same value as glibc's M_PIf32 used by the reference */
#define MPPI_DET_TWO_PI 6.28318548202514648438f /* 2*(float)pi, exact doubling */

/**
 * normalizeAngle for |angle| < 1e7 rad (q = trunc(|a| / 2 pi) < 2^21): the fast path of fmod_rb without the test for the
 * library fallback — bit-identical to normalizeAngle() on that range, and six instructions shorter on the dependent
 * chain of a rollout step.  For models whose angles are physically bounded (a pole does not spin 10^6 times per horizon).
 */
MPPI_HD static inline float normalizeAngleBounded(float angle)
{
  const float a = angle + MPPI_DET_PI;
  const float aa = fabs(a);
  const float q = trunc(aa * 0.15915494309189534561f);
  float r = fma(-q, MPPI_DET_TWO_PI, aa);
  const float rp = r + MPPI_DET_TWO_PI, rm = r - MPPI_DET_TWO_PI;
  r = (r < 0.0f) ? rp : ((r >= MPPI_DET_TWO_PI) ? rm : r);
  const float result = copysign(r, a);
  if (result <= 0.0f)
    return result + MPPI_DET_PI;
  return result - MPPI_DET_PI;
}

/** Reference: include/mppi/utils/angle_utils.cuh:21-27 (float overload), same branch structure. */
MPPI_HD static inline float normalizeAngle(float angle)
{
  const float result = fmod_rb(angle + MPPI_DET_PI, MPPI_DET_TWO_PI, 0.15915494309189534561f);
  if (result <= 0.0f)
    return result + MPPI_DET_PI;
  return result - MPPI_DET_PI;
}

/**
 * sin and cos of x in one go.  Cody-Waite reduction by pi/2 with a 3-term split and fused steps, Cephes sinf/cosf
 * kernels on [-pi/4, pi/4].  Accurate to <= 2 ulp for |x| <= 1e4; beyond that the reduction slowly degrades (still
 * deterministic).  Non-finite input returns NaN.
 */
MPPI_HD static inline void sincos(float x, float* s_out, float* c_out)
{
#if defined(MPPI_DET_MATH_LIBM) && !defined(__HIPCC__)
  /* HOST-ONLY study switch (oracle/Makefile target libm, tests/test_det_math.py::test_libm_flavour_deviation): the oracle
   * rebuilt on libm's sinf / cosf / expf / logf / tanhf / atanf measures how far the bit-reproducible functions of this
   * header move a control sequence away from a build on the platform's own math library — the distance that separates ANY
   * two implementations of the reference's formulas (its CUDA device path uses __sinf / __cosf / tanhf, none of them
   * correctly rounded).  Never defined for the product. */
  *s_out = ::sinf(x);
  *c_out = ::cosf(x);
  return;
#endif
  const float k = rint(x * 0.636619746685028076171875f);   /* x * 2/pi */
  float r = fma(-k, 1.57079637050628662109375f, x);        /* pi/2 hi  */
  r = fma(-k, -4.37113882867379277013e-08f, r);            /* pi/2 mid */
  r = fma(-k, -1.71512451000588187e-15f, r);               /* pi/2 lo  */
  const float z = r * r;
  /* sin(r) = r + r*z*(S2 + z*(S1 + z*S0)) */
  float ps = fma(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = fma(ps, z, -1.6666654611e-1f);
  const float sr = fma(ps * z, r, r);
  /* cos(r) = 1 - z/2 + z*z*(C2 + z*(C1 + z*C0)) */
  float pc = fma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = fma(pc, z, 4.166664568298827e-2f);
  const float cr = fma(pc * z, z, fma(-0.5f, z, 1.0f));
  /* quadrant: k mod 4 (k is integral; the int conversion is exact for |k| < 2^31, saturating beyond) */
  const uint32_t q = (uint32_t)(int)k;
  const float s = (q & 1u) ? cr : sr;
  const float c = (q & 1u) ? sr : cr;
  /* sign: s = -s when q & 2, c = -c when (q + 1) & 2 — as a flip of the sign bit (bit 1 of q, resp. of q + 1, moved to
   * bit 31): a shift, a mask and an xor instead of compare + select, whose VCC round trip costs a single wave three
   * issue slots on gfx950 (tools/ubench/op_latency.hip) */
  *s_out = u2f(f2u(s) ^ ((q << 30) & 0x80000000u));
  *c_out = u2f(f2u(c) ^ (((q << 30) + 0x40000000u) & 0x80000000u));
}
MPPI_HD static inline float sin(float x)
{
  float s, c;
  sincos(x, &s, &c);
  return s;
}
/** tan(x) = sin / cos with one correctly rounded division: <= 4 ulp for |x| <= 1e4 away from the poles */
MPPI_HD static inline float tan(float x)
{
  float s, c;
  sincos(x, &s, &c);
  return s / c;
}
MPPI_HD static inline float cos(float x)
{
  float s, c;
  sincos(x, &s, &c);
  return c;
}

/** 2^n * z for integer-valued n in [-252, 254] without libm (two-step scaling keeps subnormal results right). */
MPPI_HD static inline float scalbn_small(float z, int n)
{
  if (n > 127)
  {
    z *= 1.7014118346046923e38f; /* 2^127 */
    n -= 127;
  }
  else if (n < -126)
  {
    z *= 1.1754943508222875e-38f; /* 2^-126 */
    n += 126;
  }
  return z * u2f((uint32_t)(n + 127) << 23);
}

/** exp(x): n = rint(x*log2(e)); r = x - n*ln2 (3-term, fused); Cephes expf kernel; exact 2^n scaling. */
MPPI_HD static inline float exp(float x)
{
#if defined(MPPI_DET_MATH_LIBM) && !defined(__HIPCC__)
  return ::expf(x);
#endif
  if (!(x > -104.0f))
    return (x != x) ? x : 0.0f;
  if (x > 88.72283935546875f)
    return u2f(0x7f800000u);
  const float n = rint(x * 1.44269502162933349609375f);
  float r = fma(-n, 0.693115234375f, x);
  r = fma(-n, 3.194063901901245e-05f, r);
  r = fma(-n, 5.5459263847978946e-09f, r);
  float p = fma(1.9875691500e-4f, r, 1.3981999507e-3f);
  p = fma(p, r, 8.3334519073e-3f);
  p = fma(p, r, 4.1665795894e-2f);
  p = fma(p, r, 1.6666665459e-1f);
  p = fma(p, r, 5.0000001201e-1f);
  const float z = fma(p * r, r, r) + 1.0f;
  return scalbn_small(z, (int)n);
}

/** log(x), Cephes logf structure (frexp by bit manipulation, sqrt(1/2) split, degree-8 kernel). */
MPPI_HD static inline float log(float x)
{
#if defined(MPPI_DET_MATH_LIBM) && !defined(__HIPCC__)
  return ::logf(x);
#endif
  if (!(x > 0.0f))
    return (x == 0.0f) ? -u2f(0x7f800000u) : u2f(0x7fc00000u);
  if (x == u2f(0x7f800000u))
    return x;
  int e = 0;
  uint32_t ix = f2u(x);
  if (ix < 0x00800000u)
  { /* subnormal: scale up by 2^23 (exact) */
    x *= 8388608.0f;
    ix = f2u(x);
    e = -23;
  }
  e += (int)(ix >> 23) - 126;
  float m = u2f((ix & 0x007fffffu) | 0x3f000000u); /* mantissa in [0.5, 1) */
  if (m < 0.707106781186547524f)
  {
    e -= 1;
    m = (m + m) - 1.0f;
  }
  else
  {
    m = m - 1.0f;
  }
  const float z = m * m;
  float p = fma(7.0376836292e-2f, m, -1.1514610310e-1f);
  p = fma(p, m, 1.1676998740e-1f);
  p = fma(p, m, -1.2420140846e-1f);
  p = fma(p, m, 1.4249322787e-1f);
  p = fma(p, m, -1.6668057665e-1f);
  p = fma(p, m, 2.0000714765e-1f);
  p = fma(p, m, -2.4999993993e-1f);
  p = fma(p, m, 3.3333331174e-1f);
  const float fe = (float)e;
  float y = (p * m) * z;
  y = fma(-2.12194440e-4f, fe, y);
  y = fma(-0.5f, z, y);
  float r = m + y;
  r = fma(0.693359375f, fe, r);
  return r;
}

/**
 * p / q for the rational kernels below, where q is known to sit in a benign range (here q in [0.0049, 0.91]) so that no
 * operand scaling is ever needed.  On the host this is the IEEE division.  On gfx950 the compiler's own expansion of an
 * fp32 division is v_div_scale x2, v_rcp, 4 fma + 1 mul, v_div_fmas, v_div_fixup; the scale / fixup steps only act on
 * extreme exponents, so in this range the SAME Newton-Raphson core written out by hand —
 *     r = rcp(q); e = fma(-q, r, 1); r = fma(e, r, r); y = p r; y = fma(fma(-q, y, p), r, y); y = fma(fma(-q, y, p), r, y)
 * — returns the same correctly rounded quotient with three instructions fewer, and every step but the seed is an fma
 * that has a packed two-wide form (v_pk_fma_f32), which tanh2() below uses.
 * tests/test_gpu_ops.py::test_det_math_device_equals_host_bitwise checks device == host bit for bit.
 */
MPPI_HD static inline float div_benign(float p, float q)
{
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(q);
  const float e = fma(-q, r, 1.0f);
  r = fma(e, r, r);
  float y = p * r;
  y = fma(fma(-q, y, p), r, y);
  y = fma(fma(-q, y, p), r, y);
  return y;
#else
  return p / q;
#endif
}

/** 1 / x for x in a benign range (|x| in [2^-60, 2^60]): the correctly rounded reciprocal, see div_benign */
MPPI_HD static inline float rcp_benign(float x)
{
  return div_benign(1.0f, x);
}

/** Two rcp_benign at once (bit-identical to two scalar calls): the six fma steps issue as packed v_pk_fma_f32, which
 *  take one issue slot for the pair — a lone wave issues one VALU instruction every ~2 ns whether or not it depends on
 *  the previous one (tools/ubench/op_latency.hip), so instruction COUNT is what a rollout step costs. */
MPPI_HD static inline void rcp_benign2(const float xa, const float xb, float* ra, float* rb)
{
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 q = { xa, xb };
  const f32x2 one = { 1.0f, 1.0f };
  f32x2 r = { __builtin_amdgcn_rcpf(xa), __builtin_amdgcn_rcpf(xb) };
  const f32x2 e = __builtin_elementwise_fma(-q, r, one);
  r = __builtin_elementwise_fma(e, r, r);
  f32x2 y = r; /* 1.0f * r */
  y = __builtin_elementwise_fma(__builtin_elementwise_fma(-q, y, one), r, y);
  y = __builtin_elementwise_fma(__builtin_elementwise_fma(-q, y, one), r, y);
  *ra = y.x;
  *rb = y.y;
#else
  *ra = 1.0f / xa;
  *rb = 1.0f / xb;
#endif
}

/* tanh's rational: Eigen's coefficients times 2^-8 / 4.89352518554385e-03, the constant terms of P and Q both 2^-8 */
#define MPPI_DET_TANH_C0 0.00390625f
#define MPPI_DET_TANH_P1 5.086935125291348e-04f
#define MPPI_DET_TANH_P2 1.1859759069920983e-05f
#define MPPI_DET_TANH_P3 4.088866845108896e-08f
#define MPPI_DET_TANH_P4 -6.868667440373954e-11f
#define MPPI_DET_TANH_P5 1.5966473477548038e-13f
#define MPPI_DET_TANH_P6 -2.2037797495707987e-16f
#define MPPI_DET_TANH_Q1 1.8107749056071043e-03f
#define MPPI_DET_TANH_Q2 9.462016896577552e-05f
#define MPPI_DET_TANH_Q3 9.565081882101367e-07f

/**
 * tanh(x), branch-free and select-free: clamp to +-7.9053 (tanh rounds to +-1 in fp32 beyond), then the odd rational minimax
 * x * P6(x^2) / Q3(x^2) (coefficient set of Eigen's generic_fast_tanh_float, MathFunctionsImpl.h, MPL2, RESCALED so that
 * P(0) = Q(0) = 2^-8 exactly), one correctly rounded division.  No divergent control flow — the NN dynamics evaluate 64 of
 * these per rollout and step, and a two-branch exp-based form costs both branches on a SIMD machine.
 *
 * The common constant term makes tanh(x) = x EXACT for small arguments without a compare / select pair (round 3; the select
 * was 2 of the 13 issue slots of a packed tanh): for |x| < 3.6e-4 both polynomials round to 2^-8, p = x * 2^-8 and the
 * quotient p / 2^-8 are exact scalings — also for -0 and, up to the rounding of x * 2^-8, for subnormals, identically on
 * the host and in div_benign's Newton-Raphson core (a ZERO result is +0 for either sign of x, see below).  Accuracy: <= 6 ulp (4e-7 absolute) against float64 tanh (was 5 with
 * the original constants); +-1 for |x| >= 7.9053.
 * NaN: the clamp is fminf(fmaxf(x, -c), c) / v_med3_f32, both of which return the non-NaN bound, so tanh(NaN) = tanh(-c)
 * = -1 on host and device alike — a NaN pre-activation no longer propagates through the activation (it does through
 * every other path of a step: the state that produced it stays NaN, and the costs of such a rollout are clamped by the
 * cost plugins as before).  Callers that need tanhf's NaN can test their argument.
 */
MPPI_HD static inline float tanh(float x)
{
#if defined(MPPI_DET_MATH_LIBM) && !defined(__HIPCC__)
  return ::tanhf(x);
#endif
  const float xc = fminf(fmaxf(x, -7.90531110763549805f), 7.90531110763549805f);
  const float x2 = xc * xc;
  float p = fma(x2, MPPI_DET_TANH_P6, MPPI_DET_TANH_P5);
  p = fma(x2, p, MPPI_DET_TANH_P4);
  p = fma(x2, p, MPPI_DET_TANH_P3);
  p = fma(x2, p, MPPI_DET_TANH_P2);
  p = fma(x2, p, MPPI_DET_TANH_P1);
  p = fma(x2, p, MPPI_DET_TANH_C0);
  p = xc * p;
  float q = fma(x2, MPPI_DET_TANH_Q3, MPPI_DET_TANH_Q2);
  q = fma(x2, q, MPPI_DET_TANH_Q1);
  q = fma(x2, q, MPPI_DET_TANH_C0);
#if defined(__HIP_DEVICE_COMPILE__)
  return div_benign(p, q);
#else
  /* a zero quotient (x = -0, or |x| < 2^-141 where x * 2^-8 underflows) comes out of the device's Newton-Raphson core as
   * +0 whatever its sign: its exact residual fma(-q, y, p) is +0 and y + 0 * r rounds -0 + +0 to +0.  The host's IEEE quotient
   * would keep the sign; adding +0 gives it the device's value (and changes no other result). */
  return div_benign(p, q) + 0.0f;
#endif
}

/**
 * Two tanh at once: the same operations as tanh() on each argument (so the results are bit-identical to two scalar
 * calls), arranged as two-wide vectors so that gfx950 issues them as packed fp32 instructions (v_pk_mul_f32 /
 * v_pk_fma_f32: two lanes' worth of work per issue slot).  The NN dynamics spend most of their VALU time here.
 */
MPPI_HD static inline void tanh2(const float xa, const float xb, float* ra, float* rb)
{
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 xc;
  xc.x = fminf(fmaxf(xa, -7.90531110763549805f), 7.90531110763549805f);
  xc.y = fminf(fmaxf(xb, -7.90531110763549805f), 7.90531110763549805f);
  const f32x2 x2 = xc * xc;
#define MPPI_DET_PKFMA(a, b, c) __builtin_elementwise_fma((a), (b), (c))
#define MPPI_DET_SPLAT(v) (f32x2{ (v), (v) })
  f32x2 p = MPPI_DET_PKFMA(x2, MPPI_DET_SPLAT(MPPI_DET_TANH_P6), MPPI_DET_SPLAT(MPPI_DET_TANH_P5));
  p = MPPI_DET_PKFMA(x2, p, MPPI_DET_SPLAT(MPPI_DET_TANH_P4));
  p = MPPI_DET_PKFMA(x2, p, MPPI_DET_SPLAT(MPPI_DET_TANH_P3));
  p = MPPI_DET_PKFMA(x2, p, MPPI_DET_SPLAT(MPPI_DET_TANH_P2));
  p = MPPI_DET_PKFMA(x2, p, MPPI_DET_SPLAT(MPPI_DET_TANH_P1));
  p = MPPI_DET_PKFMA(x2, p, MPPI_DET_SPLAT(MPPI_DET_TANH_C0));
  p = xc * p;
  f32x2 q = MPPI_DET_PKFMA(x2, MPPI_DET_SPLAT(MPPI_DET_TANH_Q3), MPPI_DET_SPLAT(MPPI_DET_TANH_Q2));
  q = MPPI_DET_PKFMA(x2, q, MPPI_DET_SPLAT(MPPI_DET_TANH_Q1));
  q = MPPI_DET_PKFMA(x2, q, MPPI_DET_SPLAT(MPPI_DET_TANH_C0));
  f32x2 r = { __builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y) };
  const f32x2 e = MPPI_DET_PKFMA(-q, r, MPPI_DET_SPLAT(1.0f));
  r = MPPI_DET_PKFMA(e, r, r);
  f32x2 y = p * r;
  y = MPPI_DET_PKFMA(MPPI_DET_PKFMA(-q, y, p), r, y);
  y = MPPI_DET_PKFMA(MPPI_DET_PKFMA(-q, y, p), r, y);
#undef MPPI_DET_PKFMA
#undef MPPI_DET_SPLAT
  *ra = y.x;
  *rb = y.y;
#else
  *ra = tanh(xa);
  *rb = tanh(xb);
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
/**
 * tanh of NP packed pairs, evaluated in LOCKSTEP: every Horner / Newton step is issued for all pairs (and for the P and
 * Q polynomials alternately) before the next step.  Same operations per element as tanh() — the results are bit-identical —
 * but no two consecutive instructions depend on each other.  On gfx950 a packed-fp32 result may not be read by the very
 * next instruction: the compiler fills that wait state with an `s_nop 0`, and a wave that is alone on its SIMD pays an
 * issue slot for it (engine_operators.hip's issue probe: 3.4 ns per dependent v_fmac + s_nop pair against 1.7 ns per independent
 * v_fmac).  Evaluated pair after pair, the Newton-Raphson tail of every tanh2() is such a chain: 82 s_nop per AutoRally
 * step.  (Round 2 tried this form and saw no gain — the cost waves' relay chain was as long as the dynamics waves then and
 * hid it; see rollout_pipeline_kernel.hpp.)
 */
template <int NP>
__device__ inline void tanh_pairs(float (&v)[2 * NP])
{
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MPPI_DET_PKFMA(a, b, c) __builtin_elementwise_fma((a), (b), (c))
#define MPPI_DET_SPLAT(c) (f32x2{ (c), (c) })
#define MPPI_DET_ALL for (int k = 0; k < NP; k++)
  f32x2 xc[NP], x2[NP], p[NP], q[NP], r[NP], y[NP];
#pragma unroll
  MPPI_DET_ALL
  {
    xc[k].x = fminf(fmaxf(v[2 * k], -7.90531110763549805f), 7.90531110763549805f);
    xc[k].y = fminf(fmaxf(v[2 * k + 1], -7.90531110763549805f), 7.90531110763549805f);
  }
#pragma unroll
  MPPI_DET_ALL x2[k] = xc[k] * xc[k];
#pragma unroll
  MPPI_DET_ALL
  {
    p[k] = MPPI_DET_PKFMA(x2[k], MPPI_DET_SPLAT(MPPI_DET_TANH_P6), MPPI_DET_SPLAT(MPPI_DET_TANH_P5));
    q[k] = MPPI_DET_PKFMA(x2[k], MPPI_DET_SPLAT(MPPI_DET_TANH_Q3), MPPI_DET_SPLAT(MPPI_DET_TANH_Q2));
  }
#pragma unroll
  MPPI_DET_ALL
  {
    p[k] = MPPI_DET_PKFMA(x2[k], p[k], MPPI_DET_SPLAT(MPPI_DET_TANH_P4));
    q[k] = MPPI_DET_PKFMA(x2[k], q[k], MPPI_DET_SPLAT(MPPI_DET_TANH_Q1));
  }
#pragma unroll
  MPPI_DET_ALL
  {
    p[k] = MPPI_DET_PKFMA(x2[k], p[k], MPPI_DET_SPLAT(MPPI_DET_TANH_P3));
    q[k] = MPPI_DET_PKFMA(x2[k], q[k], MPPI_DET_SPLAT(MPPI_DET_TANH_C0));
  }
#pragma unroll
  MPPI_DET_ALL
  {
    p[k] = MPPI_DET_PKFMA(x2[k], p[k], MPPI_DET_SPLAT(MPPI_DET_TANH_P2));
    r[k] = f32x2{ __builtin_amdgcn_rcpf(q[k].x), __builtin_amdgcn_rcpf(q[k].y) };
  }
#pragma unroll
  MPPI_DET_ALL
  {
    p[k] = MPPI_DET_PKFMA(x2[k], p[k], MPPI_DET_SPLAT(MPPI_DET_TANH_P1));
    y[k] = MPPI_DET_PKFMA(-q[k], r[k], MPPI_DET_SPLAT(1.0f));  // e
  }
#pragma unroll
  MPPI_DET_ALL
  {
    p[k] = MPPI_DET_PKFMA(x2[k], p[k], MPPI_DET_SPLAT(MPPI_DET_TANH_C0));
    r[k] = MPPI_DET_PKFMA(y[k], r[k], r[k]);
  }
#pragma unroll
  MPPI_DET_ALL p[k] = xc[k] * p[k];
#pragma unroll
  MPPI_DET_ALL y[k] = p[k] * r[k];
#pragma unroll
  MPPI_DET_ALL x2[k] = MPPI_DET_PKFMA(-q[k], y[k], p[k]);  // x2 is free: the residual
#pragma unroll
  MPPI_DET_ALL y[k] = MPPI_DET_PKFMA(x2[k], r[k], y[k]);
#pragma unroll
  MPPI_DET_ALL x2[k] = MPPI_DET_PKFMA(-q[k], y[k], p[k]);
#pragma unroll
  MPPI_DET_ALL y[k] = MPPI_DET_PKFMA(x2[k], r[k], y[k]);
#pragma unroll
  MPPI_DET_ALL
  {
    v[2 * k] = y[k].x;
    v[2 * k + 1] = y[k].y;
  }
#undef MPPI_DET_ALL
#undef MPPI_DET_PKFMA
#undef MPPI_DET_SPLAT
}
#endif

/** tanh of N values in place, pairwise through tanh2() */
template <int N>
MPPI_HD static inline void tanh_n(float (&v)[N])
{
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2)
    tanh2(v[i], v[i + 1], &v[i], &v[i + 1]);
  if (N & 1)
    v[N - 1] = tanh(v[N - 1]);
}

/** tanh of N values in place with the packed pairs evaluated in lockstep on the device (tanh_pairs): the same bits as
 *  tanh_n, more live registers (7 pair-registers per pair), no dependent back-to-back packed instructions */
template <int N>
MPPI_HD static inline void tanh_n_lockstep(float (&v)[N])
{
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (N >= 4)
  {
    float w[2 * (N / 2)];
#pragma unroll
    for (int i = 0; i < 2 * (N / 2); i++)
      w[i] = v[i];
    tanh_pairs<N / 2>(w);
#pragma unroll
    for (int i = 0; i < 2 * (N / 2); i++)
      v[i] = w[i];
    if (N & 1)
      v[N - 1] = tanh(v[N - 1]);
  }
  else
    tanh_n<N>(v);
#else
  tanh_n<N>(v);
#endif
}

/** Device flavour of the reference's sigmoid (utils/activation_functions.cuh:49-59): (1 + tanh(x/2))/2. */
MPPI_HD static inline float sigmoid(float x)
{
  return (1.0f + tanh(x / 2.0f)) / 2.0f;
}

/** sigmoid of N values in place: the operations of sigmoid() with the tanh evaluated pairwise */
template <int N>
MPPI_HD static inline void sigmoid_n(float (&v)[N])
{
#pragma unroll
  for (int i = 0; i < N; i++)
    v[i] = v[i] / 2.0f;
  tanh_n<N>(v);
#pragma unroll
  for (int i = 0; i < N; i++)
    v[i] = (1.0f + v[i]) / 2.0f;
}

/** sigmoid_n with the tanh evaluated in lockstep (tanh_n_lockstep): the same bits.  (1 + t) / 2 is evaluated as
 *  fma(t, 0.5, 0.5): halving is exact, so RN(1 + t) / 2 == RN((1 + t) / 2) == RN(0.5 t + 0.5) — one instruction instead of
 *  two, and on the device two values per packed instruction like the halving of the argument. */
template <int N>
MPPI_HD static inline void sigmoid_n_lockstep(float (&v)[N])
{
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2)
  {
    const f32x2 h = f32x2{ v[i], v[i + 1] } * f32x2{ 0.5f, 0.5f };
    v[i] = h.x;
    v[i + 1] = h.y;
  }
  if (N & 1)
    v[N - 1] = v[N - 1] * 0.5f;
  tanh_n_lockstep<N>(v);
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2)
  {
    const f32x2 s2 = __builtin_elementwise_fma(f32x2{ v[i], v[i + 1] }, f32x2{ 0.5f, 0.5f }, f32x2{ 0.5f, 0.5f });
    v[i] = s2.x;
    v[i + 1] = s2.y;
  }
  if (N & 1)
    v[N - 1] = fma(v[N - 1], 0.5f, 0.5f);
#else
#pragma unroll
  for (int i = 0; i < N; i++)
    v[i] = v[i] / 2.0f;
  tanh_n_lockstep<N>(v);
#pragma unroll
  for (int i = 0; i < N; i++)
    v[i] = (1.0f + v[i]) / 2.0f;
#endif
}

/** atan(x), Cephes atanf structure. */
MPPI_HD static inline float atan(float x)
{
#if defined(MPPI_DET_MATH_LIBM) && !defined(__HIPCC__)
  return ::atanf(x);
#endif
  const float ax = fabs(x);
  float y, t;
  if (ax > 2.414213562373095f)
  {
    y = 1.57079637050628662109375f;
    t = -1.0f / ax;
  }
  else if (ax > 0.4142135623730950f)
  {
    y = 0.785398185253143310546875f;
    t = (ax - 1.0f) / (ax + 1.0f);
  }
  else
  {
    y = 0.0f;
    t = ax;
  }
  const float z = t * t;
  float p = fma(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = fma(p, z, 1.99777106478e-1f);
  p = fma(p, z, -3.33329491539e-1f);
  y += fma(p * z, t, t);
  return (x != x) ? x : copysign(y, x);
}

/** asin(x), Cephes asinf structure: a polynomial in x^2 on |x| <= 1/2, asin(x) = pi/2 - 2 asin(sqrt((1 - |x|) / 2)) above.
 *  |x| > 1 and NaN return NaN (libm's domain rule; the RACER static-settling code tests the result with isfinite()). */
MPPI_HD static inline float asin(float x)
{
#if defined(MPPI_DET_MATH_LIBM) && !defined(__HIPCC__)
  return ::asinf(x);
#endif
  const float a = fabs(x);
  const bool upper = a > 0.5f;
  const float z = upper ? 0.5f * (1.0f - a) : a * a;
  const float w = upper ? sqrt(z) : a;
  float p = fma(4.2163199048e-2f, z, 2.4181311049e-2f);
  p = fma(p, z, 4.5470025998e-2f);
  p = fma(p, z, 7.4953002686e-2f);
  p = fma(p, z, 1.6666752422e-1f);
  float y = fma(p * z, w, w);
  y = upper ? 1.57079637050628662109375f - (y + y) : y;
  y = copysign(y, x);
  y = (a < 1.0e-4f) ? x : y;  // selects, not early returns: no divergent branch in a rollout step
  return (a <= 1.0f) ? y : u2f(0x7fc00000u);
}

/** x^y for x > 0 as exp(y*log(x)); used only for slowly varying discount factors (|y*log x| small). */
MPPI_HD static inline float pow_pos(float x, float y)
{
  return exp(y * log(x));
}

}  // namespace det
}  // namespace mppi

#endif  // MPPI_AMD_DET_MATH_H_
