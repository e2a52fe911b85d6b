// ilqg_trig.hpp — sine / cosine / tangent for the rollout's dependent chain (gfx950).
//
// The reference evaluates std::sin / std::cos / std::tan of headings and steering angles inside the RK4
// (single_player_car_5d.h:100-111 etc.).  On the device those calls sit on the one serial chain of the trial
// kernel: two libm latencies per time step.  The library versions carry an argument reduction valid up to 1e308
// (Payne-Hanek) behind a branch; headings and steering angles of a driving game are a few radians, so here the
// reduction is the two-constant Cody-Waite form (exact products through FMA) and the kernels are the classical
// minimax polynomials on [-pi/4, pi/4] (coefficients: the published fdlibm / msun kernels, k_sin.c, k_cos.c,
// k_sindf.c, k_cosdf.c).  Arguments beyond kTrigFastLimit fall back to the library, so the functions are total.  The test
// is taken over a GROUP of lanes (`group`: a lane mask; by default the whole wavefront) — one branch for the group, and what
// a lane computes depends on its group only: where two trajectories share a wavefront (rollout_pair) each passes its own
// half, so neither sees the other's arguments.  Accuracy on the fast path: sine / cosine <= 1.5 ulp, tangent <= 3 ulp of the correctly rounded value
// (tests/host/trig_check.cpp checks it on the host over the whole range; scripts/ubench/trig_lat.hip
// prints the worst disagreement with libm) — the same order as the difference between the device libm and a host
// libm, and ten orders of magnitude inside the parity bar.
#pragma once

#include <hip/hip_runtime.h>

namespace ilqg {

constexpr double kTrigFastLimit = 1.0e5;     // |n| < 2^17: n * pi/2_hi is exact to well below an ulp of the result
constexpr float kTrigFastLimitF = 2.0e3f;

// r = x - n pi/2 with n = rint(x 2/pi); returns n's low bits
__host__ __device__ __forceinline__ int trig_reduce(double x, double* r) {
  const double n = __builtin_rint(x * 0x1.45f306dc9c883p-1);
  double t = __builtin_fma(-n, 0x1.921fb54442d18p+0, x);
  t = __builtin_fma(-n, 0x1.1a62633145c07p-54, t);
  *r = t;
  return int(n);
}
__host__ __device__ __forceinline__ int trig_reduce(float x, float* r) {
  const float n = __builtin_rintf(x * 0.6366197466850281f);
  float t = __builtin_fmaf(-n, 1.5707963705062866f, x);
  t = __builtin_fmaf(-n, -4.371138828673793e-08f, t);
  t = __builtin_fmaf(-n, -1.7151245100058819e-15f, t);
  *r = t;
  return int(n);
}

// sin and cos on [-pi/4, pi/4]
__host__ __device__ __forceinline__ void trig_kernels(double r, double* s, double* c) {
  const double z = r * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  *s = __builtin_fma(r * z, ps, r);
  // 1 - z/2 + z^2 pc, with the rounding error of (1 - z/2) put back (k_cos.c)
  const double hz = 0.5 * z, w = 1.0 - hz;
  *c = w + (((1.0 - w) - hz) + z * z * pc);
}
__host__ __device__ __forceinline__ void trig_kernels(float r, float* s, float* c) {
  const float z = r * r;
  float ps = __builtin_fmaf(z, 2.7183114939898219064e-06f, -1.98393348360966317347e-04f);
  float pc = __builtin_fmaf(z, 2.43904487962774090654e-05f, -1.38867637746099294692e-03f);
  ps = __builtin_fmaf(z, ps, 8.3333293858894631756e-03f);
  pc = __builtin_fmaf(z, pc, 4.16666233237390631894e-02f);
  ps = __builtin_fmaf(z, ps, -1.66666666416265235595e-01f);
  pc = __builtin_fmaf(z, pc, -4.99999997251031003120e-01f);
  *s = __builtin_fmaf(r * z, ps, r);
  *c = __builtin_fmaf(z, pc, 1.0f);
}

template <typename T>
__host__ __device__ __forceinline__ void fast_sincos_core(T x, T* s, T* c) {
  T r, sr, cr;
  const int n = trig_reduce(x, &r);
  trig_kernels(r, &sr, &cr);
  const bool swap = n & 1;
  const T sv = swap ? cr : sr, cv = swap ? sr : cr;
  *s = (n & 2) ? -sv : sv;
  *c = ((n + 1) & 2) ? -cv : cv;
}

// does any lane of `group` (all lanes: ~0) hold an argument the fast path does not take?  NaN goes to the library too
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ bool trig_group_any(bool mine, unsigned long long group) {
  if (group == ~0ull) return __any(mine);
  return (__ballot(mine) & group) != 0;
}
#endif

// The library fall-backs as real calls (ILQG_TRIG_NOINLINE=1): the inlined library code brings its Payne-Hanek tables and
// polynomial constants into the caller, where the optimiser hoists them out of the time-step loop and the register
// allocator spills them — for a path that never runs in a driving game.
#ifndef ILQG_TRIG_NOINLINE
#define ILQG_TRIG_NOINLINE 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && ILQG_TRIG_NOINLINE
__device__ __attribute__((noinline, cold)) inline void slow_sincos(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __attribute__((noinline, cold)) inline double slow_tan(double x) { return tan(x); }
#define ILQG_SLOW_SINCOS slow_sincos
#define ILQG_SLOW_TAN slow_tan
#else
#define ILQG_SLOW_SINCOS sincos
#define ILQG_SLOW_TAN tan
#endif

__host__ __device__ __forceinline__ void fast_sincos(double x, double* s, double* c, unsigned long long group = ~0ull) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_expect(trig_group_any(!(__builtin_fabs(x) <= kTrigFastLimit), group), 0)) {
    ILQG_SLOW_SINCOS(x, s, c);
    return;
  }
#endif
  fast_sincos_core<double>(x, s, c);
}
__host__ __device__ __forceinline__ void fast_sincos(float x, float* s, float* c, unsigned long long group = ~0ull) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_expect(trig_group_any(!(__builtin_fabsf(x) <= kTrigFastLimitF), group), 0)) {
    sincosf(x, s, c);
    return;
  }
#endif
  fast_sincos_core<float>(x, s, c);
}

// tan = sin / cos of the reduced argument (cot with the sign flipped in odd quadrants); the quotient is the
// hardware reciprocal refined to within an ulp, then one residual correction.
__host__ __device__ __forceinline__ double trig_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double rc = __builtin_amdgcn_rcp(b);
#else  // host build of the unit test (tests/host/trig_check.cpp): any starting value within 2^-20 will do
  double rc = double(1.0f / float(b));
#endif
  rc = __builtin_fma(__builtin_fma(-b, rc, 1.0), rc, rc);
  rc = __builtin_fma(__builtin_fma(-b, rc, 1.0), rc, rc);
  const double q = a * rc;
  return __builtin_fma(__builtin_fma(-q, b, a), rc, q);
}
__host__ __device__ __forceinline__ float trig_div(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  float rc = __builtin_amdgcn_rcpf(b);
#else
  float rc = 1.0f / b;
#endif
  rc = __builtin_fmaf(__builtin_fmaf(-b, rc, 1.0f), rc, rc);
  const float q = a * rc;
  return __builtin_fmaf(__builtin_fmaf(-q, b, a), rc, q);
}
template <typename T>
__host__ __device__ __forceinline__ T fast_tan_core(T x) {
  T r, sr, cr;
  const int n = trig_reduce(x, &r);
  trig_kernels(r, &sr, &cr);
  const bool odd = n & 1;
  return trig_div(odd ? -cr : sr, odd ? sr : cr);
}
__host__ __device__ __forceinline__ double fast_tan(double x, unsigned long long group = ~0ull) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_expect(trig_group_any(!(__builtin_fabs(x) <= kTrigFastLimit), group), 0)) return ILQG_SLOW_TAN(x);
#endif
  return fast_tan_core<double>(x);
}
__host__ __device__ __forceinline__ float fast_tan(float x, unsigned long long group = ~0ull) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_expect(trig_group_any(!(__builtin_fabsf(x) <= kTrigFastLimitF), group), 0)) return tanf(x);
#endif
  return fast_tan_core<float>(x);
}

}  // namespace ilqg
