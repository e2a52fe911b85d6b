// ilqg_trig.hpp — sine / cosine / tangent for the rollout's dependent chain (gfx950).
//
// The reference evaluates std::sin / std::cos / std::tan of headings and steering angles inside the RK4
// (single_player_car_5d.h:100-111 etc.).  On the device those calls sit on the one serial chain of the trial
// kernel: two libm latencies per time step.  The library versions carry an argument reduction valid up to 1e308
// (Payne-Hanek) behind a branch; headings and steering angles of a driving game are a few radians, so here the
// reduction is the two-constant Cody-Waite form (exact products through FMA) and the kernels are the classical
// minimax polynomials on [-pi/4, pi/4] (coefficients: the published fdlibm / msun kernels, k_sin.c, k_cos.c,
// k_sindf.c, k_cosdf.c).  Arguments beyond kTrigFastLimit take a reduction of their own (trig_reduce_large below: the
// Payne-Hanek scheme — the bits of 2/pi the argument's exponent selects, times the mantissa, in integer arithmetic) in
// front of the same kernels, so the functions are total: every finite argument, NaN for the rest.  Whether that branch
// exists for a lane is voted over a GROUP of lanes (`group`: a lane mask; by default the whole wavefront) — one branch for
// the group — but what a lane computes depends on its own argument only.  (Up to round 6 the branch called the device
// library's sincos / tan: inlined, its ~20 polynomial and reduction constants were hoisted out of the rollout's
// time-step loop into scalar registers that then spilled inside it; the reduction here keeps its table in memory and two
// constants in registers.)  Accuracy on the fast path: sine / cosine <= 1.5 ulp, tangent <= 3 ulp of the correctly rounded value
// (tests/host/trig_check.cpp checks it on the host over the whole range; scripts/ubench/trig_lat.hip
// prints the worst disagreement with libm) — the same order as the difference between the device libm and a host
// libm, and ten orders of magnitude inside the parity bar.
#pragma once

#include <hip/hip_runtime.h>

// No implicit contraction in this header: every fused multiply-add below is written out, so the functions compile to the
// same operations wherever they are inlined (the rollouts hand trajectories to one another bit for bit).
#pragma clang fp contract(off)

namespace ilqg {

// With FMAs every step of x - n hi - n lo - n lo2 rounds once — the first not at all: the difference is a multiple of
// 2^-52 below 2 — so three pieces of pi/2 carry the reduction as far as n fits: what is left of pi/2 beyond three doubles
// is n 2^-162.  |n| < 2^31 keeps the quadrant an int.  (Up to round 6: two pieces, |x| <= 1e5.  The rejected steps of a
// failing line search diverge through headings of 1e5 .. 1e9 by the thousand: inside the fast range they cost what every
// other step costs, and lanes that share a wavefront with them no branch.)
constexpr double kTrigFastLimit = 1.0e9;
constexpr float kTrigFastLimitF = 2.0e3f;

// r = x - n pi/2 with n = rint(x 2/pi); returns n's low bits
__host__ __device__ __forceinline__ int trig_reduce(double x, double* r) {
  const double n = __builtin_rint(x * 0x1.45f306dc9c883p-1);
  double t = __builtin_fma(-n, 0x1.921fb54442d18p+0, x);
  t = __builtin_fma(-n, 0x1.1a62633145c07p-54, t);
  t = __builtin_fma(-n, -0x1.f1976b7ed8fbcp-110, t);
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
  *c = w + __builtin_fma(z * z, pc, (1.0 - w) - hz);
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

// ---------------------------------------------------------------------------------------------------------------
// |x| > kTrigFastLimit: x = m 2^(E-52) with a 53-bit integer m.  x (2/pi) mod 4 only depends on the bits of 2/pi of
// weight below 2^(54-E): the 192 bits from bit E - 53 on (bit 1 = the first fractional bit of 2/pi; a window that
// starts in front of it reads zeros: the table's first word), as an integer W whose top bit weighs 2, times m, modulo
// 2^192: two quadrant bits, then 190 bits of fraction of which the bits of 2/pi left out disturb the last ~53.  The
// fraction, rounded to the nearest quadrant, leaves |f| <= 1/2 with at least ~75 good bits after the worst
// cancellation a double can produce (~2^-62); it becomes a double-double, times pi/2 = the reduced argument r_hi + r_lo.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned long long trig_mulhi64(unsigned long long a, unsigned long long b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (unsigned long long)(((unsigned __int128)a * b) >> 64);
#endif
}
// returns the quadrant (mod 4) of |x|; *r_hi + *r_lo = |x| - quadrant pi/2, |r| <= pi/4 (1 + 2^-50).  x finite, |x| >= 2^10.
__host__ __device__ inline int trig_reduce_large(double x, double* r_hi, double* r_lo) {
  // 64 zero bits, then the first 1216 bits of 2/pi
  const unsigned long long two_over_pi[20] = {
      0x0000000000000000ull, 0xa2f9836e4e441529ull, 0xfc2757d1f534ddc0ull, 0xdb6295993c439041ull,
      0xfe5163abdebbc561ull, 0xb7246e3a424dd2e0ull, 0x06492eea09d1921cull, 0xfe1deb1cb129a73eull,
      0xe88235f52ebb4484ull, 0xe99c7026b45f7e41ull, 0x3991d639835339f4ull, 0x9c845f8bbdf9283bull,
      0x1ff897ffde05980full, 0xef2f118b5a0a6d1full, 0x6d367ecf27cb09b7ull, 0x4f463f669e5fea2dull,
      0x7527bac7ebe5f17bull, 0x3d0739f78a5292eaull, 0x6bfb5fb11f8d5d08ull, 0x56033046fc7b6babull};
  const unsigned long long ux = (unsigned long long)__builtin_bit_cast(long long, x) & 0x7fffffffffffffffull;
  const int E = int(ux >> 52) - 1023;
  const unsigned long long m = (ux & 0x000fffffffffffffull) | 0x0010000000000000ull;
  const int pos = E + 10;  // bit E - 53 of 2/pi in the table's numbering (table bit 63 + j = bit j)
  const int w = pos >> 6, sh = pos & 63;
  const unsigned long long t0 = two_over_pi[w], t1 = two_over_pi[w + 1], t2 = two_over_pi[w + 2], t3 = two_over_pi[w + 3];
  const unsigned long long W2 = sh ? (t0 << sh) | (t1 >> (64 - sh)) : t0;
  const unsigned long long W1 = sh ? (t1 << sh) | (t2 >> (64 - sh)) : t1;
  const unsigned long long W0 = sh ? (t2 << sh) | (t3 >> (64 - sh)) : t2;
  // P = m W mod 2^192
  const unsigned long long p0 = m * W0;
  const unsigned long long c0 = trig_mulhi64(m, W0);
  const unsigned long long l1 = m * W1;
  const unsigned long long p1 = l1 + c0;
  const unsigned long long c1 = trig_mulhi64(m, W1) + (p1 < l1 ? 1ull : 0ull);
  const unsigned long long p2 = m * W2 + c1;
  int q = int(p2 >> 62);
  unsigned long long fh = (p2 << 2) | (p1 >> 62), fl = (p1 << 2) | (p0 >> 62);  // the fraction, 128 bits
  const bool neg = (fh >> 63) != 0;  // >= 1/2: the next quadrant, a negative remainder
  if (neg) {
    q += 1;
    fl = ~fl + 1ull;
    fh = ~fh + (fl == 0 ? 1ull : 0ull);
  }
  int shift = 0;
  if (fh == 0) {  // (not reached by a double: the closest one comes to a multiple of pi/2 leaves ~2^-62)
    fh = fl;
    fl = 0;
    shift = 64;
  }
  if (fh == 0) {
    *r_hi = 0.0;
    *r_lo = 0.0;
    return q & 3;
  }
  const int lz = __builtin_clzll(fh);
  const unsigned long long nh = lz ? (fh << lz) | (fl >> (64 - lz)) : fh, nl = fl << lz;
  shift += lz;
  // nh 2^64 + nl = a 2^75 + b 2^22 (+ 22 bits dropped), a and b 53-bit integers
  const double a = double((long long)(nh >> 11)), b = double((long long)(((nh & 0x7ffull) << 42) | (nl >> 22)));
  double f_hi = __builtin_ldexp(a, 75 - 128 - shift), f_lo = __builtin_ldexp(b, 22 - 128 - shift);
  if (neg) {
    f_hi = -f_hi;
    f_lo = -f_lo;
  }
  const double pio2_hi = 0x1.921fb54442d18p+0, pio2_lo = 0x1.1a62633145c07p-54;
  const double rh = f_hi * pio2_hi;
  *r_lo = __builtin_fma(f_hi, pio2_hi, -rh) + __builtin_fma(f_hi, pio2_lo, f_lo * pio2_hi);
  *r_hi = rh;
  return q & 3;
}

// Between the fast range and 2^47 the three-piece Cody-Waite form of trig_reduce still holds; only n no longer fits an int.
constexpr double kTrigMidLimit = 0x1p47;
__host__ __device__ __forceinline__ int trig_reduce_mid(double x, double* r) {
  const double n = __builtin_rint(x * 0x1.45f306dc9c883p-1);
  double t = __builtin_fma(-n, 0x1.921fb54442d18p+0, x);
  t = __builtin_fma(-n, 0x1.1a62633145c07p-54, t);
  t = __builtin_fma(-n, -0x1.f1976b7ed8fbcp-110, t);
  *r = t;
  return int(n - 4.0 * __builtin_rint(0.25 * n));  // n mod 4 in -2 .. 2 (n itself does not fit an int)
}

// sine and cosine of any double beyond the fast range
__host__ __device__ inline void large_sincos(double x, double* s, double* c) {
  if (__builtin_fabs(x) <= kTrigMidLimit) {
    double r, sr, cr;
    const int n = trig_reduce_mid(x, &r);
    trig_kernels(r, &sr, &cr);
    const bool swap = n & 1;
    const double sv = swap ? cr : sr, cv = swap ? sr : cr;
    *s = (n & 2) ? -sv : sv;
    *c = ((n + 1) & 2) ? -cv : cv;
    return;
  }
  if (!(__builtin_fabs(x) < __builtin_inf())) {  // +-inf, NaN
    *s = *c = x - x;
    return;
  }
  double rh, rl, sr, cr;
  const int n = trig_reduce_large(x, &rh, &rl);
  trig_kernels(rh, &sr, &cr);
  const double s1 = __builtin_fma(rl, cr, sr), c1 = __builtin_fma(-rl, sr, cr);  // first order in r_lo
  const bool swap = n & 1;
  const double sv = swap ? c1 : s1, cv = swap ? s1 : c1;
  const double sa = (n & 2) ? -sv : sv;
  *s = x < 0.0 ? -sa : sa;
  *c = ((n + 1) & 2) ? -cv : cv;
}

// (the branch is voted over the group on the device so that the common case is one uniform test; a lane inside the fast
// range computes the fast form whatever its neighbours hold)
__host__ __device__ __forceinline__ void fast_sincos(double x, double* s, double* c, unsigned long long group = ~0ull) {
  const bool mine = !(__builtin_fabs(x) <= kTrigFastLimit);
#if defined(__HIP_DEVICE_COMPILE__)
  const bool any = trig_group_any(mine, group);
#else
  const bool any = mine;
  (void)group;
#endif
  if (__builtin_expect(any, 0)) {
    if (mine) {
      large_sincos(x, s, c);
      return;
    }
  }
  fast_sincos_core<double>(x, s, c);
}
__host__ __device__ __forceinline__ void fast_sincos(float x, float* s, float* c, unsigned long long group = ~0ull) {
  const bool mine = !(__builtin_fabsf(x) <= kTrigFastLimitF);
#if defined(__HIP_DEVICE_COMPILE__)
  const bool any = trig_group_any(mine, group);
#else
  const bool any = mine;
  (void)group;
#endif
  if (__builtin_expect(any, 0)) {
    if (mine) {  // the reduction and the kernels in double, rounded once
      double sd, cd;
      large_sincos(double(x), &sd, &cd);
      *s = float(sd);
      *c = float(cd);
      return;
    }
  }
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
// beyond the fast range: sin / cos of the large reduction (a plain division: this path is cold)
__host__ __device__ inline double large_tan(double x) {
  if (__builtin_fabs(x) <= kTrigMidLimit) {  // as fast_tan_core, on the three-piece reduction
    double r, sr, cr;
    const int n = trig_reduce_mid(x, &r);
    trig_kernels(r, &sr, &cr);
    const bool odd = n & 1;
    return trig_div(odd ? -cr : sr, odd ? sr : cr);
  }
  double s, c;
  large_sincos(x, &s, &c);
  return s / c;
}
__host__ __device__ __forceinline__ double fast_tan(double x, unsigned long long group = ~0ull) {
  const bool mine = !(__builtin_fabs(x) <= kTrigFastLimit);
#if defined(__HIP_DEVICE_COMPILE__)
  const bool any = trig_group_any(mine, group);
#else
  const bool any = mine;
  (void)group;
#endif
  if (__builtin_expect(any, 0)) {
    if (mine) return large_tan(x);
  }
  return fast_tan_core<double>(x);
}
__host__ __device__ __forceinline__ float fast_tan(float x, unsigned long long group = ~0ull) {
  const bool mine = !(__builtin_fabsf(x) <= kTrigFastLimitF);
#if defined(__HIP_DEVICE_COMPILE__)
  const bool any = trig_group_any(mine, group);
#else
  const bool any = mine;
  (void)group;
#endif
  if (__builtin_expect(any, 0)) {
    if (mine) return float(large_tan(double(x)));
  }
  return fast_tan_core<float>(x);
}

}  // namespace ilqg

#pragma clang fp contract(fast)  // back to the translation unit's default
