// ilqg_models.hpp — device-side dynamics, geometry and cost models (gfx950).
//
// Restates, for one lane, the closed-form models the reference evaluates through
// virtual calls; every function cites the reference lines it computes the same
// quantity as.  Written for the "one lane integrates one subsystem / walks one
// player's cost list" mapping of the rollout and quadraticisation stages.
#pragma once

#include "ilqg_common.hpp"
#include "ilqg_trig.hpp"

namespace ilqg {

template <typename T> __device__ __forceinline__ T t_sin(T x);
template <> __device__ __forceinline__ float t_sin<float>(float x) { return sinf(x); }
template <> __device__ __forceinline__ double t_sin<double>(double x) { return sin(x); }
template <typename T> __device__ __forceinline__ T t_cos(T x);
template <> __device__ __forceinline__ float t_cos<float>(float x) { return cosf(x); }
template <> __device__ __forceinline__ double t_cos<double>(double x) { return cos(x); }
template <typename T> __device__ __forceinline__ T t_tan(T x);
template <> __device__ __forceinline__ float t_tan<float>(float x) { return tanf(x); }
template <> __device__ __forceinline__ double t_tan<double>(double x) { return tan(x); }
template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T t_hypot(T x, T y);
template <> __device__ __forceinline__ float t_hypot<float>(float x, float y) { return hypotf(x, y); }
template <> __device__ __forceinline__ double t_hypot<double>(double x, double y) { return hypot(x, y); }
template <typename T> __device__ __forceinline__ T t_abs(T x) { return x < T(0) ? -x : x; }

// ---------------------------------------------------------------------------
// Dynamics — x is one subsystem's state (<= 6), u its 2 inputs.
// single_player_unicycle_4d.h:90-100, single_player_car_5d.h:100-111,
// single_player_car_6d.h:102-114.
// ---------------------------------------------------------------------------
// TwoPlayerUnicycle4D (two_player_unicycle_4d.h:105-118) is the unicycle with the second player's (d0, d1)
// added to the position rates; its second row has no state of its own.
__device__ __forceinline__ bool is_unicycle(int kind) {
  return kind == ILQG_DYN_UNICYCLE_4D || kind == ILQG_DYN_UNICYCLE_4D_DISTURBED;
}

template <typename T>
__device__ __forceinline__ void sub_eval(int kind, T L, const T* x, T u0, T u1, T* xd, T d0 = T(0), T d1 = T(0)) {
  if (kind == ILQG_DYN_POINT_MASS_2D) {  // single_player_point_mass_2d.h:90-99
    xd[0] = x[2];
    xd[1] = x[3];
    xd[2] = u0;
    xd[3] = u1;
    xd[4] = xd[5] = T(0);
  } else if (kind == ILQG_DYN_AIR_3D_EVADER) {  // air_3d.h:112-125: L = evader speed, d0 = pursuer speed, u1 = its turn rate
    xd[0] = -L + d0 * t_cos(x[2]) + u0 * x[1];
    xd[1] = d0 * t_sin(x[2]) - u0 * x[0];
    xd[2] = u1 - u0;
    xd[3] = xd[4] = xd[5] = T(0);
  } else if (kind == ILQG_DYN_DUBINS_CAR) {  // single_player_dubins_car.h:94-103: constant speed L, u = omega
    xd[0] = L * t_cos(x[2]);
    xd[1] = L * t_sin(x[2]);
    xd[2] = u0;
    xd[3] = xd[4] = xd[5] = T(0);
  } else if (is_unicycle(kind) || kind == ILQG_DYN_PLANAR_DISTURBANCE) {
    xd[0] = x[3] * t_cos(x[2]) + d0;
    xd[1] = x[3] * t_sin(x[2]) + d1;
    xd[2] = u0;
    xd[3] = u1;
    xd[4] = T(0);
    xd[5] = T(0);
  } else {
    xd[0] = x[4] * t_cos(x[2]);
    xd[1] = x[4] * t_sin(x[2]);
    xd[2] = (x[4] / L) * t_tan(x[3]);
    xd[3] = u0;
    if (kind == ILQG_DYN_CAR_5D) {
      xd[4] = u1;
      xd[5] = T(0);
    } else {
      xd[4] = x[5];
      xd[5] = u1;
    }
  }
}

// MultiPlayerDynamicalSystem::Integrate (src/multi_player_dynamical_system.cpp:52-77):
// RK4, 2 sub-steps of dt/2.  Subsystems of a ConcatenatedDynamicalSystem are
// decoupled once u is fixed, so one lane integrates one block in registers.
template <typename T>
__device__ __forceinline__ void sub_integrate(int kind, T L, double interval, T* x, T u0, T u1, T d0 = T(0),
                                              T d1 = T(0)) {
  const T h = T(interval / 2.0);
#pragma unroll 1
  for (int s = 0; s < 2; s++) {
    T k1[6], k2[6], k3[6], k4[6], xt[6];
    sub_eval(kind, L, x, u0, u1, k1, d0, d1);
#pragma unroll
    for (int i = 0; i < 6; i++) { k1[i] = h * k1[i]; xt[i] = x[i] + T(0.5) * k1[i]; }
    sub_eval(kind, L, xt, u0, u1, k2, d0, d1);
#pragma unroll
    for (int i = 0; i < 6; i++) { k2[i] = h * k2[i]; xt[i] = x[i] + T(0.5) * k2[i]; }
    sub_eval(kind, L, xt, u0, u1, k3, d0, d1);
#pragma unroll
    for (int i = 0; i < 6; i++) { k3[i] = h * k3[i]; xt[i] = x[i] + k3[i]; }
    sub_eval(kind, L, xt, u0, u1, k4, d0, d1);
#pragma unroll
    for (int i = 0; i < 6; i++) {
      k4[i] = h * k4[i];
      x[i] += (k1[i] + T(2.0) * (k2[i] + k3[i]) + k4[i]) / T(6.0);
    }
  }
}

// The models outside the stage-per-lane integrator's closed forms — SinglePlayerUnicycle5D (unicycle_5d.h:92-103),
// SinglePlayerCar7D (car_7d.h:104-120), SinglePlayerDelayedDubinsCar (delayed_dubins_car.h:103-113) — and any other
// kind through sub_eval: blocks of up to 8 states, the plain RK4 of MultiPlayerDynamicalSystem::Integrate.
constexpr int kSubStatesMax = 8;
__host__ __device__ inline bool is_plain_rk4_kind(int kind) {
  return kind == ILQG_DYN_UNICYCLE_5D || kind == ILQG_DYN_CAR_7D || kind == ILQG_DYN_DELAYED_DUBINS_CAR;
}
template <typename T>
__device__ __forceinline__ void sub_eval8(int kind, T L, const T* x, T u0, T u1, T* xd, T d0 = T(0), T d1 = T(0)) {
  if (kind == ILQG_DYN_UNICYCLE_5D) {
    xd[0] = x[3] * t_cos(x[2]);
    xd[1] = x[3] * t_sin(x[2]);
    xd[2] = u0;
    xd[3] = u1;
    xd[4] = x[3];
    xd[5] = xd[6] = xd[7] = T(0);
  } else if (kind == ILQG_DYN_CAR_7D) {
    xd[0] = x[4] * t_cos(x[2]);
    xd[1] = x[4] * t_sin(x[2]);
    xd[2] = (x[4] / L) * t_tan(x[3]);
    xd[3] = u0;
    xd[4] = u1;
    const T sec_phi = T(1.0 / double(t_cos(x[3])));  // a float made from a double quotient there (car_7d.h:115)
    xd[5] = u0 * sec_phi * sec_phi / L;
    xd[6] = x[4];
    xd[7] = T(0);
  } else if (kind == ILQG_DYN_DELAYED_DUBINS_CAR) {
    xd[0] = L * t_cos(x[2]);
    xd[1] = L * t_sin(x[2]);
    xd[2] = x[3];
    xd[3] = u0;
    xd[4] = xd[5] = xd[6] = xd[7] = T(0);
  } else {
    sub_eval<T>(kind, L, x, u0, u1, xd, d0, d1);
    xd[6] = xd[7] = T(0);
  }
}
template <typename T>
__device__ __forceinline__ void sub_integrate8(int kind, T L, double interval, T* x, T u0, T u1, T d0 = T(0),
                                               T d1 = T(0)) {
  constexpr int XS = kSubStatesMax;
  const T h = T(interval / 2.0);
#pragma unroll 1
  for (int s = 0; s < 2; s++) {
    T k1[XS], k2[XS], k3[XS], k4[XS], xt[XS];
    sub_eval8(kind, L, x, u0, u1, k1, d0, d1);
#pragma unroll
    for (int i = 0; i < XS; i++) { k1[i] = h * k1[i]; xt[i] = x[i] + T(0.5) * k1[i]; }
    sub_eval8(kind, L, xt, u0, u1, k2, d0, d1);
#pragma unroll
    for (int i = 0; i < XS; i++) { k2[i] = h * k2[i]; xt[i] = x[i] + T(0.5) * k2[i]; }
    sub_eval8(kind, L, xt, u0, u1, k3, d0, d1);
#pragma unroll
    for (int i = 0; i < XS; i++) { k3[i] = h * k3[i]; xt[i] = x[i] + k3[i]; }
    sub_eval8(kind, L, xt, u0, u1, k4, d0, d1);
#pragma unroll
    for (int i = 0; i < XS; i++) {
      k4[i] = h * k4[i];
      x[i] += (k1[i] + T(2.0) * (k2[i] + k3[i]) + k4[i]) / T(6.0);
    }
  }
}

template <typename T> __device__ __forceinline__ void t_sincos(T x, T* s, T* c);
template <> __device__ __forceinline__ void t_sincos<float>(float x, float* s, float* c) { sincosf(x, s, c); }
template <> __device__ __forceinline__ void t_sincos<double>(double x, double* s, double* c) { sincos(x, s, c); }

// x / c for a divisor c that is fixed across many divisions (6, the axle length): q = x * (1/c) corrected by
// one residual step — q' = q + (x - q c) (1/c), both through FMA — which is the correctly rounded quotient
// (Markstein) at a third of the instructions of the IEEE division sequence; the RK4 has 22 of them per step.
template <typename T>
__device__ __forceinline__ T div_by(T x, T c, T rc) {
  const T q = x * rc;
  const T r = __builtin_fma(-q, c, x);
  return __builtin_fma(r, rc, q);
}
__device__ __forceinline__ float div_by(float x, float c, float rc) {
  const float q = x * rc;
  const float r = __builtin_fmaf(-q, c, x);
  return __builtin_fmaf(r, rc, q);
}

// (No implicit contraction from here to the end of the two stage-form integrators: their fused multiply-adds are written
// out, so the lane-per-stage and the stages-in-a-lane forms round alike wherever they are inlined.)
#pragma clang fp contract(off)
// The RK4 (2 sub-steps) of one subsystem with ONE STAGE PER LANE and the stage values in closed form.
// Lanes base..base+7 form the group; lane q = 4 s + j evaluates stage j of sub-step s.  What makes this possible:
// the steering angle (or the unicycle's heading), the speed and the acceleration are driven by inputs that are
// constant over the step, so RK4's stage values of these components are polynomials in h with known coefficients —
//   phi_q = phi_0 + c_q h omega,  c = (0, 1/2, 1/2, 1, 1, 3/2, 3/2, 2)
//   Car6D: v at stage j of a sub-step from (v_s, a_s):  v_s + alpha_j h a_s + beta_j h^2 jerk,
//          alpha = (0, 1/2, 1/2, 1), beta = (0, 0, 1/4, 1/2);  v_{s+1} = v_s + h a_s + h^2 jerk / 2
// — every lane forms its own stage's values with two or three FMAs instead of walking the stages in sequence.  The
// heading's stage derivative k_q = h (v_q / L) tan(phi_q) is then one tan per lane; the eight k go through LDS, every
// lane forms the heading at its stage from them; one sincos per lane gives the position rates, which go through LDS
// once more for the two RK4 combinations.  Two libm latencies and two LDS exchanges per step; the expressions are
// RK4's up to the order of the roundings (a few ulp per step).  All 8 lanes return the new state in x[].
// `gth` is LDS scratch of 64 + 128 elements (this wave's); any_car: some group of the wave holds a car model;
// `group`: the lanes of this trajectory (ilqg_trig.hpp: the library fall-back is decided over them).
template <typename T, bool DIST = false, bool DUB = false>
__device__ __forceinline__ void sub_integrate_stages(int kind, T L, double interval, T* x, T u0, T u1, int q, int lane,
                                                     T* gth, bool any_car, T d0 = T(0), T d1 = T(0),
                                                     unsigned long long group = ~0ull) {
  const T h = T(interval / 2.0);
  const T six = T(6.0), rsix = T(1.0) / T(6.0), rL = T(1.0) / L;
  const bool dubins = DUB && kind == ILQG_DYN_DUBINS_CAR;
  const bool car = DIST ? false : (kind == ILQG_DYN_CAR_5D || kind == ILQG_DYN_CAR_6D);
  const bool car6 = car && kind == ILQG_DYN_CAR_6D;
  const int j = q & 3;
  const bool s1 = q >= 4;
  const T aj = j == 0 ? T(0) : (j == 3 ? T(1) : T(0.5));
  const T cq = aj + (s1 ? T(1) : T(0));
  // ---- the input-driven components at this lane's stage and at the end of the step ----
  const T ang0 = car ? x[3] : x[2];
  const T hk = h * u0;
  const T ang_q = t_fma(cq, hk, ang0);
  const T ang_end = t_fma(T(2), hk, ang0);
  const T v0 = car ? x[4] : x[3];
  const T hj = h * u1;  // Car6D: h jerk; otherwise h a
  T v_q, v_end, a_end = T(0);
  {
    const T a0 = x[5];
    const T a_1 = a0 + hj;
    const T v_1 = t_fma(T(0.5) * h, hj, t_fma(h, a0, v0));
    const T vb = s1 ? v_1 : v0, ab = s1 ? a_1 : a0;
    const T bj = j == 2 ? T(0.25) : (j == 3 ? T(0.5) : T(0));
    const T v6_q = t_fma(bj * h, hj, t_fma(aj * h, ab, vb));
    const T v6_end = t_fma(T(0.5) * h, hj, t_fma(h, a_1, v_1));
    const T v5_q = t_fma(cq, hj, v0), v5_end = t_fma(T(2), hj, v0);
    v_q = car6 ? v6_q : v5_q;
    v_end = car6 ? v6_end : v5_end;
    a_end = t_fma(T(2), hj, a0);
    if (dubins) v_q = L;
  }
  // ---- heading at this lane's stage ----
  T th_q = ang_q, th_end = ang_end;
  if (any_car) {  // wave-uniform
    const T kth = car ? h * (div_by(v_q, L, rL) * fast_tan(ang_q, group)) : T(0);
    gth[lane] = kth;
    lds_sync(true);
    const T* g = gth + (lane & ~7);
    const T k0 = g[0], k1 = g[1], k2 = g[2], k3 = g[3], k4 = g[4], k5 = g[5], k6 = g[6], k7 = g[7];
    // the previous stage's derivative; stage 0 has none (aj = 0) and takes a zero — the neighbouring lane belongs to
    // another subsystem or, in rollout_pair, to the other trajectory, whose non-finite value 0 * would turn into a NaN here
    const T kprev = (lane & 7) ? gth[lane - 1] : T(0);
    const T th1 = x[2] + div_by(t_fma(T(2), k1 + k2, k0) + k3, six, rsix);
    const T th2 = th1 + div_by(t_fma(T(2), k5 + k6, k4) + k7, six, rsix);
    const T thb = s1 ? th1 : x[2];
    th_q = car ? t_fma(aj, kprev, thb) : ang_q;
    th_end = car ? th2 : ang_end;
  }
  // ---- position rates of this lane's stage, then the two RK4 combinations ----
  T sn, cs;
  fast_sincos(th_q, &sn, &cs, group);
  const T kx = DIST ? h * t_fma(v_q, cs, d0) : h * (v_q * cs);
  const T ky = DIST ? h * t_fma(v_q, sn, d1) : h * (v_q * sn);
  T* gxy = gth + 64;
  gxy[2 * lane] = kx;
  gxy[2 * lane + 1] = ky;
  lds_sync(true);
  const T* gq = gxy + 2 * (lane & ~7);
  T px = x[0], py = x[1];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const T a1 = gq[8 * s + 0], b1 = gq[8 * s + 1], a2 = gq[8 * s + 2], b2 = gq[8 * s + 3];
    const T a3 = gq[8 * s + 4], b3 = gq[8 * s + 5], a4 = gq[8 * s + 6], b4 = gq[8 * s + 7];
    px += div_by(t_fma(T(2), a2 + a3, a1) + a4, six, rsix);
    py += div_by(t_fma(T(2), b2 + b3, b1) + b4, six, rsix);
  }
  x[0] = px;
  x[1] = py;
  x[2] = th_end;
  if (car) {
    x[3] = ang_end;
    x[4] = v_end;
    if (car6) x[5] = a_end;
  } else if (!dubins) {
    x[3] = v_end;
  }
}

// What a lane hands another through LDS in sub_integrate_stages is a rounded value the optimiser cannot look into; here
// the eight stages sit in one lane, and a product feeding a sum of another stage would be fused into it (the
// translation unit contracts) — one rounding less than the exchange form.  This keeps a stage's results opaque.
template <typename T>
__device__ __forceinline__ T stage_value(T v) {
  asm volatile("" : "+v"(v));
  return v;
}

// sub_integrate_stages with the EIGHT STAGES IN ONE LANE: the same expressions stage by stage, the same roundings — a
// trajectory comes out bit for bit as from the stage-parallel form (tests/test_gpu_parity.py: the speculative line search
// hands a probed trajectory over in place of the regular pass's) — for rollouts that are many at once: a lane per
// (trajectory, subsystem) issues ~1/8 of the instructions per trajectory (rollout_lanes, ilqg_stages.hpp).  Stages with
// the same input-driven angle (c_q: 1 and 2, 3 and 4, 5 and 6) share their tangent.
template <typename T, bool DUB = false>
__device__ __forceinline__ void sub_integrate_stages_seq(int kind, T L, double interval, T* x, T u0, T u1) {
  const T h = T(interval / 2.0);
  const T six = T(6.0), rsix = T(1.0) / T(6.0), rL = T(1.0) / L;
  const bool dubins = DUB && kind == ILQG_DYN_DUBINS_CAR;
  const bool car = kind == ILQG_DYN_CAR_5D || kind == ILQG_DYN_CAR_6D;
  const bool car6 = car && kind == ILQG_DYN_CAR_6D;
  const T ang0 = car ? x[3] : x[2];
  const T hk = h * u0;
  const T ang_end = t_fma(T(2), hk, ang0);
  const T v0 = car ? x[4] : x[3];
  const T hj = h * u1;
  const T a0 = x[5];
  const T a_1 = a0 + hj;
  const T v_1 = t_fma(T(0.5) * h, hj, t_fma(h, a0, v0));
  const T v6_end = t_fma(T(0.5) * h, hj, t_fma(h, a_1, v_1));
  const T v5_end = t_fma(T(2), hj, v0);
  const T v_end = car6 ? v6_end : v5_end;
  const T a_end = t_fma(T(2), hj, a0);
  T ang[8], vq[8], kth[8];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int j = q & 3;
    const bool s1 = q >= 4;
    const T aj = j == 0 ? T(0) : (j == 3 ? T(1) : T(0.5));
    const T cq = aj + (s1 ? T(1) : T(0));
    ang[q] = t_fma(cq, hk, ang0);
    const T vb = s1 ? v_1 : v0, ab = s1 ? a_1 : a0;
    const T bj = j == 2 ? T(0.25) : (j == 3 ? T(0.5) : T(0));
    const T v6_q = t_fma(bj * h, hj, t_fma(aj * h, ab, vb));
    const T v5_q = t_fma(cq, hj, v0);
    vq[q] = car6 ? v6_q : v5_q;
    if (dubins) vq[q] = L;
    kth[q] = T(0);
  }
  T th1 = x[2], th2 = x[2];
  if (car) {
    T tn[8];
    tn[0] = fast_tan(ang[0]);
    tn[1] = fast_tan(ang[1]);
    tn[2] = tn[1];
    tn[3] = fast_tan(ang[3]);
    tn[4] = tn[3];
    tn[5] = fast_tan(ang[5]);
    tn[6] = tn[5];
    tn[7] = fast_tan(ang[7]);
#pragma unroll
    for (int q = 0; q < 8; q++) kth[q] = stage_value(h * (div_by(vq[q], L, rL) * tn[q]));
    th1 = x[2] + div_by(t_fma(T(2), kth[1] + kth[2], kth[0]) + kth[3], six, rsix);
    th2 = th1 + div_by(t_fma(T(2), kth[5] + kth[6], kth[4]) + kth[7], six, rsix);
  }
  const T th_end = car ? th2 : ang_end;
  T kx[8], ky[8];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int j = q & 3;
    const bool s1 = q >= 4;
    const T aj = j == 0 ? T(0) : (j == 3 ? T(1) : T(0.5));
    const T thb = s1 ? th1 : x[2];
    const T kprev = q > 0 ? kth[q - 1] : T(0);
    const T th_q = car ? t_fma(aj, kprev, thb) : ang[q];
    T sn, cs;
    fast_sincos(th_q, &sn, &cs);
    kx[q] = stage_value(h * (vq[q] * cs));
    ky[q] = stage_value(h * (vq[q] * sn));
  }
  T px = x[0], py = x[1];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    px += div_by(t_fma(T(2), kx[4 * s + 1] + kx[4 * s + 2], kx[4 * s]) + kx[4 * s + 3], six, rsix);
    py += div_by(t_fma(T(2), ky[4 * s + 1] + ky[4 * s + 2], ky[4 * s]) + ky[4 * s + 3], six, rsix);
  }
  x[0] = px;
  x[1] = py;
  x[2] = th_end;
  if (car) {
    x[3] = ang_end;
    x[4] = v_end;
    if (car6) x[5] = a_end;
  } else if (!dubins) {
    x[3] = v_end;
  }
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------
// Geometry
// ---------------------------------------------------------------------------
// ---- geometry, costs, constraints ----
// No floating-point contraction from here to the end of the file (hipcc's default fuses a * b + c into an FMA wherever the
// optimiser sees the pair in one block — "fast-honor-pragmas").  Which pairs it sees depends on the shape of the code
// around them: the row stage evaluates these functions from an interpreter loop, from straight-line code with every
// kind and index folded (ilqg_rows.hpp, ProgStatic), with run-time dimensions, gradient-only ... and each form fused
// different pairs — results that differ in the last bit, line searches that part ways.  With contraction off every form
// rounds exactly as the source is written (the reference's own arithmetic: its CI builds have no FMA either), so all of
// them agree bit for bit by construction.  The rollout's integrators above keep the default.
#pragma clang fp contract(off)
template <typename T>
struct Seg {
  T p1x, p1y, p2x, p2y, len, ux, uy;
};

// LineSegment2 ctor, include/ilqgames/geometry/line_segment2.h:55-62
template <typename T>
__device__ __forceinline__ Seg<T> make_seg(T ax, T ay, T bx, T by) {
  Seg<T> s;
  s.p1x = ax; s.p1y = ay; s.p2x = bx; s.p2y = by;
  const T dx = ax - bx, dy = ay - by;
  s.len = t_sqrt(dx * dx + dy * dy);
  s.ux = (bx - ax) / s.len;
  s.uy = (by - ay) / s.len;
  return s;
}

// LineSegment2::Side, src/line_segment2.cpp:48-54
template <typename T>
__device__ __forceinline__ bool seg_side(const Seg<T>& s, T qx, T qy) {
  const T rx = qx - s.p1x, ry = qy - s.p1y;
  return (rx * s.uy - s.ux * ry) > T(0);
}

// LineSegment2::ClosestPoint, src/line_segment2.cpp:56-100
template <typename T>
__device__ __forceinline__ void seg_closest(const Seg<T>& s, T qx, T qy, T* cx, T* cy, bool* endp, T* ssd) {
  const T rx = qx - s.p1x, ry = qy - s.p1y;
  const T dot = rx * s.ux + ry * s.uy;
  const T cross = rx * s.uy - s.ux * ry;
  const T cs = sgn(cross);
  if (dot < T(0)) {
    *endp = true;
    *ssd = cs * (rx * rx + ry * ry);
    *cx = s.p1x;
    *cy = s.p1y;
  } else if (dot > s.len) {
    *endp = true;
    const T ex = qx - s.p2x, ey = qy - s.p2y;
    *ssd = cs * (ex * ex + ey * ey);
    *cx = s.p2x;
    *cy = s.p2y;
  } else {
    *endp = false;
    *ssd = cs * cross * cross;
    *cx = s.p1x + dot * s.ux;
    *cy = s.p1y + dot * s.uy;
  }
}

template <typename T>
struct Closest {
  T cx, cy, ssd;
  bool is_vertex, is_endpoint;
  Seg<T> seg;
};

template <typename T>
__device__ __forceinline__ Seg<T> load_seg(const T* s) {
  Seg<T> o;
  o.p1x = s[0]; o.p1y = s[1]; o.p2x = s[2]; o.p2y = s[3]; o.len = s[4]; o.ux = s[5]; o.uy = s[6];
  return o;
}

// Device-side tables the cost stages read (LDS-resident in the kernels).
template <typename T>
struct QuadTables {
  const DevTerm* terms;  // [num_terms]
  const T* segs;         // [total_segs][kSegStride]
  const int* poly_off;   // [num_polylines + 1]
  const int* order;      // [N][cost_order_stride]
  // Lane-indexed fields of DevProblem (per-player and per-block tables).  Reading them from the
  // kernel-argument copy with a per-lane index is a global load (and `rgoff[pii[i]]` a chain of two);
  // from LDS it is a 64-cycle read.  Layout: LC_* offsets below, floats stored as their bit patterns.
  const int* lc;
  // The per-step nominals of the time-dependent costs (DevProblem::time_nominal of this precision), [table][T][2].
  const double* tnom;
  int tnom_T;
  const T* dense = nullptr;  // coefficient blocks of the affine constraints (DevProblem::dense_f / dense_d)
};
__host__ __device__ inline bool term_is_affine(int kind) {
  return kind == ILQG_CONSTRAINT_AFFINE_SCALAR || kind == ILQG_CONSTRAINT_AFFINE_VECTOR;
}
// Constraint::Mu(lambda, g) (constraint.h:112-117): the inactive-inequality gate; an equality constraint has none
template <typename T>
__device__ __forceinline__ T constraint_mu(T lambda, T g, T mu, bool is_equality) {
  const T al = lambda < T(0) ? -lambda : lambda;
  return (!is_equality && g <= T(1e-4f) && al <= T(1e-4f)) ? T(0) : mu;
}
// g of the two affine constraints from their coefficient block: a^T v - b (affine_scalar_constraint.h:63-66),
// |A v - b| (affine_vector_constraint.h:70-73)
template <typename T, typename V>
__device__ __forceinline__ T affine_evaluate(int kind, const T* blk, const V& v, int dim) {
  if (kind == ILQG_CONSTRAINT_AFFINE_SCALAR) {
    T s = T(0);
    for (int i = 0; i < dim; i++) s += blk[i] * v[i];
    return s - blk[dim];
  }
  const T* b = blk + dim * dim;
  T sq = T(0);
  for (int i = 0; i < dim; i++) {
    T dlt = T(0);
    for (int j = 0; j < dim; j++) dlt += blk[i + dim * j] * v[j];
    dlt -= b[i];
    sq += dlt * dlt;
  }
  return t_sqrt(sq);
}
// WeightedConvexProximityCost touches more entries than one pattern holds: the row program evaluates it as four ops —
// the cost kind itself (the position block: the relative-position pattern, and the value) and three internal kinds
// that only exist inside row programs: the speed block (PAIR2 over (v1, v2)) and the position x speed blocks of the
// two axes (CROSS4; only the active axis contributes).  Every op re-derives the few scalars it needs from the six
// state entries; where the indices that are not part of its own pattern travel is wcp_indices().
enum { ILQG_INTERNAL_WCP_SPEED = 101, ILQG_INTERNAL_WCP_CROSS_X = 102, ILQG_INTERNAL_WCP_CROSS_Y = 103 };
__host__ __device__ inline bool term_is_wcp(int kind) {
  return kind == ILQG_COST_WEIGHTED_CONVEX_PROXIMITY || (kind >= ILQG_INTERNAL_WCP_SPEED && kind <= ILQG_INTERNAL_WCP_CROSS_Y);
}
// (x1, y1, x2, y2, v1, v2) of the op's term; the op's own pattern indices are idx[], the others ride in 16-bit halves
// of `polyline` (and of idx[2], idx[3] for the speed block, whose pattern uses two indices).
struct WcpIdx { int x1, y1, x2, y2, v1, v2; };  // scalars, not an array: the kernels keep them in (scalar) registers
__host__ __device__ inline WcpIdx wcp_indices(const DevTerm& c) {
  const int lo = c.polyline & 0xffff, hi = (c.polyline >> 16) & 0xffff;
  WcpIdx ix;
  if (c.kind == ILQG_COST_WEIGHTED_CONVEX_PROXIMITY) {
    ix.x1 = c.idx[0]; ix.y1 = c.idx[1]; ix.x2 = c.idx[2]; ix.y2 = c.idx[3]; ix.v1 = lo; ix.v2 = hi;
  } else if (c.kind == ILQG_INTERNAL_WCP_SPEED) {
    ix.v1 = c.idx[0]; ix.v2 = c.idx[1];
    ix.x1 = c.idx[2] & 0xffff; ix.y1 = (c.idx[2] >> 16) & 0xffff; ix.x2 = c.idx[3] & 0xffff; ix.y2 = (c.idx[3] >> 16) & 0xffff;
  } else if (c.kind == ILQG_INTERNAL_WCP_CROSS_X) {
    ix.x1 = c.idx[0]; ix.x2 = c.idx[1]; ix.v1 = c.idx[2]; ix.v2 = c.idx[3]; ix.y1 = lo; ix.y2 = hi;
  } else {
    ix.y1 = c.idx[0]; ix.y2 = c.idx[1]; ix.v1 = c.idx[2]; ix.v2 = c.idx[3]; ix.x1 = lo; ix.x2 = hi;
  }
  return ix;
}
// The sub-op `which` (0 = the cost kind, else an internal kind) of the term whose indices are ix.
__host__ inline DevTerm wcp_sub_term(const DevTerm& c, const WcpIdx& ix, int which) {
  DevTerm o = c;
  o.kind = which == 0 ? int(ILQG_COST_WEIGHTED_CONVEX_PROXIMITY) : which;
  if (which == 0) {
    o.idx[0] = ix.x1; o.idx[1] = ix.y1; o.idx[2] = ix.x2; o.idx[3] = ix.y2; o.polyline = ix.v1 | (ix.v2 << 16);
  } else if (which == ILQG_INTERNAL_WCP_SPEED) {
    o.idx[0] = ix.v1; o.idx[1] = ix.v2; o.idx[2] = ix.x1 | (ix.y1 << 16); o.idx[3] = ix.x2 | (ix.y2 << 16); o.polyline = 0;
  } else if (which == ILQG_INTERNAL_WCP_CROSS_X) {
    o.idx[0] = ix.x1; o.idx[1] = ix.x2; o.idx[2] = ix.v1; o.idx[3] = ix.v2; o.polyline = ix.y1 | (ix.y2 << 16);
  } else {
    o.idx[0] = ix.y1; o.idx[1] = ix.y2; o.idx[2] = ix.v1; o.idx[3] = ix.v2; o.polyline = ix.x1 | (ix.x2 << 16);
  }
  return o;
}
__host__ __device__ inline bool term_is_time_dependent(int kind) {
  return kind == ILQG_COST_NOMINAL_PATH_LENGTH || kind == ILQG_COST_ROUTE_PROGRESS;
}
enum {
  LC_KIND = 0, LC_XOFF = LC_KIND + kMaxPlayers, LC_UOFF = LC_XOFF + kMaxPlayers + 1,
  LC_UDIM = LC_UOFF + kMaxPlayers + 1, LC_PARAM = LC_UDIM + kMaxPlayers, LC_SREG = LC_PARAM + kMaxPlayers,
  LC_CREG = LC_SREG + kMaxPlayers, LC_STRUCT = LC_CREG + kMaxPlayers, LC_PI = LC_STRUCT + kMaxPlayers,
  LC_PJ = LC_PI + kMaxPairs, LC_ROFF = LC_PJ + kMaxPairs, LC_RGOFF = LC_ROFF + kMaxPairs,
  LC_FROMCOST = LC_RGOFF + kMaxPairs, LC_PII = LC_FROMCOST + kMaxPairs, LC_COUNT = (LC_PII + kMaxPlayers + 3) & ~3
};

// Polyline2::ClosestPoint, src/polyline2.cpp:105-174 — linear scan over the 1..15 segments of a
// lane; the "shortcut" sign rule at interior vertices and the 1e-4 endpoint rule are reproduced.
// Segments (and the shortcut segments of interior vertices) are precomputed LineSegment2 objects.
// UNIFORM: `poly` is the same on every lane (the lane-per-time-step stage): the scan's bounds are pinned to scalar
// registers so that its loop is a scalar branch.
template <typename T, bool UNIFORM = false>
__device__ __forceinline__ Closest<T> polyline_closest(const QuadTables<T>& tb, int poly, T qx, T qy) {
  int first = tb.poly_off[poly] - poly;  // segments before this polyline
  int nseg = tb.poly_off[poly + 1] - tb.poly_off[poly] - 1;
  if constexpr (UNIFORM) {
    first = __builtin_amdgcn_readfirstlane(first);
    nseg = __builtin_amdgcn_readfirstlane(nseg);
  }
  const T* base = tb.segs + size_t(first) * kSegStride;
  Closest<T> out;
  T best = dinf<T>();
  out.cx = T(0);
  out.cy = T(0);
  out.is_vertex = false;
  int best_idx = 0;
  for (int c = 0; c < nseg; c++) {
    const Seg<T> s = load_seg<T>(base + c * kSegStride);
    T px, py, cur;
    bool se;
    seg_closest(s, qx, qy, &px, &py, &se, &cur);
    if (t_abs(cur) < t_abs(best)) {
      const bool at2 = (px == s.p2x && py == s.p2y);
      const bool at1 = (px == s.p1x && py == s.p1y);
      if (se && (c > 0 || at2) && (c < nseg - 1 || at1)) {
        const Seg<T> sc = load_seg<T>(base + c * kSegStride + (at1 ? 7 : 14));
        cur *= seg_side(sc, qx, qy) ? sgn(cur) : -sgn(cur);
      }
      best = cur;
      out.cx = px;
      out.cy = py;
      out.is_vertex = se;
      best_idx = c;
    }
  }
  out.seg = load_seg<T>(base + best_idx * kSegStride);
  out.ssd = best;
  const Seg<T> s0 = load_seg<T>(base), sl = load_seg<T>(base + (nseg - 1) * kSegStride);
  const T ax = out.cx - s0.p1x, ay = out.cy - s0.p1y;
  const T bx = out.cx - sl.p2x, by = out.cy - sl.p2y;
  out.is_endpoint = (ax * ax + ay * ay < T(1e-4f)) || (bx * bx + by * by < T(1e-4f));
  return out;
}

// ---------------------------------------------------------------------------
// Costs.  `v` is the argument vector (state x or one player's u) in LDS/global,
// `dim` its length.  H is a column-major tile with leading dimension ld.
// ---------------------------------------------------------------------------
// Constraint::Mu(lambda, g), include/ilqgames/constraint/constraint.h:112-117
template <typename T>
__device__ __forceinline__ T constraint_mu(T lambda, T g, T mu) {
  return (g <= T(1e-4f) && t_abs(lambda) <= T(1e-4f)) ? T(0) : mu;
}

// OrientationCost's wrapped heading error, src/orientation_cost.cpp:54-55: the difference is taken in the argument's
// precision, pi is added and the remainder taken in double (M_PI is a double there).
template <typename T>
__device__ __forceinline__ T orientation_difference(T angle, T nominal) {
  const double kPi = 3.14159265358979323846;
  return T(fmod(double(angle - nominal) + kPi, kPi * 2.0) - kPi);
}

// `v` is anything indexable that yields the argument vector's entries: a pointer (LDS / global row) or the
// transposed-row accessor of the lane-per-time-step stage (ilqg_rows.hpp).
// `step`: the time step the term is evaluated at (the time-dependent kinds read their nominal of that step).
template <typename T, typename V>
__device__ __forceinline__ T term_evaluate_leaf_of(const QuadTables<T>& tb, const DevTerm& c, const V& v, int dim,
                                                   int step = 0) {
  const T w = T(c.weight), val = T(c.value);
  const bool oriented = c.flags & ILQG_FLAG_ORIENTED;
  switch (c.kind) {
    case ILQG_COST_NOMINAL_PATH_LENGTH: {  // src/nominal_path_length_cost.cpp:50-56
      const double nom = tb.tnom[(size_t(c.polyline) * tb.tnom_T + step) * 2];
      const T delta = T(double(v[c.idx[0]]) - nom);
      return T(0.5) * w * delta * delta;
    }
    case ILQG_COST_ROUTE_PROGRESS: {  // src/route_progress_cost.cpp:52-64
      const double* nom = tb.tnom + (size_t(c.polyline) * tb.tnom_T + step) * 2;
      const T dx = v[c.idx[0]] - T(nom[0]), dy = v[c.idx[1]] - T(nom[1]);
      return T(0.5) * w * (dx * dx + dy * dy);
    }
    case ILQG_COST_QUADRATIC: {  // src/quadratic_cost.cpp:51-63
      if (c.idx[0] >= 0) {
        const T d = v[c.idx[0]] - val;
        return T(0.5) * w * d * d;
      }
      T sq = 0;
      for (int i = 0; i < dim; i++) sq += (v[i] - val) * (v[i] - val);
      return T(0.5) * w * sq;
    }
    case ILQG_COST_SEMIQUADRATIC: {  // src/semiquadratic_cost.cpp:51-59
      const T d = v[c.idx[0]] - val;
      if ((d > T(0) && oriented) || (d < T(0) && !oriented)) return T(0.5) * w * d * d;
      return T(0);
    }
    case ILQG_COST_QUADRATIC_POLYLINE2: {  // src/quadratic_polyline2_cost.cpp:52-69
      const Closest<T> cl = polyline_closest<T>(tb, c.polyline, v[c.idx[0]], v[c.idx[1]]);
      const T ssd = cl.is_endpoint ? T(0) : cl.ssd;
      return T(0.5) * w * t_abs(ssd);
    }
    case ILQG_COST_SEMIQUADRATIC_POLYLINE2: {  // src/semiquadratic_polyline2_cost.cpp:52-74
      const Closest<T> cl = polyline_closest<T>(tb, c.polyline, v[c.idx[0]], v[c.idx[1]]);
      if (cl.is_endpoint) return T(0);
      const T sst = sgn(val) * val * val;
      const bool active = (cl.ssd > sst && oriented) || (cl.ssd < sst && !oriented);
      if (!active) return T(0);
      const T sd = sgn(cl.ssd) * t_sqrt(t_abs(cl.ssd));
      const T d = sd - val;
      return T(0.5) * w * d * d;
    }
    case ILQG_COST_PROXIMITY: {  // src/proximity_cost.cpp:52-61
      const T dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      const T dsq = dx * dx + dy * dy;
      if (dsq >= val * val) return T(0);
      const T gap = val - t_sqrt(dsq);
      return T(0.5) * w * gap * gap;
    }
    case ILQG_COST_SIGNED_DISTANCE: {  // src/signed_distance_cost.cpp:51-63
      const T dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      const T cost = val - t_hypot(dx, dy);
      return oriented ? cost : -cost;
    }
    case ILQG_COST_QUADRATIC_DIFFERENCE: {  // src/quadratic_difference_cost.cpp:51-59
      const T ex = v[c.idx[0]] - v[c.idx[2]], ey = v[c.idx[1]] - v[c.idx[3]];
      return T(0.5) * w * (ex * ex + ey * ey);
    }
    case ILQG_COST_POLYLINE2_SIGNED_DISTANCE: {  // src/polyline2_signed_distance_cost.cpp:52-65
      const Closest<T> cl = polyline_closest<T>(tb, c.polyline, v[c.idx[0]], v[c.idx[1]]);
      const T ssd = oriented ? cl.ssd : -cl.ssd;
      return sgn(ssd) * t_sqrt(t_abs(ssd)) - val;
    }
    case ILQG_CONSTRAINT_PROXIMITY: {  // src/proximity_constraint.cpp:56-62
      const T dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      const T value = t_hypot(dx, dy) - val;
      return oriented ? value : -value;
    }
    case ILQG_CONSTRAINT_SINGLE_DIMENSION:  // single_dimension_constraint.h:68-70
      return oriented ? v[c.idx[0]] - val : val - v[c.idx[0]];
    case ILQG_COST_ORIENTATION: {  // src/orientation_cost.cpp:50-58
      const T diff = orientation_difference<T>(v[c.idx[0]], val);
      return T(0.5) * w * diff * diff;
    }
    case ILQG_COST_QUADRATIC_NORM: {  // src/quadratic_norm_cost.cpp:52-57
      const T diff = t_hypot(v[c.idx[0]], v[c.idx[1]]) - val;
      return T(0.5) * w * diff * diff;
    }
    case ILQG_COST_SEMIQUADRATIC_NORM: {  // src/semiquadratic_norm_cost.cpp:52-59
      const T diff = t_hypot(v[c.idx[0]], v[c.idx[1]]) - val;
      if ((diff > T(0) && oriented) || (diff < T(0) && !oriented)) return T(0.5) * w * diff * diff;
      return T(0);
    }
    case ILQG_COST_RELATIVE_DISTANCE:  // src/relative_distance_cost.cpp:48-54
      return w * t_hypot(v[c.idx[0]] - v[c.idx[2]], v[c.idx[1]] - v[c.idx[3]]);
    case ILQG_COST_LOCALLY_CONVEX_PROXIMITY: {  // src/locally_convex_proximity_cost.cpp:50-60
      const T dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      if (dx * dx >= val * val || dy * dy >= val * val) return T(0);
      const T ax = val - t_abs(dx), ay = val - t_abs(dy);
      const T sx = ax * ax, sy = ay * ay;
      return T(0.5) * w * (sy < sx ? sy : sx);
    }
    case ILQG_COST_CURVATURE: {  // src/curvature_cost.cpp:50-53
      const T curvature = v[c.idx[0]] / v[c.idx[1]];
      return T(0.5) * w * curvature * curvature;
    }
    case ILQG_COST_WEIGHTED_CONVEX_PROXIMITY: {  // src/weighted_convex_proximity_cost.cpp:50-61
      const WcpIdx ix = wcp_indices(c);
      const T dx = v[ix.x1] - v[ix.x2], dy = v[ix.y1] - v[ix.y2];
      const T v1 = v[ix.v1], v2 = v[ix.v2];
      const T vv = v1 * v1 + v2 * v2;
      if (dx * dx >= val * val || dy * dy >= val * val) return T(0);
      const T ax = val - t_abs(dx), ay = val - t_abs(dy);
      const T sx = ax * ax, sy = ay * ay;
      return T(0.5) * w * vv * (sy < sx ? sy : sx);
    }
    case ILQG_CONSTRAINT_POLYLINE2_SIGNED_DISTANCE: {  // src/polyline2_signed_distance_constraint.cpp:57-69
      const Closest<T> cl = polyline_closest<T>(tb, c.polyline, v[c.idx[0]], v[c.idx[1]]);
      const T value = sgn(cl.ssd) * t_sqrt(t_abs(cl.ssd)) - val;
      return oriented ? value : -value;
    }
    case ILQG_CONSTRAINT_AFFINE_SCALAR:
    case ILQG_CONSTRAINT_AFFINE_VECTOR:
      return tb.dense ? affine_evaluate<T, V>(c.kind, tb.dense + c.polyline, v, dim) : T(0);
  }
  return T(0);
}
template <typename T, typename V>
__device__ __forceinline__ T term_evaluate_leaf(const QuadTables<T>& tb, int ti, const V& v, int dim, int step = 0) {
  const DevTerm c = tb.terms[ti];
  return term_evaluate_leaf_of<T, V>(tb, c, v, dim, step);
}

// ExtremeValueCost::ExtremeCost, src/extreme_value_cost.cpp:66-85: index of the active child.
template <typename T, typename V>
__device__ __forceinline__ int extreme_child(const QuadTables<T>& tb, const DevTerm& c, const V& v, int dim, T* value_out) {
  const bool is_min = c.flags & ILQG_FLAG_IS_MIN;
  T ext = is_min ? dinf<T>() : -dinf<T>();
  int best = c.child_begin;
  for (int q = 0; q < c.child_count; q++) {
    const T value = term_evaluate_leaf<T, V>(tb, c.child_begin + q, v, dim);
    if ((is_min && value < ext) || (!is_min && value > ext)) {
      ext = value;
      best = c.child_begin + q;
    }
  }
  *value_out = ext;
  return best;
}

// Constraint::ModifyDerivatives, src/constraint.cpp:63-89
template <typename T>
__device__ __forceinline__ void modify_derivatives(T lambda, T mu_in, T g, T* dx, T* ddx, T* dy, T* ddy,
                                                   T* dxdy) {
  const T mu = constraint_mu(lambda, g, mu_in);
  const T ndx = lambda * *dx + mu * g * *dx;
  const T nddx = lambda * *ddx + mu * (*dx * *dx + g * *ddx);
  if (dy) {
    const T ndy = lambda * *dy + mu * g * *dy;
    const T nddy = lambda * *ddy + mu * (*dy * *dy + g * *ddy);
    const T ndxdy = lambda * *dxdy + mu * (*dy * *dx + g * *dxdy);
    *dy = ndy;
    *ddy = nddy;
    *dxdy = ndxdy;
  }
  *dx = ndx;
  *ddx = nddx;
}

// What one cost term adds to its (Hessian, gradient) tile, as a scatter pattern plus at most
// five scalars — every in-scope Cost::Quadraticize fits one of these shapes:
//   SINGLE (d):            G[d] += gx;  H(d,d) += hxx
//   PAIR2  (x,y):          G[x] += gx; G[y] += gy; H(x,x) += hxx; H(y,y) += hyy; H(x,y), H(y,x) += hxy
//   PAIR4  (x1,y1,x2,y2):  the relative-position pattern of ProximityCost / SignedDistanceCost /
//                          ProximityConstraint: with s = (+,+,-,-) over (x1,y1,x2,y2),
//                          G[p] += s_p g_type(p),  H(p,q) += s_p s_q h_type(p),type(q)
//   ALL:                   QuadraticCost with dimension < 0: G[i] += w (v[i]-nominal), H(i,i) += w
//   CROSS4 (p1,p2,v1,v2):  the position x speed block of WeightedConvexProximityCost, no gradient part:
//                          H(p1,v1) += hxx, H(p1,v2) += hyy, H(p2,v1) -= hxx, H(p2,v2) -= hyy, and the transposed four
enum { PAT_NONE = 0, PAT_SINGLE = 1, PAT_PAIR2 = 2, PAT_PAIR4 = 3, PAT_ALL = 4, PAT_CROSS4 = 5 };


template <typename T>
struct TermOut {
  int pattern;
  int i0, i1, i2, i3;
  T gx, gy, hxx, hyy, hxy;
  T value;  // Cost::Evaluate (only meaningful for cost terms)
};

// Cost::Evaluate + Cost::Quadraticize of one LEAF term in one pass (the polyline closest-point
// search is shared).  `lambda`, `mu`: augmented-Lagrangian state of a constraint term.
// `pre`: the closest point of this term's polyline to its position, when the caller already has it.  HAVE_PRE: the
// caller always has it (the polyline searches are compiled out of this function).
// `tnom`: this step's nominal pair of a time-dependent term (term_is_time_dependent).
template <typename T, typename V, bool HAVE_PRE = false>
__device__ __forceinline__ void term_compute_leaf(const QuadTables<T>& tb, const DevTerm& c, const V& v, T lambda,
                                                  T mu, TermOut<T>* o, const Closest<T>* pre = nullptr,
                                                  const double* tnom = nullptr) {
  const T w = T(c.weight), val = T(c.value);
  const bool oriented = c.flags & ILQG_FLAG_ORIENTED;
  o->pattern = PAT_NONE;
  o->value = T(0);
  o->i0 = c.idx[0]; o->i1 = c.idx[1]; o->i2 = c.idx[2]; o->i3 = c.idx[3];
  o->gx = o->gy = o->hxx = o->hyy = o->hxy = T(0);
  switch (c.kind) {
    case ILQG_COST_NOMINAL_PATH_LENGTH: {  // src/nominal_path_length_cost.cpp:50-78
      const T delta = T(double(v[c.idx[0]]) - tnom[0]);
      o->value = T(0.5) * w * delta * delta;
      o->pattern = PAT_SINGLE;
      o->gx = w * delta;
      o->hxx = w;
      return;
    }
    case ILQG_COST_ROUTE_PROGRESS: {  // src/route_progress_cost.cpp:52-110
      const T dx = v[c.idx[0]] - T(tnom[0]), dy = v[c.idx[1]] - T(tnom[1]);
      o->value = T(0.5) * w * (dx * dx + dy * dy);
      o->pattern = PAT_PAIR2;
      o->gx = w * dx; o->gy = w * dy; o->hxx = w; o->hyy = w;
      return;
    }
    case ILQG_COST_QUADRATIC: {  // src/quadratic_cost.cpp:51-94
      if (c.idx[0] >= 0) {
        const T d = v[c.idx[0]] - val;
        o->value = T(0.5) * w * d * d;
        o->pattern = PAT_SINGLE;
        o->gx = w * d;
        o->hxx = w;
      } else {
        T sq = T(0);
        for (int i = 0; i < c.arg_dim; i++) sq += (v[i] - val) * (v[i] - val);
        o->value = T(0.5) * w * sq;
        o->pattern = PAT_ALL;
        o->gx = w;
        o->gy = val;
      }
      return;
    }
    case ILQG_COST_SEMIQUADRATIC: {  // src/semiquadratic_cost.cpp:51-85
      const T d = v[c.idx[0]] - val;
      if ((d > T(0) && oriented) || (d < T(0) && !oriented)) o->value = T(0.5) * w * d * d;
      if ((d < T(0) && oriented) || (d > T(0) && !oriented)) return;
      o->pattern = PAT_SINGLE;
      o->gx = w * d;
      o->hxx = w;
      return;
    }
    case ILQG_COST_QUADRATIC_POLYLINE2:        // src/quadratic_polyline2_cost.cpp:52-126
    case ILQG_COST_SEMIQUADRATIC_POLYLINE2: {  // src/semiquadratic_polyline2_cost.cpp:52-142
      const bool semi = c.kind == ILQG_COST_SEMIQUADRATIC_POLYLINE2;
      const T px = v[c.idx[0]], py = v[c.idx[1]];
      Closest<T> cl;
      if constexpr (HAVE_PRE) {
        cl = *pre;
      } else {
        cl = pre != nullptr ? *pre : polyline_closest<T>(tb, c.polyline, px, py);
      }
      T dx, dy;
      if (semi) {
        const T sst = sgn(val) * val * val;
        const bool active = (cl.ssd > sst && oriented) || (cl.ssd < sst && !oriented);
        if (!cl.is_endpoint && active) {
          const T sd = sgn(cl.ssd) * t_sqrt(t_abs(cl.ssd));
          const T d = sd - val;
          o->value = T(0.5) * w * d * d;
        }
        if (!active) return;
        if (cl.is_endpoint) return;
        T scaling = t_sqrt(t_abs(cl.ssd));
        scaling = (scaling - t_abs(val)) / scaling;
        dx = w * scaling * (px - cl.cx);
        dy = w * scaling * (py - cl.cy);
      } else {
        o->value = T(0.5) * w * t_abs(cl.is_endpoint ? T(0) : cl.ssd);
        if (cl.is_endpoint) return;
        dx = w * (px - cl.cx);
        dy = w * (py - cl.cy);
      }
      T ddx = w, ddy = w, dxdy = T(0);
      if (!cl.is_vertex) {
        const T relx = px - cl.seg.p1x, rely = py - cl.seg.p1y;
        ddx = w * cl.seg.uy * cl.seg.uy;
        ddy = w * cl.seg.ux * cl.seg.ux;
        dxdy = -w * cl.seg.ux * cl.seg.uy;
        const T w_cross = semi ? w * (relx * cl.seg.uy - rely * cl.seg.ux - val)
                               : w * (relx * cl.seg.uy - rely * cl.seg.ux);
        dx = w_cross * cl.seg.uy;
        dy = -w_cross * cl.seg.ux;
      }
      o->pattern = PAT_PAIR2;
      o->gx = dx; o->gy = dy; o->hxx = ddx; o->hyy = ddy; o->hxy = dxdy;
      return;
    }
    case ILQG_COST_QUADRATIC_DIFFERENCE: {  // src/quadratic_difference_cost.cpp:51-91 (two dimension pairs)
      const T ex = v[c.idx[0]] - v[c.idx[2]], ey = v[c.idx[1]] - v[c.idx[3]];
      o->value = T(0.5) * w * (ex * ex + ey * ey);
      o->pattern = PAT_PAIR4;  // +w on the diagonals, -w between partners; the cross terms get an exact +-0
      o->gx = w * ex; o->gy = w * ey; o->hxx = w; o->hyy = w; o->hxy = T(0);
      return;
    }
    case ILQG_COST_POLYLINE2_SIGNED_DISTANCE: {  // src/polyline2_signed_distance_cost.cpp:52-126
      const T px = v[c.idx[0]], py = v[c.idx[1]];
      Closest<T> cl;
      if constexpr (HAVE_PRE) cl = *pre; else cl = polyline_closest<T>(tb, c.polyline, px, py);
      const T ssd = oriented ? cl.ssd : -cl.ssd;
      const T sign = sgn(ssd);
      const T distance = t_sqrt(t_abs(ssd));
      o->value = sign * distance - val;
      const T ex = px - cl.cx, ey = py - cl.cy;
      const T denom = ssd * distance;
      o->pattern = PAT_PAIR2;
      if (cl.is_vertex) {
        o->gx = sign * ex / distance; o->gy = sign * ey / distance;
        o->hxx = ey * ey / denom; o->hyy = ex * ex / denom; o->hxy = -ex * ey / denom;
      } else {  // as written there: the segment normal, whatever the orientation flag
        o->gx = cl.seg.uy; o->gy = -cl.seg.ux;
      }
      return;
    }
    case ILQG_COST_PROXIMITY: {  // src/proximity_cost.cpp:52-122
      const T dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      const T dsq = dx * dx + dy * dy;
      if (dsq >= val * val) return;
      const T delta = t_sqrt(dsq);
      const T gap = val - delta;
      o->value = T(0.5) * w * gap * gap;
      const T wd = w / delta;
      const T dxd = dx / delta, dyd = dy / delta;
      o->pattern = PAT_PAIR4;
      o->gx = -wd * gap * dx;
      o->gy = -wd * gap * dy;
      o->hxx = wd * (dxd * (gap * dxd + dx) - gap);
      o->hyy = wd * (dyd * (gap * dyd + dy) - gap);
      o->hxy = wd * (dxd * (gap * dyd + dy));
      return;
    }
    case ILQG_COST_SIGNED_DISTANCE: {  // src/signed_distance_cost.cpp:51-113
      const T s = oriented ? T(1) : T(-1);
      const T ex = v[c.idx[0]] - v[c.idx[2]], ey = v[c.idx[1]] - v[c.idx[3]];
      const T norm = t_hypot(ex, ey);
      const T cost = val - norm;
      o->value = oriented ? cost : -cost;
      const T n3 = norm * norm * norm;
      o->pattern = PAT_PAIR4;
      o->gx = -s * ex / norm;
      o->gy = -s * ey / norm;
      o->hxx = -s * ey * ey / n3;
      o->hyy = -s * ex * ex / n3;
      o->hxy = s * ex * ey / n3;
      return;
    }
    case ILQG_CONSTRAINT_PROXIMITY: {  // src/proximity_constraint.cpp:56-116
      const T dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      const T prox = t_hypot(dx, dy);
      const T sign = oriented ? T(1) : T(-1);
      const T g = sign * (prox - val);
      o->value = g;
      const T rdx = dx / prox, rdy = dy / prox;
      T gx = sign * rdx, gy = sign * rdy;
      T hxx = sign * (T(1) - rdx * rdx) / prox;
      T hyy = sign * (T(1) - rdy * rdy) / prox;
      T hxy = -sign * rdx * rdy / prox;
      modify_derivatives(lambda, mu, g, &gx, &hxx, &gy, &hyy, &hxy);
      o->pattern = PAT_PAIR4;
      o->gx = gx; o->gy = gy; o->hxx = hxx; o->hyy = hyy; o->hxy = hxy;
      return;
    }
    case ILQG_CONSTRAINT_SINGLE_DIMENSION: {  // single_dimension_constraint.h:68-97
      const T sign = oriented ? T(1) : T(-1);
      const T g = sign * (v[c.idx[0]] - val);
      o->value = g;
      T dx = sign, ddx = T(0);
      modify_derivatives<T>(lambda, mu, g, &dx, &ddx, nullptr, nullptr, nullptr);
      o->pattern = PAT_SINGLE;
      o->gx = dx;
      o->hxx = ddx;
      return;
    }
    case ILQG_COST_ORIENTATION: {  // src/orientation_cost.cpp:50-80
      const T diff = orientation_difference<T>(v[c.idx[0]], val);
      o->value = T(0.5) * w * diff * diff;
      o->pattern = PAT_SINGLE;
      o->gx = w * diff;
      o->hxx = w;
      return;
    }
    case ILQG_COST_QUADRATIC_NORM:        // src/quadratic_norm_cost.cpp:52-94
    case ILQG_COST_SEMIQUADRATIC_NORM: {  // src/semiquadratic_norm_cost.cpp:52-99
      const bool semi = c.kind == ILQG_COST_SEMIQUADRATIC_NORM;
      const T x = v[c.idx[0]], y = v[c.idx[1]];
      const T hyp = t_hypot(x, y);
      const T diff = hyp - val;
      if (!semi || (diff > T(0) && oriented) || (diff < T(0) && !oriented)) o->value = T(0.5) * w * diff * diff;
      if (semi && ((hyp > val && !oriented) || (hyp < val && oriented))) return;
      // the norm of the derivatives: hypot and norm * norm^2 in the one-sided cost, sqrt(x^2 + y^2) and norm * (x^2 +
      // y^2) in the two-sided one (quadratic_norm_cost.cpp:75-77, semiquadratic_norm_cost.cpp:75, 81-82)
      const T norm_sq_q = x * x + y * y;
      const T norm = semi ? hyp : t_sqrt(norm_sq_q);
      const T norm3 = norm * (semi ? norm * norm : norm_sq_q);
      o->pattern = PAT_PAIR2;
      o->gx = -w * x * (T(-1) + val / norm);
      o->gy = -w * y * (T(-1) + val / norm);
      o->hxx = w - (val * y * y * w) / norm3;
      o->hyy = w - (val * x * x * w) / norm3;
      o->hxy = val * x * y * w / norm3;
      return;
    }
    case ILQG_COST_RELATIVE_DISTANCE: {  // src/relative_distance_cost.cpp:48-104
      const T ex = v[c.idx[0]] - v[c.idx[2]], ey = v[c.idx[1]] - v[c.idx[3]];
      const T dist = t_hypot(ex, ey);
      o->value = w * dist;
      const T dist3 = dist * dist * dist;
      o->pattern = PAT_PAIR4;
      o->gx = w * ex / dist;
      o->gy = w * ey / dist;
      o->hxx = w * ey * ey / dist3;
      o->hyy = w * ex * ex / dist3;
      o->hxy = -w * ex * ey / dist3;
      return;
    }
    case ILQG_COST_LOCALLY_CONVEX_PROXIMITY: {  // src/locally_convex_proximity_cost.cpp:50-108
      const T dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      if (dx * dx >= val * val || dy * dy >= val * val) return;
      const T ax = val - t_abs(dx), ay = val - t_abs(dy);
      const T sx = ax * ax, sy = ay * ay;
      o->value = T(0.5) * w * (sy < sx ? sy : sx);
      o->pattern = PAT_PAIR4;  // one axis only: the other one's entries get an exact +-0
      if (sx < sy) {
        o->gx = -w * ax; o->hxx = w;
      } else {
        o->gy = -w * ay; o->hyy = w;
      }
      return;
    }
    case ILQG_COST_WEIGHTED_CONVEX_PROXIMITY:  // src/weighted_convex_proximity_cost.cpp:50-158, as written there
    case ILQG_INTERNAL_WCP_SPEED:
    case ILQG_INTERNAL_WCP_CROSS_X:
    case ILQG_INTERNAL_WCP_CROSS_Y: {
      const WcpIdx ix = wcp_indices(c);
      const T dx = v[ix.x1] - v[ix.x2], dy = v[ix.y1] - v[ix.y2];
      const T v1 = v[ix.v1], v2 = v[ix.v2];
      const T vv = v1 * v1 + v2 * v2;
      if (dx * dx >= val * val || dy * dy >= val * val) return;
      const T ax = val - t_abs(dx), ay = val - t_abs(dy);
      const T sx = ax * ax, sy = ay * ay;
      const bool x_active = sx < sy;
      const T delta = x_active ? ax : ay, d = x_active ? dx : dy;
      if (c.kind == ILQG_COST_WEIGHTED_CONVEX_PROXIMITY) {
        o->value = T(0.5) * w * vv * (sy < sx ? sy : sx);
        o->pattern = PAT_PAIR4;  // one axis only: the other one's entries get an exact +-0
        if (x_active) { o->gx = -w * delta * vv; o->hxx = w; } else { o->gy = -w * delta * vv; o->hyy = w; }
      } else if (c.kind == ILQG_INTERNAL_WCP_SPEED) {
        o->pattern = PAT_PAIR2;
        o->gx = -w * v1 * delta * delta;
        o->gy = -w * v2 * delta * delta;
        o->hxx = w * delta * delta;
        o->hyy = o->hxx;
      } else if (x_active == (c.kind == ILQG_INTERNAL_WCP_CROSS_X)) {
        o->pattern = PAT_CROSS4;
        o->hxx = T(-2.0) * w * v1 * sgn(d);
        o->hyy = T(-2.0) * w * v2 * sgn(d);
      }
      return;
    }
    case ILQG_COST_CURVATURE: {  // src/curvature_cost.cpp:50-86: idx = (omega, v)
      const T omega = v[c.idx[0]], vel = v[c.idx[1]];
      const T curvature = omega / vel;
      o->value = T(0.5) * w * curvature * curvature;
      const T one_over_vsq = T(1) / (vel * vel);
      const T weight_over_vsq = w * one_over_vsq;
      const T weight_omega_over_vsq = omega * weight_over_vsq;
      o->pattern = PAT_PAIR2;
      o->gx = weight_omega_over_vsq;
      o->gy = -weight_omega_over_vsq * omega / vel;
      o->hxx = weight_over_vsq;
      o->hxy = T(-2) * weight_omega_over_vsq / vel;
      o->hyy = T(3) * weight_omega_over_vsq * omega * one_over_vsq;
      return;
    }
    case ILQG_CONSTRAINT_POLYLINE2_SIGNED_DISTANCE: {  // src/polyline2_signed_distance_constraint.cpp:57-144
      const T x = v[c.idx[0]], y = v[c.idx[1]];
      Closest<T> cl;
      if constexpr (HAVE_PRE) cl = *pre; else cl = polyline_closest<T>(tb, c.polyline, x, y);
      const T s = sgn(cl.ssd);
      const T sign = oriented ? T(1) : T(-1);
      const T sd = s * t_sqrt(t_abs(cl.ssd));
      const T g = oriented ? sd - val : val - sd;
      o->value = g;
      T dx = sign * cl.seg.uy, ddx = T(0), dy = -sign * cl.seg.ux, ddy = T(0), dxdy = T(0);
      if (cl.is_vertex) {
        const T px = cl.cx, py = cl.cy;
        const T rx = x - px, ry = y - py;
        const T d_sq = rx * rx + ry * ry;
        const T d = t_sqrt(d_sq);
        dx = sign * s * rx / d;
        ddx = sign * s * (d_sq - px * px - x * x + T(2) * px * x) / (d_sq * d);
        dxdy = -sign * s * rx * ry / (d_sq * d);
        dy = sign * s * ry / d;
        ddy = sign * s * (d_sq - py * py - y * y + T(2) * py * y) / (d_sq * d);
      }
      modify_derivatives(lambda, mu, g, &dx, &ddx, &dy, &ddy, &dxdy);
      o->pattern = PAT_PAIR2;
      o->gx = dx; o->gy = dy; o->hxx = ddx; o->hyy = ddy; o->hxy = dxdy;
      return;
    }
  }
}

#pragma clang fp contract(fast)  // back to the translation unit's default

}  // namespace ilqg
