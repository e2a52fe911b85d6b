// Receding-horizon harness kernels: one wavefront per instance, plans stored [B][cap][...] with a
// per-instance length and start time (SolutionSplicer keeps up to five rows of the old plan in front
// of a new solution, so a stored plan is T .. T+5 rows long).
//
//   plan_integrate_kernel   MultiPlayerIntegrableSystem::Integrate(t0, t, x0, operating_point, strategies)
//                           (src/multi_player_integrable_system.cpp:54-155)
//   receding_sync_kernel    Problem::SyncToExistingProblem + SetUpNextRecedingHorizon (src/problem.cpp:64-186)
//   splice_kernel           SolutionSplicer::SolutionSplicer / Splice (src/solution_splicer.cpp:56-129)
//
// These are bookkeeping kernels (a few hundred rows moved per instance per MPC step); the solves
// between them are where the time goes.  All time arithmetic is double, as in the reference (Time = double).
#pragma once
#include "ilqg_common.hpp"
#include "ilqg_models.hpp"

namespace ilqg {

// constants::kSmallNumber is a float (types.h); it enters double expressions with its float value
constexpr float kSmallNumberF = 1e-4f;

template <typename T>
struct PlanRef {  // one instance's stored plan
  const T *xs, *us, *P, *al;
  int len;
  double t0;
};

// State stepping under a stored plan: Strategy::operator() (strategy.h:73-76) + Integrate (RK4, two sub-steps).
// sx [n] holds the state, su [m] the controls; every thread of the wavefront calls.
template <typename T>
struct PlanStepper {
  const DevProblem& p;
  PlanRef<T> pl;
  T *sx, *su;
  int t;
  __device__ __forceinline__ void controls(int k, bool interpolate, float frac_f) {
    const int n = p.n, m = p.m;
    if (t < m) {
      const T frac = T(frac_f);
      T s = T(0);
      for (int c = 0; c < n; c++) {
        T ref;
        if (!interpolate)
          ref = pl.xs[size_t(k) * n + c];
        else if (k + 1 < pl.len)
          ref = frac * pl.xs[size_t(k) * n + c] + (T(1) - frac) * pl.xs[size_t(k + 1) * n + c];
        else
          ref = pl.xs[size_t(pl.len - 1) * n + c];
        s += pl.P[(size_t(k) * n + c) * m + t] * (sx[c] - ref);
      }
      su[t] = (pl.us[size_t(k) * m + t] - s) - pl.al[size_t(k) * m + t];
    }
    __syncthreads();
  }
  __device__ __forceinline__ void integrate(double interval) {
    if (t < p.N) {
      const int xo = p.xoff[t], uo = p.uoff[t], xd = p.xoff[t + 1] - xo;
      T xj[kSubStatesMax];
      for (int e = 0; e < kSubStatesMax; e++) xj[e] = e < xd ? sx[xo + e] : T(0);
      const bool dist = p.sub_kind[t] == ILQG_DYN_UNICYCLE_4D_DISTURBED;  // the next player's (dx, dy)
      const bool air = p.sub_kind[t] == ILQG_DYN_AIR_3D_EVADER;           // the next row's parameter: pursuer speed
      sub_integrate8<T>(p.sub_kind[t], T(p.sub_param[t]), interval, xj, su[uo], su[uo + 1],
                       dist ? su[uo + 2] : (air ? T(p.sub_param[t + 1]) : T(0)), dist ? su[uo + 3] : T(0));
      for (int e = 0; e < xd; e++) sx[xo + e] = xj[e];
    }
    __syncthreads();
  }
  // IntegrateToNextTimeStep (:95-130)
  __device__ __forceinline__ void to_next_step(double t_abs) {
    const double rel = t_abs - pl.t0;
    const size_t ks = static_cast<size_t>((rel + kSmallNumberF) / p.dt);
    const double remaining = p.dt * (ks + 1) - rel;
    controls(int(ks), true, float(remaining / p.dt));
    integrate(remaining);
  }
  // Integrate(initial_timestep, final_timestep, ...) (:76-93)
  __device__ __forceinline__ void whole_steps(int begin, int end) {
    for (int kk = begin; kk < end; kk++) {
      controls(kk, false, 0.0f);
      integrate(p.dt);
    }
  }
  // IntegrateFromPriorTimeStep (:132-155)
  __device__ __forceinline__ void from_prior_step(double t_abs) {
    const double rel = t_abs - pl.t0;
    const size_t ks = static_cast<size_t>(rel / p.dt);
    controls(int(ks), false, 0.0f);
    integrate(rel - p.dt * ks);
  }
};

template <typename T>
struct PlanBuffers {
  T *xs, *us, *P, *al;  // [B][cap][n | m | m*n | m]
  int* len;             // [B], or null: every plan is T rows
  double* t0;           // [B], or null: every plan starts at uniform_t0
  double uniform_t0;
  int cap;
  __device__ __forceinline__ PlanRef<T> ref(const DevProblem& p, size_t b) const {
    return PlanRef<T>{xs + b * cap * p.n, us + b * cap * p.m, P + b * cap * p.m * p.n, al + b * cap * p.m,
                      len ? len[b] : p.T, t0 ? t0[b] : uniform_t0};
  }
};

// ---------------------------------------------------------------------------------------------
template <typename T>
struct PlanIntegrateArgs {
  PlanBuffers<T> plan;
  double t_from, t_to, must_contain;
  T* x;         // [B][n] in/out
  int* active;  // [B] in/out
};

template <typename T>
__global__ void __launch_bounds__(64) plan_integrate_kernel(DevProblem p, PlanIntegrateArgs<T> a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sx = reinterpret_cast<T*>(smem_raw);
  T* su = sx + p.n;
  const size_t b = blockIdx.x;
  const int t = threadIdx.x;
  if (!a.active[b]) return;
  const PlanRef<T> pl = a.plan.ref(p, b);
  const double dt = p.dt;
  // SolutionSplicer::ContainsTime (solution_splicer.h:66-71) and Integrate's own CHECKs (:57-58,111-112,141-142)
  const bool contains = pl.t0 <= a.must_contain && pl.t0 + pl.len * dt >= a.must_contain;
  bool valid = a.t_to >= a.t_from && a.t_from >= pl.t0;
  const size_t itn = static_cast<size_t>(((a.t_from - pl.t0) + kSmallNumberF) / dt);
  const size_t current = static_cast<size_t>((a.t_from - pl.t0) / dt);
  const size_t final_step = static_cast<size_t>((a.t_to - pl.t0) / dt);
  valid = valid && int(itn) < pl.len && int(final_step) < pl.len;
  if (!contains || !valid) {
    if (t == 0) a.active[b] = 0;
    return;
  }
  if (t < p.n) sx[t] = a.x[b * p.n + t];
  __syncthreads();
  PlanStepper<T> st{p, pl, sx, su, t};
  if (a.t_from > pl.t0) st.to_next_step(a.t_from);
  st.whole_steps(int(current) + 1, int(final_step));
  st.from_prior_step(a.t_to);
  if (t < p.n) a.x[b * p.n + t] = sx[t];
}

// ---------------------------------------------------------------------------------------------
template <typename T>
struct RecedingArgs {
  PlanBuffers<T> plan;           // stored plan (may be the output buffers themselves when cap == T)
  const T* x;                    // [B][n] measured state at time t
  double t, planner_runtime;
  T *xs, *us, *P, *alpha;        // [B][T][...] out: warm start of the next solve
  T* x0_next;                    // [B][n]
  double* solve_t0;              // [B] out (nullable): OperatingPoint::t0 of the next solve
  int* first_step;               // [B]
  int* active;                   // [B] in/out (nullable): cleared where the reference would CHECK-abort
};

template <typename T>
__global__ void __launch_bounds__(64) receding_sync_kernel(DevProblem p, RecedingArgs<T> a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sx = reinterpret_cast<T*>(smem_raw);
  const int n = p.n, m = p.m, Tn = p.T;
  T* su = sx + n;
  const size_t b = blockIdx.x;
  const int t = threadIdx.x;
  if (a.active && !a.active[b]) return;
  const PlanRef<T> pl = a.plan.ref(p, b);
  const double dt = p.dt;
  // ---- SyncToExistingProblem's time bookkeeping (src/problem.cpp:75-102) ----
  const float kRoundingError = 0.9f;
  const double rel = a.t - pl.t0;
  bool valid = !(a.planner_runtime < 0.0 || a.planner_runtime + a.t > pl.t0 + dt * Tn || a.t < pl.t0);  // :68-70
  size_t current = valid ? static_cast<size_t>(rel / dt) : 0;
  double remaining = (current + 1) * dt - rel;
  if (remaining < kRoundingError * dt) {
    current += 1;
    remaining = dt - remaining;
  }
  const size_t itn = valid ? static_cast<size_t>((rel + kSmallNumberF) / dt) : 0;
  double new_t0 = a.t + remaining;
  int int_begin = int(current) + 1, int_end = int_begin;
  if (remaining <= a.planner_runtime) {
    const size_t num_steps = static_cast<size_t>(kSmallNumberF + (a.planner_runtime - remaining) / dt);
    int_end = int(current + num_steps);
    if (int_end < int_begin) int_end = int_begin;
    new_t0 += dt * double(num_steps);
  }
  valid = valid && int(itn) < pl.len && int_end <= pl.len;
  if (!valid) {
    if (t == 0) {
      if (a.active) a.active[b] = 0;
      a.first_step[b] = -1;
    }
    return;
  }
  if (t < n) sx[t] = a.x[b * n + t];
  __syncthreads();
  PlanStepper<T> st{p, pl, sx, su, t};
  st.to_next_step(a.t);
  st.whole_steps(int_begin, int_end);
  // nearest plan state in the first subsystem's metric (concatenated_dynamical_system.cpp:109-113: its position for
  // the car / unicycle / point-mass models and TwoPlayerUnicycle4D, its whole state where the model inherits the default
  // squared norm (SinglePlayerDubinsCar) — DevProblem::sync_dist_dims); std::min_element keeps the first minimum
  T bestd = dinf<T>();
  int bestk = 0x7fffffff;
  for (int k = t; k < pl.len; k += 64) {
    T d = T(0);
    for (int e = 0; e < p.sync_dist_dims; e++) {
      const T de = sx[e] - pl.xs[size_t(k) * n + e];
      d += de * de;
    }
    if (d < bestd) {
      bestd = d;
      bestk = k;
    }
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const T od = __shfl_xor(bestd, off, 64);
    const int ok = __shfl_xor(bestk, off, 64);
    if (od < bestd || (od == bestd && ok < bestk)) {
      bestd = od;
      bestk = ok;
    }
  }
  const int first = bestk;
  if (t == 0) {
    a.first_step[b] = first;
    if (a.solve_t0) a.solve_t0[b] = new_t0;
  }
  // Stitch (concatenated_dynamical_system.h:75-84)
  const int ego = p.xoff[1] - p.xoff[0];
  if (t < n) a.x0_next[b * n + t] = t < ego ? pl.xs[size_t(first) * n + t] : sx[t];
  T* xs = a.xs + b * Tn * n;
  T* us = a.us + b * Tn * m;
  T* P = a.P + b * Tn * m * n;
  T* al = a.alpha + b * Tn * m;
  // rows [first, end) of the plan become rows [0, cnt) (:136-157); ascending, so a plan shifted in place
  // always reads ahead of what it has written
  const int end = (first + Tn < pl.len) ? first + Tn : pl.len;
  const int cnt = end - first;
  __syncthreads();
  if (first > 0 || xs != pl.xs) {
    // Each array as one ascending copy, eight loads per lane in flight: a batch's loads all precede its stores and every
    // later batch reads higher addresses than anything written so far (the source is `first` rows ahead), so the
    // in-place shift stays correct on the one wavefront this kernel runs (row by row, every row's loads waited behind
    // the previous row's stores, which may alias).
    auto shift = [&](T* dst, const T* src, int count) {
      constexpr int U = 8;
      for (int e0 = t; e0 < count; e0 += 64 * U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int e = e0 + 64 * u;
          v[u] = e < count ? src[e] : T(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int e = e0 + 64 * u;
          if (e < count) dst[e] = v[u];
        }
      }
    };
    shift(xs, pl.xs + size_t(first) * n, cnt * n);
    shift(us, pl.us + size_t(first) * m, cnt * m);
    shift(al, pl.al + size_t(first) * m, cnt * m);
    shift(P, pl.P + size_t(first) * m * n, cnt * m * n);
  }
  __syncthreads();
  if (cnt < Tn) {
    // zero strategies / controls of the tail and propagate the state through it (:170-184)
    if (t < n) sx[t] = xs[size_t(cnt - 1) * n + t];
    if (t < m) su[t] = us[size_t(cnt - 1) * m + t];
    __syncthreads();
    for (int kk = cnt; kk < Tn; kk++) {
      st.integrate(dt);  // xs[kk] = Integrate(dt, xs[kk-1], us[kk-1])
      if (t < n) xs[size_t(kk) * n + t] = sx[t];
      if (t < m) {
        su[t] = T(0);
        us[size_t(kk) * m + t] = T(0);
        al[size_t(kk) * m + t] = T(0);
      }
      for (int e = t; e < m * n; e += 64) P[size_t(kk) * m * n + e] = T(0);
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <typename T>
struct SpliceArgs {
  PlanBuffers<T> plan;                // in/out (len and t0 per instance, required)
  const T *xs, *us, *P, *alpha;       // [B][T][...] the new solution
  const double* solve_t0;             // [B] its OperatingPoint::t0
  const int *converged, *active;      // [B], nullable
};

template <typename T>
__global__ void __launch_bounds__(64) splice_kernel(DevProblem p, SpliceArgs<T> a) {
  const int n = p.n, m = p.m, Tn = p.T;
  const size_t b = blockIdx.x;
  const int t = threadIdx.x;
  if (a.active && !a.active[b]) return;
  const int len = a.plan.len[b];
  const double pt0 = a.plan.t0[b], st0 = a.solve_t0[b];
  T* pxs = a.plan.xs + b * a.plan.cap * n;
  T* pus = a.plan.us + b * a.plan.cap * m;
  T* pP = a.plan.P + b * a.plan.cap * m * n;
  T* pal = a.plan.al + b * a.plan.cap * m;
  int at = 0;  // row of the plan the new solution starts at
  if (len > 0) {
    if (a.converged && !a.converged[b]) return;  // receding_horizon_simulator.cpp:133
    if (st0 < pt0) return;                       // CHECK_GE (:61)
    const size_t current = static_cast<size_t>(1e-4 + (st0 - pt0) / p.dt);  // :65-67
    if (int(current) > len) return;
    constexpr size_t kSave = 5;                  // kNumPreviousTimeStepsToSave (:72)
    const size_t initial = (int(current) < int(kSave)) ? 0 : current - kSave;
    if (initial > 0) {
      for (size_t kk = initial; kk < current; kk++) {  // :85-95, ascending
        const size_t d = kk - initial;
        for (int e = t; e < n; e += 64) pxs[d * n + e] = pxs[kk * n + e];
        for (int e = t; e < m; e += 64) {
          pus[d * m + e] = pus[kk * m + e];
          pal[d * m + e] = pal[kk * m + e];
        }
        for (int e = t; e < m * n; e += 64) pP[d * m * n + e] = pP[kk * m * n + e];
        __syncthreads();
      }
    }
    at = int(current - initial);
    if (t == 0) {
      a.plan.len[b] = at + Tn;                     // :100-103
      a.plan.t0[b] = pt0 + initial * p.dt;         // :107
    }
  } else if (t == 0) {  // SolutionSplicer(const SolverLog&), :56-58
    a.plan.len[b] = Tn;
    a.plan.t0[b] = st0;
  }
  const T* sxs = a.xs + b * Tn * n;
  const T* sus = a.us + b * Tn * m;
  const T* sP = a.P + b * Tn * m * n;
  const T* sal = a.alpha + b * Tn * m;
  for (int e = t; e < Tn * n; e += 64) pxs[size_t(at) * n + e] = sxs[e];
  for (int e = t; e < Tn * m; e += 64) {
    pus[size_t(at) * m + e] = sus[e];
    pal[size_t(at) * m + e] = sal[e];
  }
  for (int e = t; e < Tn * m * n; e += 64) pP[size_t(at) * m * n + e] = sP[e];
}

}  // namespace ilqg
