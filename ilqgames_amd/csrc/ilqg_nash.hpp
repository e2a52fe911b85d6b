// Equilibrium checks on the device: ComputeStrategyCosts (src/compute_strategy_costs.cpp:61-106) and
// NumericalCheckLocalNashEquilibrium (src/check_local_nash_equilibrium.cpp:60-133).
// The check plays 1 + 2 m (T-1) variants of one instance's strategies (every entry of every alpha_i[k], k < T-1,
// moved down and up by max_perturbation) — ~1200 independent rollouts per instance for the intersection games,
// one wavefront each: grid = (instance, move).  A second kernel folds the costs into the verdict.
#pragma once
#include "ilqg_common.hpp"
#include "ilqg_models.hpp"
#include "ilqg_stages.hpp"

namespace ilqg {

template <typename T>
struct StrategyCostArgs {
  const T *x0, *xs, *us, *P, *alpha;  // [B][n], [B][T][n], [B][T][m], [B][T][m*n], [B][T][m]
  T* costs;                           // [moves][B][N]; move 0 = the strategies as given
  T eps;                              // max_perturbation
  int open_loop, euler, batch;
};

// LDS elements behind the cost tables: [x | u], [next x | u], term values.
__host__ __device__ inline size_t strategy_cost_lds_elems(int n, int m, int num_terms) {
  return size_t(2 * (n + m) + (num_terms > 0 ? num_terms : 1) + 4 + 2);  // +2: a one-control model's unused second input
}

template <typename T>
__global__ void __launch_bounds__(64) strategy_costs_kernel(DevProblem p, StrategyCostArgs<T> a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const QuadTables<T> tb = quad_tables_load<T>(p, smem_raw);
  T* sx = reinterpret_cast<T*>(smem_raw + quad_tables_bytes(p, sizeof(T)));  // [x | u]
  const int n = p.n, m = p.m, N = p.N, Tn = p.T;
  T* snx = sx + n + m;       // [next x | u]
  T* sval = snx + n + m;     // [num_terms]
  const size_t b = blockIdx.x;
  const int q = blockIdx.y;
  const int t = threadIdx.x;
  // move q > 0: entry `ja` of the stacked alpha at step `kp`, lower (-) before upper (+)
  const int r = (q - 1) >> 1;
  const int kp = q > 0 ? r / m : -1, ja = q > 0 ? r % m : -1;
  const T shift = (q > 0) ? (((q - 1) & 1) ? a.eps : -a.eps) : T(0);
  const T* xs = a.xs + b * Tn * n;
  const T* us = a.us + b * Tn * m;
  const T* P = a.P + b * Tn * m * n;
  const T* al = a.alpha + b * Tn * m;
  if (t < n) sx[t] = a.x0[b * n + t];
  __syncthreads();
  T acc = T(0);  // lane i < N: player i's total
  const int steps = a.open_loop ? Tn - 1 : Tn;
  for (int kk = 0; kk < steps; kk++) {
    if (t < m) {  // Strategy::operator() (strategy.h:73-76); open loop: delta_x = 0 (compute_strategy_costs.cpp:79-81)
      T s = T(0);
      if (!a.open_loop)
        for (int c = 0; c < n; c++) s += P[(size_t(kk) * n + c) * m + t] * (sx[c] - xs[size_t(kk) * n + c]);
      T alpha = al[size_t(kk) * m + t];
      if (kk == kp && t == ja) alpha += shift;
      const T u = (us[size_t(kk) * m + t] - s) - alpha;
      sx[n + t] = u;
      snx[n + t] = u;
    }
    __syncthreads();
    if (t < N) {  // dynamics.Integrate(t, dt, x, us) (:85)
      const int xo = p.xoff[t], uo = p.uoff[t], xd = p.xoff[t + 1] - xo;
      T xj[kSubStatesMax], f[kSubStatesMax];
      for (int e = 0; e < kSubStatesMax; e++) xj[e] = e < xd ? sx[xo + e] : T(0);
      const bool dist = p.sub_kind[t] == ILQG_DYN_UNICYCLE_4D_DISTURBED;
      const bool air = p.sub_kind[t] == ILQG_DYN_AIR_3D_EVADER;  // d0 carries the pursuer's speed there
      const T d0 = dist ? sx[n + uo + 2] : (air ? T(p.sub_param[t + 1]) : T(0)), d1 = dist ? sx[n + uo + 3] : T(0);
      if (a.euler) {  // multi_player_dynamical_system.cpp:57-58
        sub_eval8<T>(p.sub_kind[t], T(p.sub_param[t]), xj, sx[n + uo], sx[n + uo + 1], f, d0, d1);
        for (int e = 0; e < kSubStatesMax; e++) xj[e] += T(p.dt) * f[e];
      } else {
        sub_integrate8<T>(p.sub_kind[t], T(p.sub_param[t]), p.dt, xj, sx[n + uo], sx[n + uo + 1], d0, d1);
      }
      for (int e = 0; e < xd; e++) snx[xo + e] = xj[e];
    }
    __syncthreads();
    // PlayerCost::Evaluate(t, x, us) / EvaluateOffset(t, next_t, next_x, us) (player_cost.cpp:128-144,175-190)
    const T* at = a.open_loop ? snx : sx;
    for (int ti = t; ti < p.num_terms; ti += 64) {
      const DevTerm c = tb.terms[ti];
      T value = T(0);
      // FinalTimeCost gates on the time the term is evaluated AT: next_t for state costs in the open-loop form
      const int at_step = (a.open_loop && c.role == ILQG_ROLE_STATE_COST) ? kk + 1 : kk;
      if ((c.role == ILQG_ROLE_STATE_COST || c.role == ILQG_ROLE_CONTROL_COST) && at_step >= c.k_start) {
        if (c.kind == ILQG_COST_EXTREME_VALUE)
          (void)extreme_child<T>(tb, c, at + c.arg_off, c.arg_dim, &value);
        else
          value = term_evaluate_leaf<T>(tb, ti, at + c.arg_off, c.arg_dim, at_step);
      }
      sval[ti] = value;
    }
    __syncthreads();
    if (t < N) {
      const int* order = tb.order + t * p.cost_order_stride;
      T total = T(0);
      for (int e = 0; e < order[0]; e++) total += sval[order[1 + e]];
      acc += total;
    }
    if (t < n) sx[t] = snx[t];
    __syncthreads();
  }
  if (t < N) a.costs[(size_t(q) * a.batch + b) * N + t] = acc;
}

// is_nash[b] = no move lowered its mover's cost; margin[b] = min over moves of (moved - nominal).
template <typename T>
__global__ void __launch_bounds__(64) nash_verdict_kernel(DevProblem p, const T* costs, int moves, int batch,
                                                          int* is_nash, T* margin) {
  const size_t b = blockIdx.x;
  const int t = threadIdx.x, N = p.N, m = p.m;
  T worst = dinf<T>();
  for (int q = 1 + t; q < moves; q += 64) {
    const int ja = ((q - 1) >> 1) % m;
    int mover = 0;
    for (int i = 0; i < N; i++)
      if (ja >= p.uoff[i]) mover = i;
    const T diff = costs[(size_t(q) * batch + b) * N + mover] - costs[b * N + mover];
    worst = diff < worst ? diff : worst;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const T o = __shfl_xor(worst, off, 64);
    worst = o < worst ? o : worst;
  }
  if (t == 0) {
    is_nash[b] = worst < T(0) ? 0 : 1;
    if (margin) margin[b] = worst;
  }
}

// ---------------------------------------------------------------------------------------------
// CheckSufficientLocalNashEquilibrium (src/check_local_nash_equilibrium.cpp:144-201): is every Q_i and R_ij of a
// full quadraticisation free of eigenvalues below -kErrorMargin = -1e-4?  A symmetric M has no eigenvalue below
// -eps exactly when M + eps I is positive semidefinite, which a Cholesky factorisation decides (it breaks down on
// a non-positive pivot otherwise) without computing the spectrum: one lane per matrix, in place in the scratch
// copy the quadraticisation kernel wrote.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ bool cholesky_is_positive(T* a, int d, T shift) {
  for (int j = 0; j < d; j++) {
    T s = a[j + d * j] + shift;
    for (int k = 0; k < j; k++) s -= a[j + d * k] * a[j + d * k];
    if (!(s > T(0))) return false;
    const T ljj = t_sqrt(s);
    a[j + d * j] = ljj;
    for (int i = j + 1; i < d; i++) {
      T v = a[i + d * j];
      for (int k = 0; k < j; k++) v -= a[i + d * k] * a[j + d * k];
      a[i + d * j] = v / ljj;
    }
  }
  return true;
}

// grid = (T, instances of this chunk); ok[b] must hold 1 on entry and is cleared by any failing matrix.
template <typename T>
__global__ void __launch_bounds__(64) psd_check_kernel(DevProblem p, T* Q, T* R, int* ok) {
  const int k = blockIdx.x;
  const size_t b = blockIdx.y;
  const int t = threadIdx.x, n = p.n, N = p.N;
  const T shift = T(1e-4f);  // kErrorMargin (:174)
  bool good = true;
  if (t < N) {
    good = cholesky_is_positive<T>(Q + ((b * p.T + k) * N + t) * size_t(n) * n, n, shift);
  } else if (t < N + p.pairs.npairs) {
    const int q = t - N;
    const int mj = p.udim[p.pairs.pj[q]];
    good = cholesky_is_positive<T>(R + (b * p.T + k) * size_t(p.pairs.Rsz) + p.pairs.roff[q], mj, shift);
  }
  if (!good) atomicAnd(ok + b, 0);
}

template <typename T>
__global__ void fill_int_kernel(int* v, int value, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) v[i] = value;
}

}  // namespace ilqg
