// ilqg_stages.hpp — rollout, linearise+quadraticise and reduction stages for ONE game
// instance per workgroup (gfx950).  Runtime dimensions (these stages are bound by the
// bytes they write and by the sequential rollout chain, not by unrolled math).
#pragma once

#include "ilqg_common.hpp"
#include "ilqg_models.hpp"

namespace ilqg {

// ---------------------------------------------------------------------------
// Rollout — ILQSolver::CurrentOperatingPoint (src/ilq_solver.cpp:174-206).
// Eight lanes per subsystem run the RK4 (2 sub-steps) with one stage each, so the 24 serial
// sin/cos/tan of a step collapse to two libm latencies (sub_integrate_lanes); lanes rho < m
// evaluate u_rho = u_ref - P[rho,:] dx - s*alpha (Strategy::operator(), strategy.h:73-76)
// against dx broadcast through LDS.  The step's (P, alpha, u_ref, x_ref) block is
// prefetched one step ahead so the only exposed latency is the dependent chain.
// ---------------------------------------------------------------------------
template <typename T>
struct RolloutArgs {
  const T* x0;      // [n]
  const T* xs_ref;  // [T][n]
  const T* us_ref;  // [T][m]
  const T* P;       // [T][m*n]
  const T* alpha;   // [T][m]
  T alpha_scale;
  T* xs;            // [T][n]
  T* us;            // [T][m]
};

// [x | dx | u] + two staged [P | alpha | u_ref | x_ref] blocks (the one in use, the one the DMA is filling)
__host__ __device__ inline int rollout_stage_elems(int n, int m) { return (m * n + 2 * m + n + 3) & ~3; }
constexpr int kRolloutGatherElems = 192;  // sub_integrate_stages: 64 heading rates + 64 (x, y) position-rate pairs
__host__ __device__ inline int rollout_lds_elems(int n, int m) {
  return ((2 * n + m + 3) & ~3) + 2 * rollout_stage_elems(n, m) + kRolloutGatherElems;
}

// Workgroup-scope publish / observe of a progress counter in LDS.  Waves of one workgroup share the
// CU's vector L1, so release/acquire at workgroup scope is enough for the global-memory rows the
// counter covers (no cache maintenance on gfx950 outside tgsplit mode).
__device__ __forceinline__ void progress_publish(int* flag, int value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int progress_observe(int* flag) {
  return __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// CN, CM > 0: compile-time state / control dimensions (fully unrolled inner loops); 0: run-time.
// Executed by ONE wavefront (lane t of 64).  `ready` (LDS, may be null) is set to k+1 once row k of
// xs / us is in memory, so other waves of the workgroup can consume the trajectory while it is
// still being integrated.
// DIST: the dynamics may be TwoPlayerUnicycle4D (a disturbed unicycle row + a state-less disturbance row).
// DUB: it may contain SinglePlayerDubinsCar rows.  Both are compile-time so that the common car / unicycle games
// keep the integrator they had.
// AIR: the dynamics are Air3D — its position rates depend on the position itself, so the stage-parallel integrator
// does not apply and lane 0 runs the plain RK4.
// PM: the game may be one of point masses (then every row is one): the plain RK4 again, one copy per lane of the
// group — the right-hand side is two moves and there is no transcendental to take out of the chain.
template <typename T, int CN = 0, int CM = 0, bool DIST = false, bool DUB = false, bool AIR = false, bool PM = false>
__device__ __forceinline__ void rollout_instance(const DevProblem& p, const RolloutArgs<T>& a, T* sm, int t,
                                                 int* ready = nullptr, long long* phacc = nullptr) {
  const int n = CN > 0 ? CN : p.n, m = CM > 0 ? CM : p.m, N = p.N, Tn = p.T;
  constexpr int NT = 64;
  T* sx = sm;       // [n] current state
  T* sdx = sx + n;  // [n]
  T* su = sdx + n;  // [m]
  T* stg = sm + ((2 * n + m + 3) & ~3);  // two staged blocks [P (m*n) | alpha (m) | u_ref (m) | x_ref (n)]
  const int WP = rollout_stage_elems(n, m);
  constexpr int S = int(sizeof(T));
  // The step's gains and references go global -> LDS by DMA, one step ahead, into the block not in use:
  // no registers are held across the integration (the loop is the kernel's tightest spot for registers).
  auto issue = [&](int k, int buf) {
    T* d = stg + buf * WP;
    dma_g2l<NT, false>(a.P + size_t(k) * m * n, d, m * n * S, t);
    dma_g2l<NT, false>(a.alpha + size_t(k) * m, d + m * n, m * S, t);
    dma_g2l<NT, false>(a.us_ref + size_t(k) * m, d + m * n + m, m * S, t);
    dma_g2l<NT, false>(a.xs_ref + size_t(k) * n, d + m * n + 2 * m, n * S, t);
  };
  // lane group g = t / 8 integrates subsystem g; lane q = t % 8 owns RK4 stage q of that group
  T xj[6];
  const int grp = t >> 3, q = t & 7;
  const bool integ = grp < N && t < 64;
  int kind = ILQG_DYN_UNICYCLE_4D, xo = 0, uo = 0, xd = 0;
  T Lp = T(1);
#pragma unroll
  for (int e = 0; e < 6; e++) xj[e] = T(0);
  if (integ) {
    kind = p.sub_kind[grp];
    xo = p.xoff[grp];
    uo = p.uoff[grp];
    xd = p.xoff[grp + 1] - xo;
    Lp = T(p.sub_param[grp]);
#pragma unroll
    for (int e = 0; e < 6; e++) xj[e] = (e < xd) ? a.x0[xo + e] : T(0);
  }
  T* const gth = stg + 2 * WP;  // exchange scratch of the stage-parallel integrator
  const bool any_car = __any(integ && (kind == ILQG_DYN_CAR_5D || kind == ILQG_DYN_CAR_6D));
  issue(0, 0);
  long long rc0 = (kProfile && phacc) ? clock64() : 0, rc1;
#define ILQG_RPH(i) do { if (kProfile && phacc) { __builtin_amdgcn_sched_barrier(0); rc1 = clock64(); __builtin_amdgcn_sched_barrier(0); phacc[i] += rc1 - rc0; rc0 = rc1; } } while (0)
#pragma unroll 1
  for (int k = 0; k < Tn; k++) {
    // Block k of [P | alpha | u_ref | x_ref] was requested a whole step ago, and the rows of step k - 1 were stored
    // then too: this wait finds nothing in flight, and with it the release below is free.
    dma_wait();
    if (ready) progress_publish(ready, k);
    const T* sP = stg + (k & 1) * WP;  // [m*n] gains of this step
    const T* sal = sP + m * n;         // [m]
    const T* sur = sal + m;            // [m] u_ref
    const T* sxr = sur + m;            // [n] x_ref
    {
      // every lane of a group holds the group's state: lane q publishes entry q (one LDS round trip for the row)
      T mine = xj[0];
#pragma unroll
      for (int e = 1; e < 6; e++) mine = (q == e) ? xj[e] : mine;
      if (integ && q < xd) {
        sdx[xo + q] = mine - sxr[xo + q];
        a.xs[size_t(k) * n + xo + q] = mine;
      }
    }
    lds_sync(NT <= 64);
    if (k + 1 < Tn) issue(k + 1, (k + 1) & 1);  // into the block step k - 1 read: a whole step to land
    ILQG_RPH(0);
    if (t < m) {
      T s = T(0);
      if constexpr (CN > 0) {
        // loads in blocks of eight ahead of the reference's left-to-right accumulation (the whole row
        // at once costs 4 CN registers in the kernel's tightest loop)
        constexpr int CH = 8;
#pragma unroll
        for (int c0 = 0; c0 < CN; c0 += CH) {
          T pr[CH], dv[CH];
#pragma unroll
          for (int c = 0; c < CH; c++)
            if (c0 + c < CN) {
              pr[c] = sP[t + m * (c0 + c)];
              dv[c] = sdx[c0 + c];
            }
#pragma unroll
          for (int c = 0; c < CH; c++)
            if (c0 + c < CN) s += pr[c] * dv[c];
        }
      } else {
        for (int c = 0; c < n; c++) s += sP[t + m * c] * sdx[c];
      }
      const T u = (sur[t] - s) - a.alpha_scale * sal[t];
      su[t] = u;
      a.us[size_t(k) * m + t] = u;
    }
    lds_sync(NT <= 64);
    ILQG_RPH(1);
    if (t < 64 && k + 1 < Tn) {  // whole first wave: the exchanges inside need every group lane live
      const T u0 = integ ? su[uo] : T(0), u1 = integ ? su[uo + 1] : T(0);
      if constexpr (AIR) {
        if (grp == 0) sub_integrate<T>(kind, Lp, p.dt, xj, u0, u1, T(p.sub_param[1]));  // every lane of the group keeps the state
      } else if constexpr (DIST) {
        const bool dist = integ && kind == ILQG_DYN_UNICYCLE_4D_DISTURBED;  // the next player's (dx, dy)
        const T d0 = dist ? su[uo + 2] : T(0), d1 = dist ? su[uo + 3] : T(0);
        sub_integrate_stages<T, true>(kind, Lp, p.dt, xj, u0, u1, q, t, gth, false, d0, d1);
      } else if (PM && p.sub_kind[0] == ILQG_DYN_POINT_MASS_2D) {
        sub_integrate<T>(kind, Lp, p.dt, xj, u0, u1);
      } else {
        sub_integrate_stages<T, false, DUB>(kind, Lp, p.dt, xj, u0, u1, q, t, gth, any_car);
      }
    }
    ILQG_RPH(2);
  }
#undef ILQG_RPH
  if (ready) progress_publish(ready, Tn);
}

// ---------------------------------------------------------------------------
// Linearise + quadraticise one time step of one instance:
//   ILQSolver::ComputeLinearization (src/ilq_solver.cpp:437-455),
//   ILQSolver::ComputeCostQuadraticization (:471-490) -> PlayerCost::Quadraticize
//   (src/player_cost.cpp:194-225), plus the per-step pieces of MeritFunction
//   (:400-435) and TotalCosts (:220-257).
// The step's [A | B | Q | l | R | r] image is assembled in LDS (lane i < N walks
// player i's cost list in the reference's accumulation order and scatters <= 16
// entries per term), then streamed out with fully coalesced stores — the stage's
// cost is the ~n^2 N words it has to write.
// ---------------------------------------------------------------------------
// Bytes of LDS the cost tables take (terms, precomputed segments, polyline offsets, cost order).
__host__ __device__ inline size_t quad_tables_bytes(const DevProblem& p, size_t elem) {
  size_t b = size_t(p.total_segs) * kSegStride * elem;          // segs first: keeps T alignment
  b += size_t(p.num_terms > 0 ? p.num_terms : 1) * sizeof(DevTerm);
  b += size_t(p.num_polylines + 1) * sizeof(int);
  b += size_t(p.N) * p.cost_order_stride * sizeof(int);
  b += size_t(LC_COUNT) * sizeof(int);
  return (b + 15) & ~size_t(15);
}

// Cooperative copy of the tables into LDS; every thread of the workgroup calls, then syncs.
template <typename T>
__device__ __forceinline__ QuadTables<T> quad_tables_load(const DevProblem& p, void* region) {
  const int t = threadIdx.x, NT = blockDim.x;
  T* segs = reinterpret_cast<T*>(region);
  const T* gsegs = problem_segs<T>(p);
  for (int e = t; e < p.total_segs * kSegStride; e += NT) segs[e] = gsegs[e];
  int* terms_i = reinterpret_cast<int*>(segs + size_t(p.total_segs) * kSegStride);
  const int* gterms = reinterpret_cast<const int*>(p.terms);
  const int nti = p.num_terms * int(sizeof(DevTerm) / sizeof(int));
  for (int e = t; e < nti; e += NT) terms_i[e] = gterms[e];
  int* poff = terms_i + (p.num_terms > 0 ? p.num_terms : 1) * int(sizeof(DevTerm) / sizeof(int));
  for (int e = t; e <= p.num_polylines; e += NT) poff[e] = p.poly_off[e];
  int* order = poff + p.num_polylines + 1;
  for (int e = t; e < p.N * p.cost_order_stride; e += NT) order[e] = p.cost_order[e];
  int* lc = order + p.N * p.cost_order_stride;
  if (t < kMaxPlayers) {
    lc[LC_KIND + t] = p.sub_kind[t];
    lc[LC_UDIM + t] = p.udim[t];
    lc[LC_PARAM + t] = __float_as_int(p.sub_param[t]);
    lc[LC_SREG + t] = __float_as_int(p.state_reg[t]);
    lc[LC_CREG + t] = __float_as_int(p.control_reg[t]);
    lc[LC_STRUCT + t] = p.structure[t];
    lc[LC_PII + t] = p.pairs.pii[t];
  }
  if (t <= kMaxPlayers) {
    lc[LC_XOFF + t] = p.xoff[t];
    lc[LC_UOFF + t] = p.uoff[t];
  }
  if (t < kMaxPairs) {
    lc[LC_PI + t] = p.pairs.pi[t];
    lc[LC_PJ + t] = p.pairs.pj[t];
    lc[LC_ROFF + t] = p.pairs.roff[t];
    lc[LC_RGOFF + t] = p.pairs.rgoff[t];
    lc[LC_FROMCOST + t] = p.pairs.from_cost[t];
  }
  for (int e = t; e < 4 * kMaxClosestQueries; e += NT) lc[LC_CQTAB + e] = p.cq_tab[e / 4][e % 4];
  for (int e = t; e < 2 * kMaxClosestItems; e += NT) lc[LC_CQITEMS + e] = p.cq_items[e / 2][e % 2];
  __syncthreads();
  QuadTables<T> tb;
  tb.terms = reinterpret_cast<const DevTerm*>(terms_i);
  tb.segs = segs;
  tb.poly_off = poff;
  tb.order = order;
  tb.lc = lc;
  return tb;
}

template <typename T>
struct QuadArgs {
  const T* xs;         // [T][n]
  const T* us;         // [T][m]
  const T* lambdas;    // [num_constraints][T] or nullptr
  T mu;
  const int* t_extreme;  // [N] or nullptr
  double t_init;
  T *A, *Bm;           // [T][n*n], [T][n*m] or nullptr (skip linearisation)
  T *Q, *l, *R, *r;    // or nullptr (skip quadraticisation outputs)
  T* merit_part;       // [T][N][2] = (|r_ii|^2, |l_i|^2) or nullptr
  T* cost_part;        // [T][N] PlayerCost::Evaluate or nullptr
  long long* phacc = nullptr;  // optional phase profile accumulators (registers of the caller)
};

__host__ __device__ inline int quad_lds_elems(int n, int m, int N, int Rsz, int rsz, int num_terms) {
  // + the shared closest-point pre-pass: 4 scalars per segment item, a Closest image per query
  return n + m + n * n + n * m + N * n * n + N * n + Rsz + rsz + num_terms + 4 * kMaxClosestItems +
         kClosestStride * kMaxClosestQueries;
}

// Executed by ONE wavefront (lane t of 64) with its own LDS scratch `sm`, so several waves of a
// workgroup can take different time steps of the same instance concurrently.
//
// The step comes in three pieces so that a caller looping over steps can request the NEXT step's
// argument before it issues THIS step's stores: vmcnt retires in order, so a load queued behind 8 KB of
// stores would not be usable until they have all drained.
template <typename T>
struct LinquadCarry {
  T ms1, ms2, ctot;  // merit pieces and PlayerCost::Evaluate of this lane's player (lanes < N)
};

// Element t of the step's [x | u] row (0 for lanes beyond n + m).
template <typename T>
__device__ __forceinline__ T linquad_load_arg(const QuadArgs<T>& a, int k, int n, int m, int t) {
  T argv = T(0);
  if (t < n)
    argv = a.xs[size_t(k) * n + t];
  else if (t < n + m)
    argv = a.us[size_t(k) * m + (t - n)];
  return argv;
}

template <typename T, int CN = 0, int CM = 0, int CNP = 0>
__device__ __forceinline__ void linquad_compute(const DevProblem& p, const QuadTables<T>& tb, const QuadArgs<T>& a,
                                                int k, T* sm, int t, T argv, LinquadCarry<T>& carry) {
  const int n = CN > 0 ? CN : p.n, m = CM > 0 ? CM : p.m, N = CNP > 0 ? CNP : p.N;
  constexpr int NT = 64;
  const PairTable& pt = p.pairs;
  T* sx = sm;  // [x | u] argument image
  T* sA = sx + n + m;
  T* sB = sA + n * n;
  T* sQ = sB + n * m;  // [Q | l | R | r] tile image, same element order as the global arrays
  T* sl = sQ + N * n * n;
  T* sR = sl + N * n;
  T* sr = sR + pt.Rsz;
  const bool do_quad = a.Q != nullptr || a.merit_part != nullptr;
  long long qc0 = (kProfile && a.phacc) ? clock64() : 0, qc1;
#define ILQG_QPH(i) do { if (kProfile && a.phacc) { __builtin_amdgcn_sched_barrier(0); qc1 = clock64(); __builtin_amdgcn_sched_barrier(0); a.phacc[i] += qc1 - qc0; qc0 = qc1; } } while (0)
  // ---- load the argument, initialise the tiles ----
  // The (x, u) row was requested by the caller (argv); the tile image is cleared while that load is in
  // flight (16-byte LDS writes where the image is 16-byte aligned and even-sized).
  // PlayerCost::Quadraticize vs QuadraticizeControlCosts (src/ilq_solver.cpp:483-487)
  auto is_full = [&](int i) {
    return tb.lc[LC_STRUCT + i] == ILQG_SUM || (a.t_extreme ? a.t_extreme[i] == k : k == 0);
  };
  {
    const int z0 = a.A ? 0 : n * n + n * m;  // clear [A | B] only when linearising
    const int z1 = do_quad ? n * n + n * m + N * n * n + N * n + pt.Rsz + pt.rsz : n * n + n * m;
    T* zb = sA;
    if (((n + m) & 1) == 0 && (z0 & 1) == 0 && (z1 & 1) == 0) {
      typedef T pair2 __attribute__((ext_vector_type(2)));
      const pair2 zz = {T(0), T(0)};
      for (int e = z0 + 2 * t; e < z1; e += 2 * NT) *reinterpret_cast<pair2*>(zb + e) = zz;
    } else {
      for (int e = z0 + t; e < z1; e += NT) zb[e] = T(0);
    }
  }
  if (t < n + m) sx[t] = argv;
  lds_sync(NT <= 64);
  if (t < n) {
    if (a.A) sA[t * (n + 1)] = T(1);  // LinearDynamicsApproximation starts from (I, 0)
    if (do_quad)
      for (int i = 0; i < N; i++)
        sQ[i * n * n + t * (n + 1)] = T(__int_as_float(tb.lc[LC_SREG + i]));  // sigma_x I (player_cost.cpp:196)
  }
  if (do_quad && t < pt.npairs) {
    // sigma_u I on every control block the reference would have created (player_cost.cpp:70-74)
    const int i = tb.lc[LC_PI + t];
    if (is_full(i) || tb.lc[LC_FROMCOST + t]) {
      const int mj = tb.lc[LC_UDIM + tb.lc[LC_PJ + t]];
      const int ro = tb.lc[LC_ROFF + t];
      const T creg = T(__int_as_float(tb.lc[LC_CREG + i]));
      for (int d = 0; d < mj; d++) sR[ro + d + mj * d] = creg;
    }
  }
  lds_sync(NT <= 64);
  ILQG_QPH(0);
  if (a.A) {
    // lanes [0, N): heading of subsystem t; lanes [N, 2N): steering angle of subsystem t - N.  One
    // sincos for the whole wave, then the steering pair hops N lanes down.
    const int sub = t < N ? t : (t < 2 * N ? t - N : 0);
    const int xo = tb.lc[LC_XOFF + sub];
    T sn, cs;
    t_sincos(sx[xo + (t < N ? 2 : 3)], &sn, &cs);
    const T sphi = shfl(sn, (t + N) & 63), cphi = shfl(cs, (t + N) & 63);
    if (t < N) {
      const int uo = tb.lc[LC_UOFF + t];
      const int kind = tb.lc[LC_KIND + t];
      T aux0 = T(0), aux1 = T(0);
      if (kind == ILQG_DYN_AIR_3D_EVADER) {  // its own turn rate and the pursuer's speed (the next row's parameter)
        aux0 = sx[n + uo];
        aux1 = T(__int_as_float(tb.lc[LC_PARAM + t + 1]));
      }
      sub_linearize_trig<T>(kind, T(__int_as_float(tb.lc[LC_PARAM + t])), p.dt, sx + xo, sn, cs, sphi, cphi,
                            sA + xo + n * xo, sB + xo + n * uo, n, aux0, aux1);
    }
  }
  // ---- one lane per cost term: value + derivative pattern (the expensive part, in parallel) ----
  const double tt = double(k) * p.dt;
  const int tidx = int(static_cast<size_t>((tt - a.t_init) / p.dt));  // relative_time_tracker.h:69-72
  T* svals = sr + pt.rsz;  // [num_terms] term values for TotalCosts
  // ---- shared Polyline2::ClosestPoint searches: one lane per (query, segment), one lane per query ----
  T* sitem = svals + p.num_terms;
  T* sclo = sitem + 4 * kMaxClosestItems;
  const bool shared_closest = p.num_cq > 0;
  if (shared_closest) {
    closest_items<T>(tb, p.num_cq_items, sx, sitem, t);
    lds_sync(NT <= 64);
    closest_select<T>(tb, p.num_cq, sitem, sclo, t);
    lds_sync(NT <= 64);
  }
  ILQG_QPH(1);
  for (int base = 0; base < p.num_terms; base += NT) {
    const int ti = base + t;
    DevTerm c;
    TermOut<T> o;
    o.pattern = PAT_NONE;
    o.value = T(0);
    bool live = false;
    if (ti < p.num_terms) {
      c = tb.terms[ti];
      live = c.role != ILQG_ROLE_CHILD && k >= c.k_start;  // FinalTimeCost: nothing before its threshold
      if (live) {
        const bool is_cost = c.role == ILQG_ROLE_STATE_COST || c.role == ILQG_ROLE_CONTROL_COST;
        const bool deriv = do_quad && (is_full(c.player) || c.role == ILQG_ROLE_CONTROL_COST);
        if (deriv || (a.cost_part && is_cost)) {
          const T lambda = (c.slot >= 0 && a.lambdas) ? a.lambdas[c.slot * p.T + tidx] : T(0);
          term_compute<T>(tb, c, sx + c.arg_off, lambda, a.mu, &o, shared_closest ? sclo : nullptr);
          if (!deriv) o.pattern = PAT_NONE;
        }
      }
      if (a.cost_part) svals[ti] = o.value;
    }
    ILQG_QPH(2);
    // ---- scatter in rounds: within a round no two terms touch the same entry ----
    if (do_quad) {
      for (int r = 0; r < p.num_rounds; r++) {
        lds_sync(NT <= 64);
        if (live && c.round == r && o.pattern != PAT_NONE)
          term_scatter<T>(o, sx + c.arg_off, c.arg_dim, sQ + c.tile_h, c.ld, sQ + c.tile_g);
      }
    }
  }
  lds_sync(NT <= 64);
  ILQG_QPH(3);
  // Values first, stores last: a register that feeds a global store cannot be rewritten until the store
  // has drained (the compiler waits vmcnt(0) for it), so every store of the step is issued in one
  // burst at the very end, from registers nothing else needs.
  T ms1 = T(0), ms2 = T(0), ctot = T(0);
  if (t < N) {
    const int i = t;
    if (a.merit_part) {  // pieces of ILQSolver::MeritFunction (:419-430)
      const int rg = tb.lc[LC_RGOFF + tb.lc[LC_PII + i]], ud = tb.lc[LC_UDIM + i];
      T s1 = T(0), s2 = T(0);
      for (int d = 0; d < ud; d++) s1 += sr[rg + d] * sr[rg + d];
      if constexpr (CN > 0) {
        T lv[CN];  // loads first, then the sequential sum (same order as the reference)
#pragma unroll
        for (int d = 0; d < CN; d++) lv[d] = sl[i * CN + d];
#pragma unroll
        for (int d = 0; d < CN; d++) s2 += lv[d] * lv[d];
      } else {
        for (int d = 0; d < n; d++) s2 += sl[i * n + d] * sl[i * n + d];
      }
      ms1 = s1;
      ms2 = s2;
    }
    if (a.cost_part) {  // PlayerCost::Evaluate, src/player_cost.cpp:128-144 (state costs, then control costs)
      // Same left-to-right sum as the reference; indices and values are fetched eight at a time so the
      // LDS round trips overlap instead of chaining (index -> value -> add).
      T total = T(0);
      const int* ord = tb.order + i * p.cost_order_stride;
      const int cnt = ord[0];
      for (int q0 = 0; q0 < cnt; q0 += 8) {
        int idx[8];
        T val[8];
#pragma unroll
        for (int u = 0; u < 8; u++) idx[u] = ord[1 + ((q0 + u < cnt) ? q0 + u : cnt - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) val[u] = svals[idx[u]];
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (q0 + u < cnt) total += val[u];
      }
      ctot = total;
    }
  }
  carry.ms1 = ms1;
  carry.ms2 = ms2;
  carry.ctot = ctot;
  ILQG_QPH(4);
#undef ILQG_QPH
}

template <typename T, int CN = 0, int CM = 0, int CNP = 0>
__device__ __forceinline__ void linquad_store(const DevProblem& p, const QuadArgs<T>& a, int k, T* sm, int t,
                                              const LinquadCarry<T>& carry) {
  const int n = CN > 0 ? CN : p.n, m = CM > 0 ? CM : p.m, N = CNP > 0 ? CNP : p.N;
  constexpr int NT = 64;
  const PairTable& pt = p.pairs;
  T* sx = sm;
  T* sA = sx + n + m;
  T* sB = sA + n * n;
  T* sQ = sB + n * m;
  T* sl = sQ + N * n * n;
  T* sR = sl + N * n;
  T* sr = sR + pt.Rsz;
  const T ms1 = carry.ms1, ms2 = carry.ms2, ctot = carry.ctot;
  long long qc0 = (kProfile && a.phacc) ? clock64() : 0, qc1;
#define ILQG_QPH(i) do { if (kProfile && a.phacc) { __builtin_amdgcn_sched_barrier(0); qc1 = clock64(); __builtin_amdgcn_sched_barrier(0); a.phacc[i] += qc1 - qc0; qc0 = qc1; } } while (0)
  // ---- coalesced write-out: the whole image is read into registers, then stored in one burst ----
  typedef T pair2 __attribute__((ext_vector_type(2)));
  bool burst = false;
  if constexpr (CN > 0 && CM > 0 && CNP > 0) {
    constexpr int cA = CN * CN, cB = CN * CM, cQ = CNP * CN * CN, cl = CNP * CN;
    constexpr bool even = (cA % 2 == 0) && (cB % 2 == 0) && (cQ % 2 == 0) && (cl % 2 == 0) && ((CN + CM) % 2 == 0);
    constexpr int UA = (cA / 2 + NT - 1) / NT, UB = (cB / 2 + NT - 1) / NT, UQ = (cQ / 2 + NT - 1) / NT,
                  UL = (cl / 2 + NT - 1) / NT;
    if (even && UA + UB + UQ + UL <= 12 && (pt.Rsz & 1) == 0 && (pt.rsz & 1) == 0 && pt.Rsz <= 2 * NT &&
        pt.rsz <= 2 * NT && a.A != nullptr && a.Q != nullptr) {
      burst = true;
      pair2 rA[UA], rB[UB], rQ[UQ], rl[UL], rR = {T(0), T(0)}, rr = {T(0), T(0)};
#pragma unroll
      for (int u = 0; u < UA; u++)
        if (2 * (t + NT * u) < cA) rA[u] = *reinterpret_cast<const pair2*>(sA + 2 * (t + NT * u));
#pragma unroll
      for (int u = 0; u < UB; u++)
        if (2 * (t + NT * u) < cB) rB[u] = *reinterpret_cast<const pair2*>(sB + 2 * (t + NT * u));
#pragma unroll
      for (int u = 0; u < UQ; u++)
        if (2 * (t + NT * u) < cQ) rQ[u] = *reinterpret_cast<const pair2*>(sQ + 2 * (t + NT * u));
#pragma unroll
      for (int u = 0; u < UL; u++)
        if (2 * (t + NT * u) < cl) rl[u] = *reinterpret_cast<const pair2*>(sl + 2 * (t + NT * u));
      if (2 * t < pt.Rsz) rR = *reinterpret_cast<const pair2*>(sR + 2 * t);
      if (2 * t < pt.rsz) rr = *reinterpret_cast<const pair2*>(sr + 2 * t);
      T* gA = a.A + size_t(k) * cA;
      T* gB = a.Bm + size_t(k) * cB;
      T* gQ = a.Q + size_t(k) * cQ;
      T* gl = a.l + size_t(k) * cl;
      T* gR = a.R + size_t(k) * pt.Rsz;
      T* gr = a.r + size_t(k) * pt.rsz;
#pragma unroll
      for (int u = 0; u < UA; u++)
        if (2 * (t + NT * u) < cA) *reinterpret_cast<pair2*>(gA + 2 * (t + NT * u)) = rA[u];
#pragma unroll
      for (int u = 0; u < UB; u++)
        if (2 * (t + NT * u) < cB) *reinterpret_cast<pair2*>(gB + 2 * (t + NT * u)) = rB[u];
#pragma unroll
      for (int u = 0; u < UQ; u++)
        if (2 * (t + NT * u) < cQ) *reinterpret_cast<pair2*>(gQ + 2 * (t + NT * u)) = rQ[u];
#pragma unroll
      for (int u = 0; u < UL; u++)
        if (2 * (t + NT * u) < cl) *reinterpret_cast<pair2*>(gl + 2 * (t + NT * u)) = rl[u];
      if (2 * t < pt.Rsz) *reinterpret_cast<pair2*>(gR + 2 * t) = rR;
      if (2 * t < pt.rsz) *reinterpret_cast<pair2*>(gr + 2 * t) = rr;
    }
  }
  if (!burst) {
    auto copy_out = [&](T* dst, const T* src, int count) {
      for (int e = t; e < count; e += NT) dst[e] = src[e];
    };
    if (a.A) {
      copy_out(a.A + size_t(k) * n * n, sA, n * n);
      copy_out(a.Bm + size_t(k) * n * m, sB, n * m);
    }
    if (a.Q) {
      copy_out(a.Q + size_t(k) * N * n * n, sQ, N * n * n);
      copy_out(a.l + size_t(k) * N * n, sl, N * n);
      copy_out(a.R + size_t(k) * pt.Rsz, sR, pt.Rsz);
      copy_out(a.r + size_t(k) * pt.rsz, sr, pt.rsz);
    }
  }
  if (t < N) {
    if (a.merit_part) {
      a.merit_part[(size_t(k) * N + t) * 2 + 0] = ms1;
      a.merit_part[(size_t(k) * N + t) * 2 + 1] = ms2;
    }
    if (a.cost_part) a.cost_part[size_t(k) * N + t] = ctot;
  }
  lds_sync(NT <= 64);
  ILQG_QPH(5);
#undef ILQG_QPH
}

template <typename T, int CN = 0, int CM = 0, int CNP = 0>
__device__ __forceinline__ void linquad_step(const DevProblem& p, const QuadTables<T>& tb, const QuadArgs<T>& a, int k,
                                             T* sm, int t) {
  const int n = CN > 0 ? CN : p.n, m = CM > 0 ? CM : p.m;
  LinquadCarry<T> carry;
  linquad_compute<T, CN, CM, CNP>(p, tb, a, k, sm, t, linquad_load_arg<T>(a, k, n, m, t), carry);
  linquad_store<T, CN, CM, CNP>(p, a, k, sm, t, carry);
}

// Sequential left-to-right sum of `count` LDS values, eight loads in flight at a time (the adds keep
// the reference's order; only the LDS latency overlaps).
template <typename T, typename F>
__device__ __forceinline__ void lds_ordered_visit(const T* v, int count, F&& visit) {
  int e = 0;
  for (; e + 8 <= count; e += 8) {
    T x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) x[u] = v[e + u];
#pragma unroll
    for (int u = 0; u < 8; u++) visit(e + u, x[u]);
  }
  for (; e < count; e++) visit(e, v[e]);
}

// ILQSolver::MeritFunction's reduction (:408-434): 0.5 * sum_k sum_i (|r_ii|^2 + [k>0]|l_i|^2),
// accumulated in the reference's order by one lane.  Returns the value on every thread.
// `sm` is LDS scratch of `sm_elems` elements: when the partials fit they are pulled in by the whole
// workgroup first (one global round trip instead of one per term of the sum).
template <typename T>
__device__ __forceinline__ T merit_reduce(const DevProblem& p, const T* merit_part, T* sm, int sm_elems = 0) {
  __syncthreads();  // partials were written to global memory by other lanes
  const int count = p.T * p.N * 2;
  if (count + 1 <= sm_elems) {
    for (int e = threadIdx.x; e < count; e += blockDim.x) sm[e] = merit_part[e];
    __syncthreads();
    if (threadIdx.x == 0) {
      T merit = T(0);
      const int skip = p.N * 2;  // the |l_i|^2 terms of k = 0 do not enter (:421)
      lds_ordered_visit<T>(sm, count, [&](int e, T x) {
        if ((e & 1) == 0 || e >= skip) merit += x;
      });
      sm[count] = T(0.5) * merit;
    }
    __syncthreads();
    const T v = sm[count];
    __syncthreads();
    return v;
  }
  if (threadIdx.x == 0) {
    T merit = T(0);
    for (int k = 0; k < p.T; k++)
      for (int i = 0; i < p.N; i++) {
        merit += merit_part[(size_t(k) * p.N + i) * 2 + 0];
        if (k > 0) merit += merit_part[(size_t(k) * p.N + i) * 2 + 1];
      }
    sm[0] = T(0.5) * merit;
  }
  __syncthreads();
  const T v = sm[0];
  __syncthreads();
  return v;
}

// ILQSolver::TotalCosts reduction (:220-257): sum / max / min over time per player, and
// the time of the extreme cost (first strict improvement wins, as the reference's `>` / `<`).
template <typename T>
__device__ __forceinline__ void costs_reduce(const DevProblem& p, const T* cost_part, T* costs_out, int* t_extreme,
                                             T* sm = nullptr, int sm_elems = 0) {
  __syncthreads();  // partials were written to global memory by other lanes
  const int i = threadIdx.x;
  const int count = p.T * p.N;
  const bool staged = sm != nullptr && count <= sm_elems;
  if (staged) {
    for (int e = threadIdx.x; e < count; e += blockDim.x) sm[e] = cost_part[e];
    __syncthreads();
  }
  if (i < p.N) {
    const int st = p.structure[i];
    T c = st == ILQG_SUM ? T(0) : (st == ILQG_MAX ? -dinf<T>() : dinf<T>());
    int te = t_extreme ? t_extreme[i] : 0;
    auto visit = [&](int k, T v) {
      if (st == ILQG_SUM)
        c += v;
      else if (st == ILQG_MAX && v > c) {
        c = v;
        te = k;
      } else if (st == ILQG_MIN && v < c) {
        c = v;
        te = k;
      }
    };
    if (staged) {
      int k = 0;
      for (; k + 8 <= p.T; k += 8) {
        T x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = sm[(k + u) * p.N + i];
#pragma unroll
        for (int u = 0; u < 8; u++) visit(k + u, x[u]);
      }
      for (; k < p.T; k++) visit(k, sm[k * p.N + i]);
    } else {
      for (int k = 0; k < p.T; k++) visit(k, cost_part[size_t(k) * p.N + i]);
    }
    costs_out[i] = c;
    if (t_extreme) t_extreme[i] = te;
  }
  __syncthreads();
}

}  // namespace ilqg

#include "ilqg_rows.hpp"
