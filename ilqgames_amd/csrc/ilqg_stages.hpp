// ilqg_stages.hpp — rollout, linearise+quadraticise and reduction stages for ONE game
// instance per workgroup (gfx950).  Runtime dimensions (these stages are bound by the
// bytes they write and by the sequential rollout chain, not by unrolled math).
#pragma once

#include "ilqg_common.hpp"
#include "ilqg_models.hpp"

#ifndef ILQG_ROLLOUT_PAIRS
#define ILQG_ROLLOUT_PAIRS 1  // two rollouts per wavefront in the split rollout kernels (0: one, for A/B measurements)
#endif

namespace ilqg {

// ---------------------------------------------------------------------------
// Rollout — ILQSolver::CurrentOperatingPoint (src/ilq_solver.cpp:174-206).
// Eight lanes per subsystem run the RK4 (2 sub-steps) with one stage each, so the 24 serial
// sin/cos/tan of a step collapse to two (sub_integrate_stages); lanes rho < m
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
// GEN: the game may contain the models the stage-parallel integrator has no closed forms for (Unicycle5D, Car7D,
// DelayedDubinsCar; blocks of up to 8 states): every row then takes the plain RK4, one copy per lane of its group.
// The instantiations that may hold such games, by dimension:
__host__ __device__ constexpr bool dims_use_plain_rk4(int nx, int np, int mu) {
  return (nx == 17 && np == 3 && mu == 2) || (nx == 8 && np == 2 && mu == 1);
}
template <typename T, int CN = 0, int CM = 0, bool DIST = false, bool DUB = false, bool AIR = false, bool PM = false,
          bool GEN = false>
__device__ __forceinline__ void rollout_instance(const DevProblem& p, const RolloutArgs<T>& a, T* sm, int t,
                                                 int* ready = nullptr, long long* phacc = nullptr,
                                                 long long* tl = nullptr, int tl_b = 0) {
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
  constexpr int XS = GEN ? kSubStatesMax : 6;
  T xj[XS];
  const int grp = t >> 3, q = t & 7;
  const bool integ = grp < N && t < 64;
  int kind = ILQG_DYN_UNICYCLE_4D, xo = 0, uo = 0, xd = 0;
  T Lp = T(1);
#pragma unroll
  for (int e = 0; e < XS; e++) xj[e] = T(0);
  if (integ) {
    kind = p.sub_kind[grp];
    xo = p.xoff[grp];
    uo = p.uoff[grp];
    xd = p.xoff[grp + 1] - xo;
    Lp = T(p.sub_param[grp]);
#pragma unroll
    for (int e = 0; e < XS; e++) xj[e] = (e < xd) ? a.x0[xo + e] : T(0);
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
    if (kTimeline && (k & 31) == 0) tl_stamp(tl, tl_b, 20 + (k >> 5), t == 0);
    if (ready) progress_publish(ready, k);
    const T* sP = stg + (k & 1) * WP;  // [m*n] gains of this step
    const T* sal = sP + m * n;         // [m]
    const T* sur = sal + m;            // [m] u_ref
    const T* sxr = sur + m;            // [n] x_ref
    {
      // every lane of a group holds the group's state: lane q publishes entry q (one LDS round trip for the row)
      T mine = xj[0];
#pragma unroll
      for (int e = 1; e < XS; e++) mine = (q == e) ? xj[e] : mine;
      if (integ && q < xd) {
        sdx[xo + q] = mine - sxr[xo + q];
        a.xs[size_t(k) * n + xo + q] = mine;
      }
    }
    lds_sync(NT <= 64);
    ILQG_RPH(0);
    // (Measured and dropped, round 5: four lanes per control, each a quarter of the dot product, DPP quad sums —
    // -0.3 % on the headline: the chain is not what this phase costs.)
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
            if (c0 + c < CN) s = t_fma(pr[c], dv[c], s);
        }
      } else {
        for (int c = 0; c < n; c++) s = t_fma(sP[t + m * c], sdx[c], s);
      }
      const T u = t_fma(-a.alpha_scale, sal[t], sur[t] - s);
      su[t] = u;
      a.us[size_t(k) * m + t] = u;
    }
    lds_sync(NT <= 64);
    ILQG_RPH(1);
    // the next block's DMA (into the block step k - 1 read) is requested here, behind the controls — it still has most of
    // a step to land, and its dozen address / M0 instructions are off the chain publish -> u -> integrate (round 5:
    // +0.7 % on the headline; it used to follow the publication barrier)
    if (k + 1 < Tn) issue(k + 1, (k + 1) & 1);
    if (t < 64 && k + 1 < Tn) {  // whole first wave: the exchanges inside need every group lane live
      const T u0 = integ ? su[uo] : T(0), u1 = integ ? su[uo + 1] : T(0);
      if constexpr (GEN) {
        if (integ) sub_integrate8<T>(kind, Lp, p.dt, xj, u0, u1);
      } else if constexpr (AIR) {
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
// Two rollouts per wavefront.  rollout_instance uses 8 lanes per subsystem and m lanes for the controls — 24 of 64 for
// three players — and the split rollout kernel at four waves per SIMD is bound by the instructions it issues (~520 per
// step, ~300 of them fp64), not by their latency: the second half of the wave takes a second trajectory through the same
// instruction stream.  Lanes 0-31 integrate `a0`, lanes 32-63 `a1`: the same lane roles within each half (subsystem
// g = (t & 31) / 8, stage q = t & 7, control lane (t & 31) < m), each half with its own [x | dx | u] and staged
// [P | alpha | u_ref | x_ref] blocks in LDS, the same arithmetic per lane — a trajectory comes out bit for bit as
// rollout_instance would produce it.  `act0` / `act1`: whether the half's trajectory is wanted (an idle half repeats
// the other's inputs and stores nothing).  N <= 4, m <= 32; the integrators with their own lane layouts (Air3D, point
// masses, the plain-RK4 models, the disturbed unicycle) stay on rollout_instance.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr bool rollout_pairs(int nx, int np, int mu) {
  return nx > 0 && np <= 4 && np * mu <= 32 && !(nx == 4 && np == 2) && !(nx == 3 && np == 2 && mu == 1) &&
         !(nx == 4 * np && mu == 2 && np <= 2) && !dims_use_plain_rk4(nx, np, mu) && ILQG_ROLLOUT_PAIRS;
}
__host__ __device__ inline int rollout_pair_lds_elems(int n, int m) {
  return 2 * ((2 * n + m + 3) & ~3) + 4 * rollout_stage_elems(n, m) + kRolloutGatherElems;
}

// one half's LDS-DMA: lanes 32 H .. 32 H + 31 move `nbytes` from g to l in 4-byte pieces (the hardware writes lane L's
// piece at base + 4 L, so the upper half aims 128 bytes low)
template <int H>
__device__ __forceinline__ void dma_g2l_half(const void* g_, void* l, int nbytes, int tl) {
  const void* g = uniform_ptr(g_);
  for (int off = 0; off < nbytes; off += 128) {
    const int my = off + tl * 4;
    if (my < nbytes)
      __builtin_amdgcn_global_load_lds((glb_vptr)((const char*)g + unsigned(my)), (lds_vptr)((char*)l + off - 128 * H), 4, 0, 0);
  }
}

template <typename T, int CN, int CM, bool DUB>
__device__ __forceinline__ void rollout_pair(const DevProblem& p, const RolloutArgs<T>& a0, const RolloutArgs<T>& a1,
                                             bool act0, bool act1, T* sm, int t) {
  static_assert(CN > 0 && CM > 0 && CM <= 32, "compile-time dimensions");
  constexpr int n = CN, m = CM;
  const int N = p.N, Tn = p.T;
  const int h = t >> 5, tl = t & 31;
  constexpr int S0 = (2 * n + m + 3) & ~3;
  const int WP = rollout_stage_elems(n, m);
  T* const sx = sm + h * S0;  // this half's [x | dx | u]
  T* const sdx = sx + n;
  T* const su = sdx + n;
  T* const stg = sm + 2 * S0 + h * 2 * WP;  // this half's two staged blocks
  T* const gth = sm + 2 * S0 + 4 * WP;      // the wave's exchange scratch (indexed by lane)
  constexpr int S = int(sizeof(T));
  const bool act = h ? act1 : act0;
  auto issue = [&](int k, int buf) {
    if (h == 0) {
      T* d = sm + 2 * S0 + buf * WP;
      dma_g2l_half<0>(a0.P + size_t(k) * m * n, d, m * n * S, tl);
      dma_g2l_half<0>(a0.alpha + size_t(k) * m, d + m * n, m * S, tl);
      dma_g2l_half<0>(a0.us_ref + size_t(k) * m, d + m * n + m, m * S, tl);
      dma_g2l_half<0>(a0.xs_ref + size_t(k) * n, d + m * n + 2 * m, n * S, tl);
      // The two arms must stay two instruction streams: the DMA's LDS base travels in M0 and has to be uniform, and an
      // optimiser that sinks the arms' common tail into one block hands it a per-lane select of the two bases (seen
      // in the ISA: v_readfirstlane of a VGPR base — the upper half's rows then land in the lower half's block).
      // Distinct markers at the end of each arm leave nothing identical to sink.
      asm volatile("; rollout_pair: lower half staged" ::: "memory");
    } else {
      T* d = sm + 2 * S0 + 2 * WP + buf * WP;
      dma_g2l_half<1>(a1.P + size_t(k) * m * n, d, m * n * S, tl);
      dma_g2l_half<1>(a1.alpha + size_t(k) * m, d + m * n, m * S, tl);
      dma_g2l_half<1>(a1.us_ref + size_t(k) * m, d + m * n + m, m * S, tl);
      dma_g2l_half<1>(a1.xs_ref + size_t(k) * n, d + m * n + 2 * m, n * S, tl);
      asm volatile("; rollout_pair: upper half staged" ::: "memory");
    }
  };
  constexpr int XS = 6;
  T xj[XS];
  const int grp = tl >> 3, q = t & 7;
  const bool integ = grp < N;
  int kind = ILQG_DYN_UNICYCLE_4D, xo = 0, uo = 0, xd = 0;
  T Lp = T(1);
  const T* const x0 = h ? a1.x0 : a0.x0;
  T* const xs_out = h ? a1.xs : a0.xs;
  T* const us_out = h ? a1.us : a0.us;
  const T alpha_scale = h ? a1.alpha_scale : a0.alpha_scale;
#pragma unroll
  for (int e = 0; e < XS; e++) xj[e] = T(0);
  if (integ) {
    kind = p.sub_kind[grp];
    xo = p.xoff[grp];
    uo = p.uoff[grp];
    xd = p.xoff[grp + 1] - xo;
    Lp = T(p.sub_param[grp]);
#pragma unroll
    for (int e = 0; e < XS; e++) xj[e] = (e < xd) ? x0[xo + e] : T(0);
  }
  const bool any_car = __any(integ && (kind == ILQG_DYN_CAR_5D || kind == ILQG_DYN_CAR_6D));
  issue(0, 0);
#pragma unroll 1
  for (int k = 0; k < Tn; k++) {
    dma_wait();
    const T* sP = stg + (k & 1) * WP;
    const T* sal = sP + m * n;
    const T* sur = sal + m;
    const T* sxr = sur + m;
    {
      T mine = xj[0];
#pragma unroll
      for (int e = 1; e < XS; e++) mine = (q == e) ? xj[e] : mine;
      if (integ && q < xd) {
        sdx[xo + q] = mine - sxr[xo + q];
        if (act) xs_out[size_t(k) * n + xo + q] = mine;
      }
    }
    lds_sync(true);
    if (tl < m) {
      T s = T(0);
      constexpr int CH = 8;
#pragma unroll
      for (int c0 = 0; c0 < CN; c0 += CH) {
        T pr[CH], dv[CH];
#pragma unroll
        for (int c = 0; c < CH; c++)
          if (c0 + c < CN) {
            pr[c] = sP[tl + m * (c0 + c)];
            dv[c] = sdx[c0 + c];
          }
#pragma unroll
        for (int c = 0; c < CH; c++)
          if (c0 + c < CN) s = t_fma(pr[c], dv[c], s);
      }
      const T u = t_fma(-alpha_scale, sal[tl], sur[tl] - s);
      su[tl] = u;
      if (act) us_out[size_t(k) * m + tl] = u;
    }
    lds_sync(true);
    if (k + 1 < Tn) issue(k + 1, (k + 1) & 1);  // behind the controls, as in rollout_instance
    if (k + 1 < Tn) {  // the whole wave: the exchanges inside need every group lane live
      const T u0 = integ ? su[uo] : T(0), u1 = integ ? su[uo + 1] : T(0);
      sub_integrate_stages<T, false, DUB>(kind, Lp, p.dt, xj, u0, u1, q, t, gth, any_car, T(0), T(0),
                                          h ? 0xffffffff00000000ull : 0x00000000ffffffffull);
    }
  }
}

// ---------------------------------------------------------------------------
// Many rollouts of ONE instance per wavefront: a lane per (trajectory, subsystem).  The candidates of a speculative line
// search differ in one number — the step size that scales alpha — and share the instance's gains, references and
// starting point; a probing round integrates thousands of them (a failing search walks through all 100 step sizes), and
// at two trajectories per wavefront it was bound by the instructions it issues (~500 per step for the pair).  Here lane
// c N + i carries subsystem i of trajectory c (C = 64 / N trajectories: 21 for three players) through the same
// arithmetic: its state in registers, its rows of  u = (u_ref - P dx) - step alpha  as MU chains in the reference's
// order over the trajectory's dx (exchanged through LDS), and the eight RK4 stages one after the other
// (sub_integrate_stages_seq, ilqg_models.hpp) — ~1/8 of the instructions per trajectory, bit for bit the trajectory
// rollout_instance / rollout_pair produce.  `steps[c]`, `xs[c]`, `us[c]` per lane; act: the lane's trajectory is wanted.
// The stage-parallel integrator's family only (cars, unicycles, Dubins cars: where rollout_pairs() holds).
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int rollout_lanes_per_wave(int np) { return 64 / np; }
__host__ __device__ inline int rollout_lanes_dx_stride(int n) { return n | 1; }  // odd: the candidates' rows on distinct banks
__host__ __device__ inline int rollout_lanes_lds_elems(int n, int m, int np) {
  return 2 * rollout_stage_elems(n, m) + ((rollout_lanes_per_wave(np) * rollout_lanes_dx_stride(n) + 3) & ~3);
}

template <typename T, int CN, int CM, int NP, bool DUB>
__device__ __forceinline__ void rollout_lanes(const DevProblem& p, const RolloutArgs<T>& a, T step, T* xs_out, T* us_out,
                                              bool act, T* sm, int t) {
  static_assert(CN > 0 && CM > 0 && NP > 0, "compile-time dimensions");
  constexpr int n = CN, m = CM, MU = CM / NP, C = rollout_lanes_per_wave(NP);
  const int Tn = p.T;
  const int WP = rollout_stage_elems(n, m);
  T* const stg = sm;  // two staged blocks [P | alpha | u_ref | x_ref]
  const int DS = rollout_lanes_dx_stride(n);
  const int c = t / NP, i = t - c * NP;
  const bool live = c < C;                   // (64 - C NP lanes idle)
  T* const sdx = sm + 2 * WP + (live ? c : 0) * DS;  // this trajectory's dx
  constexpr int S = int(sizeof(T));
  auto issue = [&](int k, int buf) {
    T* d = stg + buf * WP;
    dma_g2l<64, false>(a.P + size_t(k) * m * n, d, m * n * S, t);
    dma_g2l<64, false>(a.alpha + size_t(k) * m, d + m * n, m * S, t);
    dma_g2l<64, false>(a.us_ref + size_t(k) * m, d + m * n + m, m * S, t);
    dma_g2l<64, false>(a.xs_ref + size_t(k) * n, d + m * n + 2 * m, n * S, t);
  };
  constexpr int XS = 6;
  T xj[XS];
  const int kind = p.sub_kind[i];
  const int xo = p.xoff[i], uo = p.uoff[i], xd = p.xoff[i + 1] - xo, ud = p.udim[i];
  const T Lp = T(p.sub_param[i]);
#pragma unroll
  for (int e = 0; e < XS; e++) xj[e] = (e < xd) ? a.x0[xo + e] : T(0);
  const bool store = act && live;
  issue(0, 0);
#pragma unroll 1
  for (int k = 0; k < Tn; k++) {
    dma_wait();
    const T* sP = stg + (k & 1) * WP;
    const T* sal = sP + m * n;
    const T* sur = sal + m;
    const T* sxr = sur + m;
#pragma unroll
    for (int e = 0; e < XS; e++)
      if (e < xd) {
        if (live) sdx[xo + e] = xj[e] - sxr[xo + e];
        if (store) xs_out[size_t(k) * n + xo + e] = xj[e];
      }
    lds_sync(true);
    T u[MU];
    {
      T sacc[MU];
#pragma unroll
      for (int e = 0; e < MU; e++) sacc[e] = T(0);
      constexpr int CH = 8;
#pragma unroll
      for (int c0 = 0; c0 < CN; c0 += CH) {
        T dv[CH], pr[MU][CH];
#pragma unroll
        for (int cc = 0; cc < CH; cc++)
          if (c0 + cc < CN) {
            dv[cc] = sdx[c0 + cc];
#pragma unroll
            for (int e = 0; e < MU; e++) pr[e][cc] = sP[uo + (e < ud ? e : 0) + m * (c0 + cc)];
          }
#pragma unroll
        for (int cc = 0; cc < CH; cc++)
          if (c0 + cc < CN) {
#pragma unroll
            for (int e = 0; e < MU; e++) sacc[e] = t_fma(pr[e][cc], dv[cc], sacc[e]);
          }
      }
#pragma unroll
      for (int e = 0; e < MU; e++) {
        const int row = uo + (e < ud ? e : 0);
        u[e] = t_fma(-step, sal[row], sur[row] - sacc[e]);
        if (store && e < ud) us_out[size_t(k) * m + row] = u[e];
      }
    }
    lds_sync(true);  // every lane has read the staged block and the dx rows: the next request may overwrite block k - 1, the next step's dx these
    if (k + 1 < Tn) {
      issue(k + 1, (k + 1) & 1);
      sub_integrate_stages_seq<T, DUB>(kind, Lp, p.dt, xj, u[0], MU > 1 ? u[MU > 1 ? 1 : 0] : T(0));
    }
  }
}

// The rollout with run-time dimensions, the integrator picked from the problem's models at run time (what the
// instantiated solves pick at compile time from their dimensions): the stand-alone rollout entry point and the
// run-time-dimensioned solve path.
template <typename T>
__device__ __forceinline__ void rollout_instance_rt(const DevProblem& p, const RolloutArgs<T>& a, T* sm, int t) {
  bool dubins = false, plain = false;
  for (int i = 0; i < p.N; i++) {
    dubins = dubins || p.sub_kind[i] == ILQG_DYN_DUBINS_CAR;
    plain = plain || is_plain_rk4_kind(p.sub_kind[i]);
  }
  if (plain)
    rollout_instance<T, 0, 0, false, false, false, false, true>(p, a, sm, t);
  else if (p.sub_kind[0] == ILQG_DYN_AIR_3D_EVADER)
    rollout_instance<T, 0, 0, false, false, true>(p, a, sm, t);
  else if (p.sub_kind[0] == ILQG_DYN_UNICYCLE_4D_DISTURBED)
    rollout_instance<T, 0, 0, true>(p, a, sm, t);
  else if (dubins)
    rollout_instance<T, 0, 0, false, true>(p, a, sm, t);
  else if (p.sub_kind[0] == ILQG_DYN_POINT_MASS_2D)
    rollout_instance<T, 0, 0, false, false, false, true>(p, a, sm, t);
  else
    rollout_instance<T>(p, a, sm, t);
}

// ---------------------------------------------------------------------------
// Cost tables in LDS for the kernels that walk a player's cost list with a lane-varying index (strategy
// costs / Nash checks, the multiplier update of the exit path).  The linearise + quadraticise stage itself is
// ilqg_rows.hpp (one lane per time step, tables in scalar memory).
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
  __syncthreads();
  QuadTables<T> tb;
  tb.terms = reinterpret_cast<const DevTerm*>(terms_i);
  tb.segs = segs;
  tb.poly_off = poff;
  tb.order = order;
  tb.lc = lc;
  tb.tnom = problem_time_nominal<T>(p);
  tb.tnom_T = p.T;
  tb.dense = problem_dense<T>(p);
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
  T* compact = nullptr;  // [T][rp_compact_w] or nullptr: compact rows (ilqg_common.hpp) instead of the dense arrays —
  bool compact_lin = false, compact_quad = false;  // ... of the linearisation (A, Bm) / of the quadraticisation (Q, l, R, r)
  T* merit_part;       // [T][N][2] = (|r_ii|^2, |l_i|^2) or nullptr
  T* cost_part;        // [T][N] PlayerCost::Evaluate or nullptr
  long long* phacc = nullptr;  // optional phase profile accumulators (registers of the caller)
  long long* tl = nullptr;     // optional timeline stamps (ilqg_common.hpp, -DILQG_TIMELINE=1): slots 40.. of instance tl_b
  int tl_b = 0;
};

// Global -> LDS copy of `count` elements by the whole workgroup with a thread's loads issued together, eight at a time,
// ahead of its LDS stores: one exposed global round trip per eight elements of a thread instead of one per element
// (the staging loops of the reductions below were five and three dependent round trips: ~8 of the fused trial kernel's
// last 15 us, timeline of round 6).
template <typename T>
__device__ __forceinline__ void stage_to_lds(T* sm, const T* g, int count) {
  constexpr int UN = 8;
  const int nt = blockDim.x;
  for (int e0 = threadIdx.x; e0 < count; e0 += nt * UN) {
    T v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int e = e0 + u * nt;
      v[u] = g[e < count ? e : e0];
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int e = e0 + u * nt;
      if (e < count) sm[e] = v[u];
    }
  }
}

// Sequential left-to-right sum of `count` LDS values, eight loads in flight at a time (the adds keep
// the reference's order; only the LDS latency overlaps).
template <typename T, typename F>
__device__ __forceinline__ void lds_ordered_visit(const T* v, int count, F&& visit) {
  // software-pipelined: the loads of the next eight are in flight while this eight is added (the chain of dependent
  // adds is what the reduction costs; an LDS round trip per batch on top of it doubled that)
  constexpr int U = 8;
  int e = 0;
  if (count >= U) {
    T x[U];
#pragma unroll
    for (int u = 0; u < U; u++) x[u] = v[u];
    for (; e + 2 * U <= count; e += U) {
      T y[U];
#pragma unroll
      for (int u = 0; u < U; u++) y[u] = v[e + U + u];
#pragma unroll
      for (int u = 0; u < U; u++) visit(e + u, x[u]);
#pragma unroll
      for (int u = 0; u < U; u++) x[u] = y[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) visit(e + u, x[u]);
    e += U;
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
    stage_to_lds<T>(sm, merit_part, count);
    __syncthreads();
    if (threadIdx.x == 0) {
      T merit = T(0);
      const int skip = p.N * 2;  // the |l_i|^2 terms of k = 0 do not enter (:421)
      // (row 0 on its own, so that the chain over the other rows is one add per element and not an add and a select)
      for (int e = 0; e < skip && e < count; e += 2) merit += sm[e];
      lds_ordered_visit<T>(sm + skip, count - skip, [&](int, T x) { merit += x; });
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

// Both reductions of an accepted-or-not line-search trial in one go: the merit value (merit_reduce) on the first lane
// of the workgroup and, beside it on another wave (when there is one), the per-player cost totals of costs_reduce, held
// back in LDS until the caller knows whether the trial is accepted (costs_commit).  One staging of both partial arrays,
// one pair of workgroup barriers.  Returns false (nothing done) when the partials do not fit `sm`.
template <typename T>
__device__ __forceinline__ bool merit_costs_reduce(const DevProblem& p, const T* merit_part, const T* cost_part,
                                                   const int* t_extreme, T* sm, int sm_elems, T* merit_out) {
  const int cm = p.T * p.N * 2, cc = p.T * p.N;
  if (cm + cc + 2 * kMaxPlayers + 4 > sm_elems) return false;
  __syncthreads();  // partials were written to global memory by other lanes
  T* const smc = sm + cm;
  T* const res = smc + cc;  // [merit | cost_i ... | t_extreme_i (as T) ...]
  stage_to_lds<T>(sm, merit_part, cm);
  stage_to_lds<T>(smc, cost_part, cc);
  __syncthreads();
  const int cw0 = blockDim.x > 64 ? 64 : 0;  // first lane of the cost reduction: the second wave when there is one
  if (threadIdx.x == 0) {
    T merit = T(0);
    const int skip = p.N * 2;  // the |l_i|^2 terms of k = 0 do not enter (:421)
    for (int e = 0; e < skip && e < cm; e += 2) merit += sm[e];
    lds_ordered_visit<T>(sm + skip, cm - skip, [&](int, T x) { merit += x; });
    res[0] = T(0.5) * merit;
  }
  const int i = int(threadIdx.x) - cw0;
  if (i >= 0 && i < p.N) {
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
    int k = 0;
    for (; k + 8 <= p.T; k += 8) {
      T x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = smc[(k + u) * p.N + i];
#pragma unroll
      for (int u = 0; u < 8; u++) visit(k + u, x[u]);
    }
    for (; k < p.T; k++) visit(k, smc[k * p.N + i]);
    res[1 + i] = c;
    res[1 + kMaxPlayers + i] = T(te);
  }
  __syncthreads();
  *merit_out = res[0];
  return true;
}
// The cost totals merit_costs_reduce left in `sm`, written out (TotalCosts of an accepted iterate).
template <typename T>
__device__ __forceinline__ void costs_commit(const DevProblem& p, const T* sm, T* costs_out, int* t_extreme) {
  const T* const res = sm + p.T * p.N * 3;
  const int i = threadIdx.x;
  if (i < p.N) {
    costs_out[i] = res[1 + i];
    if (t_extreme) t_extreme[i] = int(res[1 + kMaxPlayers + i]);
  }
  __syncthreads();
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
    stage_to_lds<T>(sm, cost_part, count);
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
