// ilqg_solve.hpp — the whole iterative-LQ loop of one game instance inside one
// persistent workgroup (gfx950).
//
// ILQSolver::Solve (src/ilq_solver.cpp:76-172) with ModifyLQStrategies (:289-348)
// inlined: rollout -> total costs -> quadraticise, then
//   { LQ sweep (+ expected decrease) -> scaled rollout -> Armijo back-tracking on the
//     KKT-residual merit -> total costs } until converged / max_solver_iters / failure.
// Instances finish after different numbers of outer iterations and back-tracks; since a
// workgroup owns its instance for the whole solve there are no masks, no host round
// trips and no compaction — a finished workgroup simply retires and the CU picks up the
// next instance.  Co-resident workgroups are in different phases, which is what keeps
// the CU busy: the latency-bound rollout of one instance overlaps the LDS/VALU-bound
// sweep of its neighbours.
#pragma once

#include "ilqg_lq.hpp"
#include "ilqg_lq_openloop.hpp"
#include "ilqg_stages.hpp"

namespace ilqg {

template <typename T>
struct SolveArgs {
  const T* x0;          // [B][n]
  T *xs, *us;           // [B][T][n], [B][T][m]   in: warm start, out: result (buffer 0)
  T *P, *alpha;         // [B][T][m*n], [B][T][m] in: warm start, out: result (buffer 0)
  T* total_costs;       // [B][N]
  int *iters, *status, *converged;
  T* ws;                // workspace, ws_stride elements per instance
  size_t ws_stride;
  int fixed_iters;
  int batch;
  int ol_row;           // elements per open-loop scratch row (0 when the feedback sweep is used)
  int al_mode;          // 1: AugmentedLagrangianSolver::Solve around the inner iLQ solve
  ilqg_solver_params prm;
  long long* prof;      // optional [B][8] shader-clock cycles per stage (diagnostics) or nullptr
};

// Elements of one open-loop scratch row ([X|y|W|w|M|m|Q l], OLCfg::ROW) from run-time dimensions.
__host__ __device__ inline int ol_row_elems(int n, int m, int N) {
  return (n * n + n + m * n + m + N * n * n + N * n + N * n + 3) & ~3;
}

// Per-instance workspace layout (in elements of T).
struct WsLayout {
  size_t xs1, us1, P1, al1, A, B, Q, l, R, r, lqscr, dx, mpart, cpart, ints, lambdas, wxs, wus, wP, wal, total;
  __host__ __device__ WsLayout(int n, int m, int N, int T, int Rsz, int rsz, int ol_row = 0, int num_constraints = 0,
                               int al_mode = 0) {
    size_t o = 0;
    auto take = [&](size_t cnt) {
      const size_t at = o;
      o += (cnt + 3) & ~size_t(3);  // every array starts 16-byte aligned (fp32 and fp64): LDS-DMA pieces
      return at;
    };
    xs1 = take(size_t(T) * n);
    us1 = take(size_t(T) * m);
    P1 = take(size_t(T) * m * n);
    al1 = take(size_t(T) * m);
    A = take(size_t(T) * n * n);
    B = take(size_t(T) * n * m);
    Q = take(size_t(T) * N * n * n);
    l = take(size_t(T) * N * n);
    R = take(size_t(T) * Rsz);
    r = take(size_t(T) * rsz);
    lqscr = take(size_t(T) * (size_t(ol_row) > size_t(N * (n + 1) + n) ? size_t(ol_row) : size_t(N * (n + 1) + n)));
    dx = take(size_t(T) * n);
    mpart = take(size_t(T) * N * 2);
    cpart = take(size_t(T) * N);
    ints = take(2 * kMaxPlayers);  // t_extreme as int32 (room for fp32 or fp64 elements)
    lambdas = take(size_t(num_constraints) * T);  // per-instance Constraint::lambdas_ (constraint.h:136)
    wxs = take(al_mode ? size_t(T) * n : 0);      // Problem's stored solution = warm start of the next inner solve
    wus = take(al_mode ? size_t(T) * m : 0);
    wP = take(al_mode ? size_t(T) * m * n : 0);
    wal = take(al_mode ? size_t(T) * m : 0);
    total = o;
  }
};

// The loop is written as a small stage interpreter so that each heavy stage (rollout, the
// per-step linearise+quadraticise, the LQ sweep) is instantiated — and inlined — exactly once
// in the persistent kernel; all transitions are wave-uniform.
template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void ilq_solve_instance(const DevProblem& p, const QuadTables<T>& tb, const SolveArgs<T>& sa,
                                                   int b, T* sm) {
  constexpr int n = NX, N = NP, m = NP * MU;
  const int Tn = p.T;
  const PairTable& pt = p.pairs;
  const ilqg_solver_params& prm = sa.prm;
  const WsLayout L(n, m, N, Tn, pt.Rsz, pt.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
  T* w = sa.ws + size_t(b) * sa.ws_stride;
  // two operating-point buffers and two strategy buffers; buffer 0 is the caller's
  T* const xs0 = sa.xs + size_t(b) * Tn * n;
  T* const us0 = sa.us + size_t(b) * Tn * m;
  T* const P0 = sa.P + size_t(b) * Tn * m * n;
  T* const al0 = sa.alpha + size_t(b) * Tn * m;
  auto XS = [&](int i) { return i ? w + L.xs1 : xs0; };
  auto US = [&](int i) { return i ? w + L.us1 : us0; };
  auto PB = [&](int i) { return i ? w + L.P1 : P0; };
  auto AL = [&](int i) { return i ? w + L.al1 : al0; };
  int* t_extreme = reinterpret_cast<int*>(w + L.ints);
  const T* x0 = sa.x0 + size_t(b) * n;
  T* costs = sa.total_costs + size_t(b) * N;
  const int t = threadIdx.x;

  if (t < N) t_extreme[t] = 0;  // PlayerCost::time_of_extreme_cost_ starts at 0 (player_cost.h:70)
  T* lambdas = w + L.lambdas;     // Constraint::lambdas_, zero-initialised (types.h:128)
  for (int e = t; e < p.num_constraints * Tn; e += blockDim.x) lambdas[e] = T(0);
  T mu = T(10);                   // Constraint::mu_ = kDefaultMu (src/constraint.cpp:61) — one per instance
  if (sa.al_mode) {               // Problem's stored solution (what OverwriteSolution maintains)
    for (int e = t; e < Tn * n; e += blockDim.x) (w + L.wxs)[e] = xs0[e];
    for (int e = t; e < Tn * m; e += blockDim.x) (w + L.wus)[e] = us0[e];
    for (int e = t; e < Tn * m * n; e += blockDim.x) (w + L.wP)[e] = P0[e];
    for (int e = t; e < Tn * m; e += blockDim.x) (w + L.wal)[e] = al0[e];
  }
  __syncthreads();

  long long pr_acc[5] = {0, 0, 0, 0, 0};
  const long long pr_start = clock64();

  enum { ST_ROLLOUT = 0, ST_QUAD = 1, ST_LQ = 2, ST_INNER_DONE = 3, ST_DONE = 4 };
  enum { Q_COSTS = 0, Q_INIT = 1, Q_TRIAL = 2, Q_LIN = 3 };
  int stage = ST_ROLLOUT, qmode = Q_COSTS;
  bool initial = true;
  int cur = 0;   // op buffer holding the current (last accepted) operating point
  int sacc = 0;  // strategy buffer holding the last accepted strategies
  T acc_scale = T(1), step = T(1);
  T last_merit = dinf<T>(), expected_decrease = dinf<T>();
  int num_iterations = 0, bt = 0, accepted_iters = 0;
  bool has_converged = false, ok = true;
  // AugmentedLagrangianSolver builds its inner ILQSolver with unconstrained_solver_max_iters
  // (augmented_lagrangian_solver.h:80-84)
  const int max_iters = sa.fixed_iters > 0 ? sa.fixed_iters
                                          : (sa.al_mode ? prm.unconstrained_solver_max_iters : prm.max_solver_iters);
  int logged = 0, inner_calls = 0;
  bool al_success = true;
  T max_err = dinf<T>();

#pragma unroll 1
  while (stage != ST_DONE) {
    __syncthreads();  // stage boundary: global-memory hand-off between lanes
    const long long pr_t0 = clock64();
    const int stage_was = stage;
    if (stage == ST_ROLLOUT) {
      // initial: from the warm start (:100-104); later: trial point of the line search (:309-342)
      const int snew = 1 - sacc;
      RolloutArgs<T> ra;
      ra.x0 = initial ? x0 : XS(cur);
      ra.xs_ref = initial ? XS(0) : XS(cur);
      ra.us_ref = initial ? US(0) : US(cur);
      ra.P = initial ? PB(0) : PB(snew);
      ra.alpha = initial ? AL(0) : AL(snew);
      ra.alpha_scale = initial ? T(1) : step;
      ra.xs = initial ? XS(1) : XS(1 - cur);
      ra.us = initial ? US(1) : US(1 - cur);
      rollout_instance<T, NX, NP * MU>(p, ra, sm);
      if (initial) {
        cur = 1;
        qmode = Q_COSTS;  // TotalCosts (:107) before quadraticising (:116): costs set t_extreme
      } else {
        qmode = prm.linesearch ? Q_TRIAL : Q_LIN;
      }
      stage = ST_QUAD;
    } else if (stage == ST_QUAD) {
      const int at = (qmode == Q_COSTS || qmode == Q_INIT) ? cur : 1 - cur;
      QuadArgs<T> qa;
      qa.xs = XS(at);
      qa.us = US(at);
      qa.lambdas = p.num_constraints > 0 ? lambdas : nullptr;
      qa.mu = mu;
      qa.t_extreme = t_extreme;
      qa.t_init = 0.0;
      const bool lin = qmode != Q_COSTS, quad = qmode == Q_INIT || qmode == Q_TRIAL;
      qa.A = lin ? w + L.A : nullptr;
      qa.Bm = lin ? w + L.B : nullptr;
      qa.Q = quad ? w + L.Q : nullptr;
      qa.l = quad ? w + L.l : nullptr;
      qa.R = quad ? w + L.R : nullptr;
      qa.r = quad ? w + L.r : nullptr;
      qa.merit_part = qmode == Q_TRIAL ? w + L.mpart : nullptr;
      qa.cost_part = (qmode == Q_COSTS || qmode == Q_TRIAL || qmode == Q_LIN) ? w + L.cpart : nullptr;
#pragma unroll 1
      for (int k = 0; k < Tn; k++) linquad_step<T, NX, NP * MU, NP>(p, tb, qa, k, sm);
      if (qmode == Q_COSTS) {
        costs_reduce<T>(p, w + L.cpart, costs, t_extreme);
        qmode = Q_INIT;
      } else if (qmode == Q_INIT) {
        initial = false;
        stage = (num_iterations < max_iters) ? ST_LQ : ST_INNER_DONE;
      } else {
        bool accepted = true;
        if (qmode == Q_TRIAL) {
          const T merit = merit_reduce<T>(p, w + L.mpart, sm);
          const T scaled = T(prm.expected_decrease_fraction) * step * expected_decrease;
          accepted = (last_merit - merit >= scaled);  // CheckArmijoCondition :350-362
          if (accepted) {
            const T diff = last_merit - merit;
            has_converged =
                (merit <= last_merit) && ((diff < T(0) ? -diff : diff) < T(prm.convergence_tolerance));
            last_merit = merit;
          }
        }
        if (accepted) {
          cur = 1 - cur;
          sacc = 1 - sacc;
          acc_scale = step;
          accepted_iters++;
          costs_reduce<T>(p, w + L.cpart, costs, t_extreme);  // TotalCosts of the accepted iterate (:158)
          stage = (num_iterations < max_iters && (sa.fixed_iters > 0 || !has_converged)) ? ST_LQ : ST_INNER_DONE;
        } else {
          bt++;
          if (bt >= prm.max_backtracking_steps) {  // :346-347, :146-155 — keep the last accepted iterate
            ok = false;
            stage = ST_INNER_DONE;
          } else {
            step *= T(prm.geometric_alpha_scaling);
            stage = ST_ROLLOUT;
          }
        }
      }
    } else if (stage == ST_LQ) {  // LQ game at the current operating point (:136-143) + ExpectedDecrease (:303)
      num_iterations++;
      LQArgs<T> la;
      la.A = w + L.A;
      la.Bm = w + L.B;
      la.Q = w + L.Q;
      la.l = w + L.l;
      la.R = w + L.R;
      la.r = w + L.r;
      la.x0 = nullptr;
      la.P = PB(1 - sacc);
      la.alpha = AL(1 - sacc);
      la.dx = w + L.dx;
      la.scratch = w + L.lqscr;
      la.ed_out = sm + LQCfg<T, NX, NP, MU>::oX;  // an LDS slot that is free once the sweep ends
      la.T_steps = Tn;
      la.adaptive = 1;
      la.ph = sa.prof ? sa.prof + size_t(b) * 16 + 8 : nullptr;
      if (prm.open_loop)
        lq_openloop_instance<T, NX, NP, MU>(la, pt, sm);  // SolverParams::open_loop (ilq_solver.h:76-81)
      else
        lq_feedback_dispatch<T, NX, NP, MU>(la, pt, sm);
      __syncthreads();
      expected_decrease = sm[LQCfg<T, NX, NP, MU>::oX];
      __syncthreads();
      step = T(prm.initial_alpha_scaling);
      bt = 0;
      stage = ST_ROLLOUT;
    } else {  // ST_INNER_DONE: one ILQSolver::Solve call has returned
      // ---- the log's final iterate goes back through buffer 0 (alpha carries the accepted step) ----
      if (cur == 1) {
        for (int e = t; e < Tn * n; e += blockDim.x) xs0[e] = (w + L.xs1)[e];
        for (int e = t; e < Tn * m; e += blockDim.x) us0[e] = (w + L.us1)[e];
      }
      if (sacc == 1)
        for (int e = t; e < Tn * m * n; e += blockDim.x) P0[e] = (w + L.P1)[e];
      {
        const T* src = AL(sacc);
        for (int e = t; e < Tn * m; e += blockDim.x) al0[e] = src[e] * acc_scale;
      }
      __syncthreads();
      stage = ST_DONE;
      if (sa.al_mode) {  // AugmentedLagrangianSolver::Solve, src/augmented_lagrangian_solver.cpp:72-210
        logged += 1 + accepted_iters;  // SolverLog entries of this inner call (:94, :185)
        al_success = al_success && ok;
        if (inner_calls > 0 && !ok) {  // :166-178
          for (int e = t; e < p.num_constraints * Tn; e += blockDim.x)
            lambdas[e] *= T(prm.geometric_lambda_downscaling);
          mu *= T(prm.geometric_mu_downscaling);
          __syncthreads();
        }
        inner_calls++;
        if (p.num_constraints > 0 && logged < prm.max_solver_iters &&
            max_err > T(prm.constraint_error_tolerance)) {
          // ---- multiplier update at the final operating point (:116-140) ----
          T my_err = -dinf<T>();
          if (t < p.num_constraints) {
            int ti = 0;
            for (int e = 0; e < p.num_terms; e++)
              if (tb.terms[e].slot == t) ti = e;
            const DevTerm c = tb.terms[ti];
            const bool on_state = c.role == ILQG_ROLE_STATE_CONSTRAINT;
            for (int k = 0; k < Tn; k++) {
              const T* v = on_state ? xs0 + size_t(k) * n : us0 + size_t(k) * m + p.uoff[c.arg];
              const T err = term_evaluate_leaf<T>(tb, ti, v, c.arg_dim);
              my_err = err > my_err ? err : my_err;
              // Constraint::IncrementLambda (constraint.h:98-102) at TimeIndex(t0 + dt*float(k))
              const double tt = 0.0 + p.dt * double(float(k));
              const int tidx = int(static_cast<size_t>(tt / p.dt));
              const T nl = lambdas[t * Tn + tidx] + mu * err;
              lambdas[t * Tn + tidx] = nl > T(0) ? nl : T(0);
            }
          }
          if (t < 64) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
              const T o = __shfl_xor(my_err, off, 64);
              my_err = o > my_err ? o : my_err;
            }
            if (t == 0) sm[0] = my_err;
          }
          __syncthreads();
          max_err = sm[0];
          __syncthreads();
          mu *= T(prm.geometric_mu_scaling);  // :143
          // Problem::OverwriteSolution only after a successful inner solve (:151-154)
          if (ok) {
            for (int e = t; e < Tn * n; e += blockDim.x) (w + L.wxs)[e] = xs0[e];
            for (int e = t; e < Tn * m; e += blockDim.x) (w + L.wus)[e] = us0[e];
            for (int e = t; e < Tn * m * n; e += blockDim.x) (w + L.wP)[e] = P0[e];
            for (int e = t; e < Tn * m; e += blockDim.x) (w + L.wal)[e] = al0[e];
          } else {
            for (int e = t; e < Tn * n; e += blockDim.x) xs0[e] = (w + L.wxs)[e];
            for (int e = t; e < Tn * m; e += blockDim.x) us0[e] = (w + L.wus)[e];
            for (int e = t; e < Tn * m * n; e += blockDim.x) P0[e] = (w + L.wP)[e];
            for (int e = t; e < Tn * m; e += blockDim.x) al0[e] = (w + L.wal)[e];
          }
          __syncthreads();
          // ---- next ILQSolver::Solve call: fresh locals, persistent last_merit / t_extreme ----
          stage = ST_ROLLOUT;
          initial = true;
          cur = 0;
          sacc = 0;
          acc_scale = T(1);
          num_iterations = 0;
          accepted_iters = 0;
          has_converged = false;
          ok = true;
        }
      }
    }
    pr_acc[stage_was] += clock64() - pr_t0;
  }
  if (sa.al_mode && p.num_constraints > 0 && max_err > T(prm.constraint_error_tolerance)) al_success = false;  // :188-191

  if (t == 0 && sa.prof) {
    long long* o = sa.prof + size_t(b) * 16;
    o[0] = pr_acc[0]; o[1] = pr_acc[1]; o[2] = pr_acc[2]; o[3] = 0; o[4] = clock64() - pr_start;
  }
  if (t == 0) {
    sa.iters[b] = sa.al_mode ? logged : num_iterations;
    sa.status[b] = (sa.al_mode ? al_success : ok) ? 1 : 0;
    sa.converged[b] = has_converged ? 1 : 0;
  }
}

}  // namespace ilqg
