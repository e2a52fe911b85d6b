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
  ilqg_solver_params prm;
};

// Per-instance workspace layout (in elements of T).
struct WsLayout {
  size_t xs1, us1, P1, al1, A, B, Q, l, R, r, lqscr, dx, mpart, cpart, ints, total;
  __host__ __device__ WsLayout(int n, int m, int N, int T, int Rsz, int rsz) {
    size_t o = 0;
    auto take = [&](size_t cnt) {
      const size_t at = o;
      o += (cnt + 1) & ~size_t(1);  // keep 16-byte alignment for fp64, 8 for fp32
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
    lqscr = take(size_t(T) * (N * (n + 1) + n));
    dx = take(size_t(T) * n);
    mpart = take(size_t(T) * N * 2);
    cpart = take(size_t(T) * N);
    ints = take(2 * kMaxPlayers);  // t_extreme as int32 (room for fp32 or fp64 elements)
    total = o;
  }
};

template <typename T, int NX, int NP, int MU>
__device__ void ilq_solve_instance(const DevProblem& p, const SolveArgs<T>& sa, int b, T* sm) {
  const int n = NX, N = NP, m = NP * MU, Tn = p.T;
  const PairTable& pt = p.pairs;
  const ilqg_solver_params& prm = sa.prm;
  const WsLayout L(n, m, N, Tn, pt.Rsz, pt.rsz);
  T* w = sa.ws + size_t(b) * sa.ws_stride;
  // two operating-point buffers and two strategy buffers; buffer 0 is the caller's
  T* xsb[2] = {sa.xs + size_t(b) * Tn * n, w + L.xs1};
  T* usb[2] = {sa.us + size_t(b) * Tn * m, w + L.us1};
  T* Pb[2] = {sa.P + size_t(b) * Tn * m * n, w + L.P1};
  T* alb[2] = {sa.alpha + size_t(b) * Tn * m, w + L.al1};
  int* t_extreme = reinterpret_cast<int*>(w + L.ints);
  const T* x0 = sa.x0 + size_t(b) * n;
  T* costs = sa.total_costs + size_t(b) * N;
  const int t = threadIdx.x;

  if (t < N) t_extreme[t] = 0;  // PlayerCost::time_of_extreme_cost_ starts at 0 (player_cost.h:70)
  __syncthreads();

  QuadArgs<T> qa;
  qa.lambdas = nullptr;
  qa.mu = T(10);
  qa.t_extreme = t_extreme;
  qa.t_init = 0.0;
  qa.A = w + L.A;
  qa.Bm = w + L.B;
  qa.Q = w + L.Q;
  qa.l = w + L.l;
  qa.R = w + L.R;
  qa.r = w + L.r;
  qa.merit_part = w + L.mpart;
  qa.cost_part = w + L.cpart;

  int cur = 0;   // op buffer holding the current (last accepted) operating point
  int sacc = 0;  // strategy buffer holding the last accepted strategies
  T acc_scale = T(1);

  // ---- initial rollout from the warm start (:100-104) ----
  {
    RolloutArgs<T> ra{x0, xsb[0], usb[0], Pb[0], alb[0], T(1), xsb[1], usb[1]};
    rollout_instance<T>(p, ra, sm);
    cur = 1;
  }
  // TotalCosts (:107) then quadraticise (:116) — costs first: they set t_extreme.
  {
    QuadArgs<T> qc = qa;
    qc.xs = xsb[cur];
    qc.us = usb[cur];
    qc.A = nullptr;
    qc.Q = nullptr;
    qc.merit_part = nullptr;
    for (int k = 0; k < Tn; k++) linquad_step<T>(p, qc, k, sm);
    costs_reduce<T>(p, w + L.cpart, costs, t_extreme);
    QuadArgs<T> qq = qa;
    qq.xs = xsb[cur];
    qq.us = usb[cur];
    qq.cost_part = nullptr;
    for (int k = 0; k < Tn; k++) linquad_step<T>(p, qq, k, sm);
  }

  T last_merit = dinf<T>();
  int num_iterations = 0;
  bool has_converged = false, ok = true;
  const int max_iters = sa.fixed_iters > 0 ? sa.fixed_iters : prm.max_solver_iters;
  while (num_iterations < max_iters && (sa.fixed_iters > 0 || !has_converged)) {
    num_iterations++;
    // ---- LQ game at the current operating point (:136-143) + ExpectedDecrease (:303) ----
    const int snew = 1 - sacc;
    LQArgs<T> la;
    la.A = w + L.A;
    la.Bm = w + L.B;
    la.Q = w + L.Q;
    la.l = w + L.l;
    la.R = w + L.R;
    la.r = w + L.r;
    la.x0 = nullptr;
    la.P = Pb[snew];
    la.alpha = alb[snew];
    la.dx = w + L.dx;
    la.scratch = w + L.lqscr;
    la.ed_out = sm + LQCfg<T, NX, NP, MU>::oX;  // any LDS slot free at the end of the sweep
    la.T_steps = Tn;
    la.adaptive = 1;
    lq_feedback_dispatch<T, NX, NP, MU>(la, pt, sm, false);
    __syncthreads();
    const T expected_decrease = sm[LQCfg<T, NX, NP, MU>::oX];
    __syncthreads();

    // ---- line search (:309-347) ----
    T step = T(prm.initial_alpha_scaling);
    {
      RolloutArgs<T> ra{xsb[cur], xsb[cur], usb[cur], Pb[snew], alb[snew], step, xsb[1 - cur], usb[1 - cur]};
      rollout_instance<T>(p, ra, sm);
    }
    bool accepted = !prm.linesearch;
    if (!prm.linesearch) {
      // the reference re-linearises every iteration but never re-quadraticises (:322)
      QuadArgs<T> ql = qa;
      ql.xs = xsb[1 - cur];
      ql.us = usb[1 - cur];
      ql.Q = nullptr;
      ql.merit_part = nullptr;
      for (int k = 0; k < Tn; k++) linquad_step<T>(p, ql, k, sm);
    } else {
      for (int bt = 0; bt < prm.max_backtracking_steps; bt++) {
        QuadArgs<T> qt = qa;
        qt.xs = xsb[1 - cur];
        qt.us = usb[1 - cur];
        for (int k = 0; k < Tn; k++) linquad_step<T>(p, qt, k, sm);
        const T merit = merit_reduce<T>(p, w + L.mpart, sm);
        const T scaled = T(prm.expected_decrease_fraction) * step * expected_decrease;
        if (last_merit - merit >= scaled) {  // CheckArmijoCondition :350-362
          const T diff = last_merit - merit;
          has_converged = (merit <= last_merit) && ((diff < T(0) ? -diff : diff) < T(prm.convergence_tolerance));
          last_merit = merit;
          accepted = true;
          break;
        }
        step *= T(prm.geometric_alpha_scaling);
        RolloutArgs<T> ra{xsb[cur], xsb[cur], usb[cur], Pb[snew], alb[snew], step, xsb[1 - cur], usb[1 - cur]};
        rollout_instance<T>(p, ra, sm);
      }
    }
    if (!accepted) {  // :146-155 — keep the last accepted iterate
      ok = false;
      break;
    }
    cur = 1 - cur;
    sacc = snew;
    acc_scale = step;
    costs_reduce<T>(p, w + L.cpart, costs, t_extreme);  // TotalCosts of the accepted iterate (:158)
  }

  // ---- hand the result back through buffer 0 ----
  __syncthreads();
  if (cur == 1) {
    for (int e = t; e < Tn * n; e += blockDim.x) xsb[0][e] = xsb[1][e];
    for (int e = t; e < Tn * m; e += blockDim.x) usb[0][e] = usb[1][e];
  }
  if (sacc == 1)
    for (int e = t; e < Tn * m * n; e += blockDim.x) Pb[0][e] = Pb[1][e];
  for (int e = t; e < Tn * m; e += blockDim.x) alb[0][e] = alb[sacc][e] * acc_scale;
  if (t == 0) {
    sa.iters[b] = num_iterations;
    sa.status[b] = ok ? 1 : 0;
    sa.converged[b] = has_converged ? 1 : 0;
  }
}

}  // namespace ilqg
