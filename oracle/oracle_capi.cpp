// oracle_capi.cpp — C entry points over ilqg_oracle.hpp for ctypes.  TEST INFRASTRUCTURE ONLY
// (see the header of ilqg_oracle.hpp).  All pointers are HOST pointers; array
// layouts are exactly those of include/ilqg.h so tests can feed both sides the
// same buffers.  `threads` > 1 parallelises over instances with OpenMP (used
// only by bench.py's cpu_baseline leg; the reference itself is single-threaded).
#include <cstdio>
#include <memory>

#include "ilqg_oracle.hpp"

using namespace oracle;

namespace {

template <class S>
void UnpackLQ(const Problem<S>& p, int T, const S* A, const S* Bm, const S* Q, const S* l, const S* R,
              const S* r, LQInputs<S>* in) {
  const int n = p.n, m = p.m, N = p.N;
  in->A.assign(T, Mat<S>(n, n));
  in->B.assign(T, Mat<S>(n, m));
  in->q.assign(T, std::vector<Quad<S>>(N));
  const int np = (int)p.pairs.size();
  for (int k = 0; k < T; k++) {
    std::memcpy(in->A[k].d.data(), A + size_t(k) * n * n, sizeof(S) * n * n);
    std::memcpy(in->B[k].d.data(), Bm + size_t(k) * n * m, sizeof(S) * n * m);
    for (int i = 0; i < N; i++) {
      Quad<S>& q = in->q[k][i];
      q.Q = Mat<S>(n, n);
      std::memcpy(q.Q.d.data(), Q + (size_t(k) * N + i) * n * n, sizeof(S) * n * n);
      q.l.assign(l + (size_t(k) * N + i) * n, l + (size_t(k) * N + i + 1) * n);
      q.R.resize(np);
      q.r.resize(np);
      q.has.assign(np, 0);
      for (int pr = 0; pr < np; pr++) {
        if (p.pairs[pr].i != i) continue;
        const int mj = p.udim(p.pairs[pr].j);
        q.R[pr] = Mat<S>(mj, mj);
        std::memcpy(q.R[pr].d.data(), R + size_t(k) * p.Rsz + p.roff[pr], sizeof(S) * mj * mj);
        q.r[pr].assign(r + size_t(k) * p.rsz + p.rgoff[pr], r + size_t(k) * p.rsz + p.rgoff[pr] + mj);
        q.has[pr] = 1;
      }
    }
  }
}

template <class S>
void PackStrategies(const Problem<S>& p, const Strategies<S>& st, S* P, S* alpha) {
  const int T = (int)st.P.size(), n = p.n, m = p.m;
  for (int k = 0; k < T; k++) {
    std::memcpy(P + size_t(k) * m * n, st.P[k].d.data(), sizeof(S) * m * n);
    std::memcpy(alpha + size_t(k) * m, st.alpha[k].data(), sizeof(S) * m);
  }
}
template <class S>
void UnpackStrategies(const Problem<S>& p, int T, const S* P, const S* alpha, Strategies<S>* st) {
  const int n = p.n, m = p.m;
  *st = Strategies<S>(T, n, m);
  for (int k = 0; k < T; k++) {
    std::memcpy(st->P[k].d.data(), P + size_t(k) * m * n, sizeof(S) * m * n);
    std::memcpy(st->alpha[k].data(), alpha + size_t(k) * m, sizeof(S) * m);
  }
}
template <class S>
void UnpackTraj(const Problem<S>& p, int T, const S* xs, const S* us, Trajectory<S>* tr) {
  *tr = Trajectory<S>(T, p.n, p.m);
  for (int k = 0; k < T; k++) {
    std::memcpy(tr->xs[k].data(), xs + size_t(k) * p.n, sizeof(S) * p.n);
    std::memcpy(tr->us[k].data(), us + size_t(k) * p.m, sizeof(S) * p.m);
  }
}
template <class S>
void PackTraj(const Problem<S>& p, const Trajectory<S>& tr, S* xs, S* us) {
  const int T = (int)tr.xs.size();
  for (int k = 0; k < T; k++) {
    std::memcpy(xs + size_t(k) * p.n, tr.xs[k].data(), sizeof(S) * p.n);
    std::memcpy(us + size_t(k) * p.m, tr.us[k].data(), sizeof(S) * p.m);
  }
}
template <class S>
void PackQuad(const Problem<S>& p, const LQInputs<S>& lq, S* Q, S* l, S* R, S* r) {
  const int T = (int)lq.q.size(), n = p.n, N = p.N;
  for (int k = 0; k < T; k++) {
    std::memset(R + size_t(k) * p.Rsz, 0, sizeof(S) * p.Rsz);
    std::memset(r + size_t(k) * p.rsz, 0, sizeof(S) * p.rsz);
    for (int i = 0; i < N; i++) {
      const Quad<S>& q = lq.q[k][i];
      std::memcpy(Q + (size_t(k) * N + i) * n * n, q.Q.d.data(), sizeof(S) * n * n);
      std::memcpy(l + (size_t(k) * N + i) * n, q.l.data(), sizeof(S) * n);
      for (size_t pr = 0; pr < p.pairs.size(); pr++) {
        if (p.pairs[pr].i != i || !q.has[pr]) continue;
        const int mj = p.udim(p.pairs[pr].j);
        std::memcpy(R + size_t(k) * p.Rsz + p.roff[pr], q.R[pr].d.data(), sizeof(S) * mj * mj);
        std::memcpy(r + size_t(k) * p.rsz + p.rgoff[pr], q.r[pr].data(), sizeof(S) * mj);
      }
    }
  }
}

template <class S>
int LQBatch(const ilqg_dims* d, const void* A_, const void* B_, const void* Q_, const void* l_, const void* R_,
            const void* r_, const ilqg_pair* pairs, int npairs, const void* x0_, void* P_, void* alpha_,
            void* dx_, void* co_, int open_loop, int threads) {
  Problem<S> p(*d, pairs, npairs);
  for (int i = 0; i < p.N; i++)
    if (p.pairIndex(i, i) < 0) return ILQG_ERR_INVALID;  // CHECK at lq_feedback_solver.cpp:139-140
  const int T = d->T, n = p.n, m = p.m, N = p.N;
  const S *A = (const S*)A_, *B = (const S*)B_, *Q = (const S*)Q_, *l = (const S*)l_, *R = (const S*)R_,
          *r = (const S*)r_, *x0 = (const S*)x0_;
  S *P = (S*)P_, *alpha = (S*)alpha_, *dx = (S*)dx_, *co = (S*)co_;
#pragma omp parallel for num_threads(threads) schedule(dynamic) if (threads > 1)
  for (int b = 0; b < d->batch; b++) {
    LQInputs<S> in;
    UnpackLQ(p, T, A + size_t(b) * T * n * n, B + size_t(b) * T * n * m, Q + size_t(b) * T * N * n * n,
             l + size_t(b) * T * N * n, R + size_t(b) * T * p.Rsz, r + size_t(b) * T * p.rsz, &in);
    Vec<S> x0v(n, S(0));
    if (x0) x0v.assign(x0 + size_t(b) * n, x0 + size_t(b + 1) * n);
    Strategies<S> st;
    std::vector<Vec<S>> dxs;
    std::vector<std::vector<Vec<S>>> cos;
    if (open_loop)
      SolveLQOpenLoop(p, in, x0v, &st, dx ? &dxs : nullptr, co ? &cos : nullptr);
    else
      SolveLQFeedback(p, in, x0v, d->adaptive_regularization != 0, &st, dx || co ? &dxs : nullptr,
                      co ? &cos : nullptr);
    PackStrategies(p, st, P + size_t(b) * T * m * n, alpha + size_t(b) * T * m);
    if (dx)
      for (int k = 0; k < T; k++) std::memcpy(dx + (size_t(b) * T + k) * n, dxs[k].data(), sizeof(S) * n);
    if (co)
      for (int k = 0; k < T; k++)
        for (int i = 0; i < N; i++)
          std::memcpy(co + ((size_t(b) * T + k) * N + i) * n, cos[k][i].data(), sizeof(S) * n);
  }
  return ILQG_OK;
}

struct OracleProblem {
  std::vector<ilqg_cost_term> terms;
  std::vector<int32_t> poly_off;
  std::vector<float> poly_pts;
  std::vector<float> dense;
  ilqg_problem_desc desc;
  std::unique_ptr<Problem<float>> pf;
  std::unique_ptr<Problem<double>> pd;
};
template <class S>
const Problem<S>& Get(const OracleProblem* op);
template <>
const Problem<float>& Get<float>(const OracleProblem* op) { return *op->pf; }
template <>
const Problem<double>& Get<double>(const OracleProblem* op) { return *op->pd; }

template <class S>
ALState<S> MakeAL(const Problem<S>& p, const void* lambdas, const void* mu, int b) {
  ALState<S> al(p.num_constraints, p.T, p.dt);
  if (lambdas && p.num_constraints)
    al.lambdas.assign((const S*)lambdas + size_t(b) * p.num_constraints * p.T,
                      (const S*)lambdas + size_t(b + 1) * p.num_constraints * p.T);
  if (mu) al.mu = ((const S*)mu)[b];
  return al;
}

template <class S>
void RolloutBatch(const OracleProblem* op, int batch, const void* x0, const void* xs_ref, const void* us_ref,
                  const void* P, const void* alpha, const void* alpha_scale, void* xs, void* us) {
  const Problem<S>& p = Get<S>(op);
  const int T = p.T, n = p.n, m = p.m;
  for (int b = 0; b < batch; b++) {
    Trajectory<S> last, cur;
    Strategies<S> st;
    UnpackTraj(p, T, (const S*)xs_ref + size_t(b) * T * n, (const S*)us_ref + size_t(b) * T * m, &last);
    UnpackStrategies(p, T, (const S*)P + size_t(b) * T * m * n, (const S*)alpha + size_t(b) * T * m, &st);
    if (alpha_scale)
      for (auto& a : st.alpha)
        for (auto& v : a) v *= ((const S*)alpha_scale)[b];
    Vec<S> x0v((const S*)x0 + size_t(b) * n, (const S*)x0 + size_t(b + 1) * n);
    Rollout(p, x0v, last, st, &cur);
    PackTraj(p, cur, (S*)xs + size_t(b) * T * n, (S*)us + size_t(b) * T * m);
  }
}

template <class S>
void LinearizeBatch(const OracleProblem* op, int batch, const void* xs, const void* us, void* A, void* Bm) {
  const Problem<S>& p = Get<S>(op);
  const int T = p.T, n = p.n, m = p.m;
  for (int b = 0; b < batch; b++) {
    Trajectory<S> tr;
    UnpackTraj(p, T, (const S*)xs + size_t(b) * T * n, (const S*)us + size_t(b) * T * m, &tr);
    LQInputs<S> lq;
    ComputeLinearization(p, tr, &lq);
    for (int k = 0; k < T; k++) {
      std::memcpy((S*)A + (size_t(b) * T + k) * n * n, lq.A[k].d.data(), sizeof(S) * n * n);
      std::memcpy((S*)Bm + (size_t(b) * T + k) * n * m, lq.B[k].d.data(), sizeof(S) * n * m);
    }
  }
}

template <class S>
void QuadraticizeBatch(const OracleProblem* op, int batch, const void* xs, const void* us, const void* lambdas,
                       const void* mu, const int32_t* t_extreme, void* Q, void* l, void* R, void* r) {
  const Problem<S>& p = Get<S>(op);
  const int T = p.T, n = p.n, m = p.m, N = p.N;
  for (int b = 0; b < batch; b++) {
    Trajectory<S> tr;
    UnpackTraj(p, T, (const S*)xs + size_t(b) * T * n, (const S*)us + size_t(b) * T * m, &tr);
    std::vector<int> te(N, 0);
    if (t_extreme)
      for (int i = 0; i < N; i++) te[i] = t_extreme[size_t(b) * N + i];
    ALState<S> al = MakeAL(p, lambdas, mu, b);
    LQInputs<S> lq;
    ComputeQuadraticization(p, tr, te, &al, &lq);
    PackQuad(p, lq, (S*)Q + size_t(b) * T * N * n * n, (S*)l + size_t(b) * T * N * n,
             (S*)R + size_t(b) * T * p.Rsz, (S*)r + size_t(b) * T * p.rsz);
  }
}

template <class S>
void TotalCostsBatch(const OracleProblem* op, int batch, const void* xs, const void* us, void* costs,
                     int32_t* t_extreme) {
  const Problem<S>& p = Get<S>(op);
  const int T = p.T, n = p.n, m = p.m, N = p.N;
  for (int b = 0; b < batch; b++) {
    Trajectory<S> tr;
    UnpackTraj(p, T, (const S*)xs + size_t(b) * T * n, (const S*)us + size_t(b) * T * m, &tr);
    std::vector<int> te(N, 0);
    if (t_extreme)
      for (int i = 0; i < N; i++) te[i] = t_extreme[size_t(b) * N + i];
    Vec<S> c;
    TotalCosts(p, tr, &c, &te);
    std::memcpy((S*)costs + size_t(b) * N, c.data(), sizeof(S) * N);
    if (t_extreme)
      for (int i = 0; i < N; i++) t_extreme[size_t(b) * N + i] = te[i];
  }
}

template <class S>
void SolveBatch(const OracleProblem* op, int batch, const void* x0, void* xs, void* us, void* P, void* alpha,
                void* costs, int32_t* iters, int32_t* status, int32_t* converged, int fixed_iters, void* rawP,
                void* rawAlpha, void* merit_log, int merit_log_len, int threads, const void* forced_steps = nullptr) {
  const Problem<S>& p = Get<S>(op);
  const int T = p.T, n = p.n, m = p.m, N = p.N;
#pragma omp parallel for num_threads(threads) schedule(dynamic) if (threads > 1)
  for (int b = 0; b < batch; b++) {
    Trajectory<S> tr;
    Strategies<S> st, raw;
    UnpackTraj(p, T, (S*)xs + size_t(b) * T * n, (S*)us + size_t(b) * T * m, &tr);
    UnpackStrategies(p, T, (S*)P + size_t(b) * T * m * n, (S*)alpha + size_t(b) * T * m, &st);
    Vec<S> x0v((const S*)x0 + size_t(b) * n, (const S*)x0 + size_t(b + 1) * n);
    ILQState<S> state;
    ALState<S> al(p.num_constraints, T, p.dt);
    std::vector<IterLog<S>> log;
    Vec<S> fc;
    int it = 0, conv = 0;
    bool ok;
    if (fixed_iters == -1) {  // AugmentedLagrangianSolver::Solve
      S maxerr;
      ok = SolveAL(p, x0v, &tr, &st, &fc, &it, &maxerr, &conv);
    } else {
      const S* fs = (forced_steps && fixed_iters > 0) ? (const S*)forced_steps + size_t(b) * fixed_iters : nullptr;
      ok = SolveILQ(p, x0v, &tr, &st, &state, &al, fixed_iters, &log, &fc, &it, &conv, &raw, (int*)nullptr, fs);
    }
    PackTraj(p, tr, (S*)xs + size_t(b) * T * n, (S*)us + size_t(b) * T * m);
    PackStrategies(p, st, (S*)P + size_t(b) * T * m * n, (S*)alpha + size_t(b) * T * m);
    if (rawP && !raw.P.empty())
      PackStrategies(p, raw, (S*)rawP + size_t(b) * T * m * n, (S*)rawAlpha + size_t(b) * T * m);
    std::memcpy((S*)costs + size_t(b) * N, fc.data(), sizeof(S) * N);
    iters[b] = it;
    status[b] = ok ? 1 : 0;
    converged[b] = conv;
    if (merit_log) {
      // per iteration: merit, expected_decrease, step, backtracks
      S* ml = (S*)merit_log + size_t(b) * merit_log_len * 4;
      for (int q = 0; q < merit_log_len * 4; q++) ml[q] = std::numeric_limits<S>::quiet_NaN();
      for (size_t q = 0; q < log.size() && (int)q < merit_log_len; q++) {
        ml[4 * q + 0] = log[q].merit;
        ml[4 * q + 1] = log[q].expected_decrease;
        ml[4 * q + 2] = log[q].step;
        ml[4 * q + 3] = S(log[q].backtracks);
      }
    }
  }
}

}  // namespace

template <class S>
void RecedingHorizonBatch(const OracleProblem* op, int batch, const void* x0, double t0, double planner_runtime,
                          double plan_t0, void* xs, void* us, void* P, void* alpha, void* x0_next, int32_t* first_step,
                          double* new_plan_t0) {
  const Problem<S>& p = Get<S>(op);
  const int T = p.T, n = p.n, m = p.m;
  const RecedingHorizonTimes tm = RecedingHorizonTimesOf(t0, planner_runtime, plan_t0, p.dt);
  if (new_plan_t0) *new_plan_t0 = tm.new_plan_t0;
  for (int b = 0; b < batch; b++) {
    Trajectory<S> tr;
    Strategies<S> st;
    UnpackTraj(p, T, (S*)xs + size_t(b) * T * n, (S*)us + size_t(b) * T * m, &tr);
    UnpackStrategies(p, T, (S*)P + size_t(b) * T * m * n, (S*)alpha + size_t(b) * T * m, &st);
    Vec<S> x0v((const S*)x0 + size_t(b) * n, (const S*)x0 + size_t(b + 1) * n), xn;
    first_step[b] = RecedingHorizonShift(p, tm, x0v, &tr, &st, &xn);
    PackTraj(p, tr, (S*)xs + size_t(b) * T * n, (S*)us + size_t(b) * T * m);
    PackStrategies(p, st, (S*)P + size_t(b) * T * m * n, (S*)alpha + size_t(b) * T * m);
    std::memcpy((S*)x0_next + size_t(b) * n, xn.data(), sizeof(S) * n);
  }
}


// ---- receding-horizon harness on plans stored [B][cap][...] with per-instance length and start time ----
template <class S>
Plan<S> LoadPlan(const Problem<S>& p, int cap, int len, double t0, const void* xs, const void* us, const void* P,
                 const void* alpha, int b) {
  Plan<S> pl;
  const int n = p.n, m = p.m;
  UnpackTraj(p, len, (const S*)xs + size_t(b) * cap * n, (const S*)us + size_t(b) * cap * m, &pl.op);
  UnpackStrategies(p, len, (const S*)P + size_t(b) * cap * m * n, (const S*)alpha + size_t(b) * cap * m, &pl.st);
  pl.t0 = t0;
  return pl;
}
template <class S>
void StorePlan(const Problem<S>& p, int cap, const Plan<S>& pl, void* xs, void* us, void* P, void* alpha, int b) {
  const int n = p.n, m = p.m;
  PackTraj(p, pl.op, (S*)xs + size_t(b) * cap * n, (S*)us + size_t(b) * cap * m);
  PackStrategies(p, pl.st, (S*)P + size_t(b) * cap * m * n, (S*)alpha + size_t(b) * cap * m);
}

template <class S>
void PlanIntegrateBatch(const OracleProblem* op, int batch, int cap, const void* xs, const void* us, const void* P,
                        const void* alpha, const int32_t* len, const double* t0, double t_from, double t_to,
                        double must_contain, void* x, int32_t* active) {
  const Problem<S>& p = Get<S>(op);
  for (int b = 0; b < batch; b++) {
    if (!active[b]) continue;
    const Plan<S> pl = LoadPlan<S>(p, cap, len[b], t0[b], xs, us, P, alpha, b);
    if (!PlanContainsTime(pl, must_contain, p.dt) || !IntegrateIntervalValid(p, pl, t_from, t_to)) {
      active[b] = 0;
      continue;
    }
    S* xb = (S*)x + size_t(b) * p.n;
    const Vec<S> xn = IntegrateInterval(p, t_from, t_to, Vec<S>(xb, xb + p.n), pl);
    std::memcpy(xb, xn.data(), sizeof(S) * p.n);
  }
}

template <class S>
void RecedingSyncBatch(const OracleProblem* op, int batch, int cap, const void* pxs, const void* pus, const void* pP,
                       const void* palpha, const int32_t* len, const double* t0, const void* x, double t,
                       double planner_runtime, void* xs, void* us, void* P, void* alpha, void* x0_next,
                       double* solve_t0, int32_t* first_step, int32_t* active) {
  const Problem<S>& p = Get<S>(op);
  for (int b = 0; b < batch; b++) {
    if (!active[b]) continue;
    Plan<S> pl = LoadPlan<S>(p, cap, len[b], t0[b], pxs, pus, pP, palpha, b);
    if (!RecedingHorizonTimesValid(p, pl, t, planner_runtime)) {
      active[b] = 0;
      first_step[b] = -1;
      continue;
    }
    const S* xb = (const S*)x + size_t(b) * p.n;
    Vec<S> xn;
    first_step[b] = SetUpNextRecedingHorizon(p, Vec<S>(xb, xb + p.n), t, planner_runtime, &pl, &xn);
    StorePlan<S>(p, p.T, pl, xs, us, P, alpha, b);
    std::memcpy((S*)x0_next + size_t(b) * p.n, xn.data(), sizeof(S) * p.n);
    solve_t0[b] = pl.t0;
  }
}

template <class S>
void SpliceBatch(const OracleProblem* op, int batch, int cap, void* pxs, void* pus, void* pP, void* palpha, int32_t* len,
                 double* t0, const void* xs, const void* us, const void* P, const void* alpha, const double* solve_t0,
                 const int32_t* converged, const int32_t* active) {
  const Problem<S>& p = Get<S>(op);
  for (int b = 0; b < batch; b++) {
    if (active && !active[b]) continue;
    const Plan<S> sol = LoadPlan<S>(p, p.T, p.T, solve_t0[b], xs, us, P, alpha, b);
    if (len[b] == 0) {  // SolutionSplicer's constructor
      StorePlan<S>(p, cap, sol, pxs, pus, pP, palpha, b);
      len[b] = p.T;
      t0[b] = sol.t0;
      continue;
    }
    if (converged && !converged[b]) continue;
    Plan<S> pl = LoadPlan<S>(p, cap, len[b], t0[b], pxs, pus, pP, palpha, b);
    SplicePlan(p, sol, &pl);
    StorePlan<S>(p, cap, pl, pxs, pus, pP, palpha, b);
    len[b] = pl.len();
    t0[b] = pl.t0;
  }
}

// One solver object per instance called again: `last_merit` [B] carries ILQSolver::last_merit_function_value_.
template <class S>
void SolveResumeBatch(const OracleProblem* op, int batch, const void* x0, void* xs, void* us, void* P, void* alpha,
                      void* costs, int32_t* iters, int32_t* status, int32_t* converged, int al_mode, void* last_merit,
                      int threads) {
  const Problem<S>& p = Get<S>(op);
  const int T = p.T, n = p.n, m = p.m, N = p.N;
#pragma omp parallel for num_threads(threads) schedule(dynamic) if (threads > 1)
  for (int b = 0; b < batch; b++) {
    Plan<S> pl = LoadPlan<S>(p, T, T, 0.0, xs, us, P, alpha, b);
    Vec<S> x0v((const S*)x0 + size_t(b) * n, (const S*)x0 + size_t(b + 1) * n);
    ILQState<S> state;
    state.last_merit = ((S*)last_merit)[b];
    ALState<S> al(p.num_constraints, T, p.dt);
    Vec<S> fc;
    int it = 0, conv = 0;
    bool ok;
    if (al_mode) {
      S maxerr;
      ok = SolveAL(p, x0v, &pl.op, &pl.st, &fc, &it, &maxerr, &conv, &state);
    } else {
      ok = SolveILQ(p, x0v, &pl.op, &pl.st, &state, &al, 0, (std::vector<IterLog<S>>*)nullptr, &fc, &it, &conv);
    }
    ((S*)last_merit)[b] = state.last_merit;
    StorePlan<S>(p, T, pl, xs, us, P, alpha, b);
    std::memcpy((S*)costs + size_t(b) * N, fc.data(), sizeof(S) * N);
    iters[b] = it;
    status[b] = ok ? 1 : 0;
    converged[b] = conv;
  }
}

// The whole harness per instance.  Records are stored [B][max_records][...].
template <class S>
void SimulateBatch(const OracleProblem* op, int batch, const void* x_init, double final_time, double planner_runtime,
                   double extra_time, double solve_time, int use_al, int max_records, int32_t* num_records,
                   double* t_call, void* x_measured, void* x0, double* plan_t0, int32_t* first_step, void* xs, void* us,
                   void* P, void* alpha, int32_t* iters, int32_t* ok, int32_t* converged, int32_t* max_bt, int cap, void* fxs,
                   void* fus, void* fP, void* falpha, int32_t* flen, double* ft0, void* fx, int threads) {
  const Problem<S>& p = Get<S>(op);
  const int T = p.T, n = p.n, m = p.m;
#pragma omp parallel for num_threads(threads) schedule(dynamic) if (threads > 1)
  for (int b = 0; b < batch; b++) {
    const S* xb = (const S*)x_init + size_t(b) * n;
    Plan<S> fin;
    Vec<S> xe;
    const auto recs = RecedingHorizonSimulate(p, Vec<S>(xb, xb + n), final_time, planner_runtime, extra_time,
                                              solve_time, use_al != 0, &fin, &xe, max_records);
    num_records[b] = int(recs.size());
    for (size_t r = 0; r < recs.size(); r++) {
      const size_t at = size_t(b) * max_records + r;
      t_call[at] = recs[r].t_call;
      plan_t0[at] = recs[r].plan_t0;
      first_step[at] = recs[r].first_step;
      iters[at] = recs[r].iters;
      ok[at] = recs[r].ok;
      converged[at] = recs[r].converged;
      max_bt[at] = recs[r].max_backtracks;
      std::memcpy((S*)x_measured + at * n, recs[r].x_measured.data(), sizeof(S) * n);
      std::memcpy((S*)x0 + at * n, recs[r].x0.data(), sizeof(S) * n);
      PackTraj(p, recs[r].solution.op, (S*)xs + at * T * n, (S*)us + at * T * m);
      PackStrategies(p, recs[r].solution.st, (S*)P + at * T * m * n, (S*)alpha + at * T * m);
    }
    StorePlan<S>(p, cap, fin, fxs, fus, fP, falpha, b);
    flen[b] = fin.len();
    ft0[b] = fin.t0;
    std::memcpy((S*)fx + size_t(b) * n, xe.data(), sizeof(S) * n);
  }
}


template <class S>
void NashBatch(const OracleProblem* op, int batch, const void* x0, const void* xs, const void* us, const void* P,
               const void* alpha, double max_perturbation, int open_loop, int euler, void* costs, int32_t* is_nash,
               void* margin, int threads) {
  const Problem<S>& p = Get<S>(op);
  const int T = p.T, n = p.n;
#pragma omp parallel for num_threads(threads) schedule(dynamic) if (threads > 1)
  for (int b = 0; b < batch; b++) {
    const Plan<S> pl = LoadPlan<S>(p, T, T, 0.0, xs, us, P, alpha, b);
    const S* xb = (const S*)x0 + size_t(b) * n;
    const Vec<S> x0v(xb, xb + n);
    if (costs) {
      const Vec<S> c = ComputeStrategyCosts(p, x0v, pl.op, pl.st, open_loop != 0, euler != 0);
      std::memcpy((S*)costs + size_t(b) * p.N, c.data(), sizeof(S) * p.N);
    }
    if (is_nash) {
      S mg;
      is_nash[b] = CheckLocalNash(p, x0v, pl.op, pl.st, S(max_perturbation), open_loop != 0, &mg) ? 1 : 0;
      if (margin) ((S*)margin)[b] = mg;
    }
  }
}

// Diagnosis: ILQSolver::Solve of ONE instance (zero warm start) with every CheckArmijoCondition call recorded:
// out [max_entries][8] doubles = (iteration, backtrack, accepted, step, last_merit, merit, expected_decrease, scaled).
// Returns the number of calls made (entries beyond max_entries are dropped); *ok_out = the solve's success flag.
template <class S>
int ArmijoTraceOne(const OracleProblem* op, const void* x0, int max_entries, double* out, int* ok_out, int* iters_out) {
  const Problem<S>& p = Get<S>(op);
  Trajectory<S> tr;
  Strategies<S> st;
  std::vector<S> zx(size_t(p.T) * p.n, S(0)), zu(size_t(p.T) * p.m, S(0)), zP(size_t(p.T) * p.m * p.n, S(0)), za(size_t(p.T) * p.m, S(0));
  UnpackTraj(p, p.T, zx.data(), zu.data(), &tr);
  UnpackStrategies(p, p.T, zP.data(), za.data(), &st);
  Vec<S> x0v((const S*)x0, (const S*)x0 + p.n);
  ILQState<S> state;
  ALState<S> al(p.num_constraints, p.T, p.dt);
  std::vector<ArmijoTrace<S>> trace;
  Vec<S> fc;
  int it = 0, conv = 0;
  const bool ok = SolveILQ(p, x0v, &tr, &st, &state, &al, 0, (std::vector<IterLog<S>>*)nullptr, &fc, &it, &conv,
                           (Strategies<S>*)nullptr, (int*)nullptr, (const S*)nullptr, &trace);
  for (size_t q = 0; q < trace.size() && (int)q < max_entries; q++) {
    const ArmijoTrace<S>& t = trace[q];
    double* o = out + 8 * q;
    o[0] = t.iteration; o[1] = t.backtrack; o[2] = t.accepted; o[3] = double(t.step);
    o[4] = double(t.last_merit); o[5] = double(t.merit); o[6] = double(t.expected_decrease); o[7] = double(t.scaled);
  }
  if (ok_out) *ok_out = ok ? 1 : 0;
  if (iters_out) *iters_out = it;
  return int(trace.size());
}
extern "C" {

int oracle_lq_feedback(const ilqg_dims* d, const void* A, const void* Bm, const void* Q, const void* l,
                       const void* R, const void* r, const ilqg_pair* pairs, int npairs, const void* x0, void* P,
                       void* alpha, void* dx, void* costates, int threads) {
  if (d->dtype == ILQG_F32)
    return LQBatch<float>(d, A, Bm, Q, l, R, r, pairs, npairs, x0, P, alpha, dx, costates, 0, threads);
  return LQBatch<double>(d, A, Bm, Q, l, R, r, pairs, npairs, x0, P, alpha, dx, costates, 0, threads);
}

int oracle_lq_openloop(const ilqg_dims* d, const void* A, const void* Bm, const void* Q, const void* l,
                       const void* R, const void* r, const ilqg_pair* pairs, int npairs, const void* x0, void* P,
                       void* alpha, void* dx, void* costates, int threads) {
  if (d->dtype == ILQG_F32)
    return LQBatch<float>(d, A, Bm, Q, l, R, r, pairs, npairs, x0, P, alpha, dx, costates, 1, threads);
  return LQBatch<double>(d, A, Bm, Q, l, R, r, pairs, npairs, x0, P, alpha, dx, costates, 1, threads);
}

void* oracle_problem_create(const ilqg_problem_desc* desc) {
  auto* op = new OracleProblem;
  op->terms.assign(desc->terms, desc->terms + desc->num_terms);
  op->poly_off.assign(desc->polyline_offsets, desc->polyline_offsets + desc->num_polylines + 1);
  const int npts = desc->num_polylines ? desc->polyline_offsets[desc->num_polylines] : 0;
  op->poly_pts.assign(desc->polyline_points, desc->polyline_points + 2 * npts);
  op->desc = *desc;
  op->dense.assign(desc->dense_params, desc->dense_params + (desc->dense_params ? desc->num_dense_params : 0));
  op->desc.dense_params = op->dense.data();
  op->desc.terms = op->terms.data();
  op->desc.polyline_offsets = op->poly_off.data();
  op->desc.polyline_points = op->poly_pts.data();
  op->pf.reset(new Problem<float>(op->desc));
  op->pd.reset(new Problem<double>(op->desc));
  return op;
}
void oracle_problem_destroy(void* h) { delete (OracleProblem*)h; }

int oracle_problem_pairs(void* h, ilqg_pair* pairs, int* npairs) {
  const auto* op = (OracleProblem*)h;
  *npairs = (int)op->pd->pairs.size();
  for (int q = 0; q < *npairs; q++) pairs[q] = op->pd->pairs[q];
  return 0;
}
int oracle_problem_num_constraints(void* h) { return ((OracleProblem*)h)->pd->num_constraints; }

#define DISPATCH(dtype, fn, ...)        \
  do {                                  \
    if ((dtype) == ILQG_F32)            \
      fn<float>(__VA_ARGS__);           \
    else                                \
      fn<double>(__VA_ARGS__);          \
  } while (0)

void oracle_rollout(void* h, int dtype, int batch, const void* x0, const void* xs_ref, const void* us_ref,
                    const void* P, const void* alpha, const void* alpha_scale, void* xs, void* us) {
  DISPATCH(dtype, RolloutBatch, (OracleProblem*)h, batch, x0, xs_ref, us_ref, P, alpha, alpha_scale, xs, us);
}
void oracle_linearize(void* h, int dtype, int batch, const void* xs, const void* us, void* A, void* Bm) {
  DISPATCH(dtype, LinearizeBatch, (OracleProblem*)h, batch, xs, us, A, Bm);
}
void oracle_quadraticize(void* h, int dtype, int batch, const void* xs, const void* us, const void* lambdas,
                         const void* mu, const int32_t* t_extreme, void* Q, void* l, void* R, void* r) {
  DISPATCH(dtype, QuadraticizeBatch, (OracleProblem*)h, batch, xs, us, lambdas, mu, t_extreme, Q, l, R, r);
}
void oracle_total_costs(void* h, int dtype, int batch, const void* xs, const void* us, void* costs,
                        int32_t* t_extreme) {
  DISPATCH(dtype, TotalCostsBatch, (OracleProblem*)h, batch, xs, us, costs, t_extreme);
}
void oracle_ilq_solve(void* h, int dtype, int batch, const void* x0, void* xs, void* us, void* P, void* alpha,
                      void* costs, int32_t* iters, int32_t* status, int32_t* converged, int fixed_iters,
                      void* rawP, void* rawAlpha, void* merit_log, int merit_log_len, int threads) {
  DISPATCH(dtype, SolveBatch, (OracleProblem*)h, batch, x0, xs, us, P, alpha, costs, iters, status, converged,
           fixed_iters, rawP, rawAlpha, merit_log, merit_log_len, threads);
}

int oracle_armijo_trace(void* h, int dtype, const void* x0, int max_entries, double* out, int* ok_out, int* iters_out) {
  return dtype == 0 ? ArmijoTraceOne<float>((OracleProblem*)h, x0, max_entries, out, ok_out, iters_out)
                    : ArmijoTraceOne<double>((OracleProblem*)h, x0, max_entries, out, ok_out, iters_out);
}

// The same with per-iteration step sizes supplied by the caller ([batch][fixed_iters]) instead of the line search.
void oracle_ilq_solve_forced(void* h, int dtype, int batch, const void* x0, void* xs, void* us, void* P, void* alpha,
                             void* costs, int32_t* iters, int32_t* status, int32_t* converged, int fixed_iters,
                             const void* forced_steps, void* rawP, void* rawAlpha, void* merit_log, int merit_log_len) {
  DISPATCH(dtype, SolveBatch, (OracleProblem*)h, batch, x0, xs, us, P, alpha, costs, iters, status, converged,
           fixed_iters, rawP, rawAlpha, merit_log, merit_log_len, 1, forced_steps);
}

// Problem::SetUpNextRecedingHorizon for a batch of plans sharing one time base.
void oracle_receding_horizon_shift(void* h, int dtype, int batch, const void* x0, double t0, double planner_runtime,
                                   double plan_t0, void* xs, void* us, void* P, void* alpha, void* x0_next,
                                   int32_t* first_step, double* new_plan_t0) {
  DISPATCH(dtype, RecedingHorizonBatch, (OracleProblem*)h, batch, x0, t0, planner_runtime, plan_t0, xs, us, P, alpha,
           x0_next, first_step, new_plan_t0);
}


// MultiPlayerIntegrableSystem::Integrate(t0, t, x0, operating_point, strategies) under per-instance plans;
// instances whose plan does not contain `must_contain` (SolutionSplicer::ContainsTime) are deactivated instead.
void oracle_plan_integrate(void* h, int dtype, int batch, int cap, const void* xs, const void* us, const void* P,
                           const void* alpha, const int32_t* len, const double* t0, double t_from, double t_to,
                           double must_contain, void* x, int32_t* active) {
  DISPATCH(dtype, PlanIntegrateBatch, (OracleProblem*)h, batch, cap, xs, us, P, alpha, len, t0, t_from, t_to,
           must_contain, x, active);
}
// Problem::OverwriteSolution(plan) + SetUpNextRecedingHorizon(x, t, planner_runtime) per instance.
void oracle_receding_horizon_sync(void* h, int dtype, int batch, int cap, const void* pxs, const void* pus,
                                  const void* pP, const void* palpha, const int32_t* len, const double* t0, const void* x,
                                  double t, double planner_runtime, void* xs, void* us, void* P, void* alpha,
                                  void* x0_next, double* solve_t0, int32_t* first_step, int32_t* active) {
  DISPATCH(dtype, RecedingSyncBatch, (OracleProblem*)h, batch, cap, pxs, pus, pP, palpha, len, t0, x, t, planner_runtime,
           xs, us, P, alpha, x0_next, solve_t0, first_step, active);
}
// SolutionSplicer: construction (len == 0) or Splice of a converged solution.
void oracle_solution_splice(void* h, int dtype, int batch, int cap, void* pxs, void* pus, void* pP, void* palpha,
                            int32_t* len, double* t0, const void* xs, const void* us, const void* P, const void* alpha,
                            const double* solve_t0, const int32_t* converged, const int32_t* active) {
  DISPATCH(dtype, SpliceBatch, (OracleProblem*)h, batch, cap, pxs, pus, pP, palpha, len, t0, xs, us, P, alpha, solve_t0,
           converged, active);
}
void oracle_solve_resume(void* h, int dtype, int batch, const void* x0, void* xs, void* us, void* P, void* alpha,
                         void* costs, int32_t* iters, int32_t* status, int32_t* converged, int al_mode,
                         void* last_merit, int threads) {
  DISPATCH(dtype, SolveResumeBatch, (OracleProblem*)h, batch, x0, xs, us, P, alpha, costs, iters, status, converged,
           al_mode, last_merit, threads);
}
void oracle_receding_horizon_simulate(void* h, int dtype, int batch, const void* x_init, double final_time,
                                      double planner_runtime, double extra_time, double solve_time, int use_al,
                                      int max_records, int32_t* num_records, double* t_call, void* x_measured, void* x0,
                                      double* plan_t0, int32_t* first_step, void* xs, void* us, void* P, void* alpha,
                                      int32_t* iters, int32_t* ok, int32_t* converged, int32_t* max_bt, int cap, void* fxs,
                                      void* fus, void* fP, void* falpha, int32_t* flen, double* ft0, void* fx,
                                      int threads) {
  DISPATCH(dtype, SimulateBatch, (OracleProblem*)h, batch, x_init, final_time, planner_runtime, extra_time, solve_time,
           use_al, max_records, num_records, t_call, x_measured, x0, plan_t0, first_step, xs, us, P, alpha, iters, ok,
           converged, max_bt, cap, fxs, fus, fP, falpha, flen, ft0, fx, threads);
}


// ComputeStrategyCosts (costs != NULL) and / or NumericalCheckLocalNashEquilibrium (is_nash != NULL) per instance.
void oracle_nash(void* h, int dtype, int batch, const void* x0, const void* xs, const void* us, const void* P,
                 const void* alpha, double max_perturbation, int open_loop, int euler, void* costs, int32_t* is_nash,
                 void* margin, int threads) {
  DISPATCH(dtype, NashBatch, (OracleProblem*)h, batch, x0, xs, us, P, alpha, max_perturbation, open_loop, euler, costs,
           is_nash, margin, threads);
}


// CheckSufficientLocalNashEquilibrium per instance: ok [B], worst eigenvalue [B] (double).
void oracle_sufficient_nash(void* h, int dtype, int batch, const void* xs, const void* us, int32_t* ok, double* worst) {
  const auto* op = (OracleProblem*)h;
  for (int b = 0; b < batch; b++) {
    if (dtype == ILQG_F32) {
      const auto& p = *op->pf;
      Trajectory<float> tr;
      UnpackTraj(p, p.T, (const float*)xs + size_t(b) * p.T * p.n, (const float*)us + size_t(b) * p.T * p.m, &tr);
      float w;
      ok[b] = CheckSufficientLocalNash(p, tr, &w) ? 1 : 0;
      worst[b] = w;
    } else {
      const auto& p = *op->pd;
      Trajectory<double> tr;
      UnpackTraj(p, p.T, (const double*)xs + size_t(b) * p.T * p.n, (const double*)us + size_t(b) * p.T * p.m, &tr);
      double w;
      ok[b] = CheckSufficientLocalNash(p, tr, &w) ? 1 : 0;
      worst[b] = w;
    }
  }
}
// Smallest eigenvalue of a symmetric n x n matrix (column-major doubles) — exposed so the tests can pin the Jacobi
// routine against numpy.linalg.eigvalsh.
double oracle_min_eigenvalue(int n, const double* a) {
  Mat<double> m(n, n);
  for (int i = 0; i < n * n; i++) m.d[i] = a[i];
  return MinEigenvalueSymmetric(m);
}

// xdot = f(x, u) and one Integrate step (double I/O regardless of dtype, for the
// finite-difference tests that restate test/test_linearization.cpp).
void oracle_dynamics(void* h, int dtype, const double* x, const double* u, double* xdot, double* xnext, int euler) {
  const auto* op = (OracleProblem*)h;
  if (dtype == ILQG_F32) {
    const auto& p = *op->pf;
    Vec<float> xv(x, x + p.n), uv(u, u + p.m);
    const Vec<float> f = Evaluate(p, xv, uv);
    const Vec<float> nx = Integrate(p, 0.0, p.dt, xv, uv, euler != 0);
    for (int i = 0; i < p.n; i++) { xdot[i] = f[i]; xnext[i] = nx[i]; }
  } else {
    const auto& p = *op->pd;
    Vec<double> xv(x, x + p.n), uv(u, u + p.m);
    const Vec<double> f = Evaluate(p, xv, uv);
    const Vec<double> nx = Integrate(p, 0.0, p.dt, xv, uv, euler != 0);
    for (int i = 0; i < p.n; i++) { xdot[i] = f[i]; xnext[i] = nx[i]; }
  }
}

// PlayerCost::Evaluate at one (x, u) in double (restating test_quadraticization.cpp's
// numerical-derivative checks needs cost values).  If include_constraints, adds the
// augmented-Lagrangian value of every constraint of the player
// (Constraint::EvaluateAugmentedLagrangian, constraint.h:83-87) with lambda/mu given.
double oracle_player_value(void* h, int player, const double* x, const double* u, int include_constraints,
                           double lambda, double mu, int step) {
  const auto& p = *((OracleProblem*)h)->pd;
  Vec<double> xv(x, x + p.n), uv(u, u + p.m);
  // step < 0: "late enough for every FinalTimeCost"; otherwise the time step the costs are evaluated at
  double v = step < 0 ? EvaluatePlayer(p, player, xv, uv) : EvaluatePlayer(p, player, xv, uv, step, step);
  if (include_constraints)
    for (size_t ti = 0; ti < p.terms.size(); ti++) {
      const auto& t = p.terms[ti];
      if (t.player != player) continue;
      double g;
      if (t.role == ILQG_ROLE_STATE_CONSTRAINT)
        g = EvaluateTerm(p, (int)ti, xv.data(), p.n);
      else if (t.role == ILQG_ROLE_CONTROL_CONSTRAINT)
        g = EvaluateTerm(p, (int)ti, &uv[p.uoff[t.arg]], p.udim(t.arg));
      else
        continue;
      v += lambda * g + 0.5 * ConstraintMu(lambda, g, mu, (t.flags & ILQG_FLAG_EQUALITY) != 0) * g * g;
    }
  return v;
}

// Known-answer access to the geometry (test/test_line_segment2.cpp, test_polyline2.cpp).
// out = {closest.x, closest.y, signed_squared_distance, is_vertex, is_endpoint, seg.p1x, seg.p1y, seg.p2x, seg.p2y}
void oracle_polyline_closest_point(int dtype, const float* pts, int npts, double qx, double qy, double* out) {
  if (dtype == ILQG_F32) {
    Polyline2<float> pl(pts, npts);
    float cx, cy, ssd;
    bool v, e;
    Segment2<float> s;
    pl.ClosestPoint((float)qx, (float)qy, &cx, &cy, &v, &s, &ssd, &e);
    const double o[9] = {cx, cy, ssd, double(v), double(e), s.p1x, s.p1y, s.p2x, s.p2y};
    std::memcpy(out, o, sizeof(o));
  } else {
    Polyline2<double> pl(pts, npts);
    double cx, cy, ssd;
    bool v, e;
    Segment2<double> s;
    pl.ClosestPoint(qx, qy, &cx, &cy, &v, &s, &ssd, &e);
    const double o[9] = {cx, cy, ssd, double(v), double(e), s.p1x, s.p1y, s.p2x, s.p2y};
    std::memcpy(out, o, sizeof(o));
  }
}
// out = {closest.x, closest.y, signed_squared_distance, is_endpoint, side}
void oracle_segment_closest_point(const float* p1p2, double qx, double qy, double* out) {
  Segment2<float> s(p1p2[0], p1p2[1], p1p2[2], p1p2[3]);
  float cx, cy, ssd;
  bool e;
  s.ClosestPoint((float)qx, (float)qy, &cx, &cy, &e, &ssd);
  out[0] = cx;
  out[1] = cy;
  out[2] = ssd;
  out[3] = e;
  out[4] = s.Side((float)qx, (float)qy);
}

}  // extern "C"
