// ilqg_solve.hpp — the iterative-LQ loop of one game instance, as three device parts that the host
// launches in rounds (gfx950).
//
// ILQSolver::Solve (src/ilq_solver.cpp:76-172) with ModifyLQStrategies (:289-348) inlined:
// rollout -> total costs -> quadraticise, then
//   { LQ sweep (+ expected decrease) -> scaled rollout -> Armijo back-tracking on the
//     KKT-residual merit -> total costs } until converged / max_solver_iters / failure,
// and AugmentedLagrangianSolver::Solve (src/augmented_lagrangian_solver.cpp:72-210) around it.
//   trial part  rollout (initial, or a line-search trial) pipelined with linearise + quadraticise over the
//               waves of the workgroup, reductions, Armijo decision — loops until a trial is accepted;
//   LQ part     the Riccati sweep at the accepted operating point;
//   exit part   the return path of Solve and the AL bookkeeping.
// Every instance carries its own loop state (SolveState) in the workspace, so instances finish after
// different numbers of iterations and back-tracks without masks or compaction: a part returns at once
// for an instance that is not waiting for it.
#pragma once
#ifndef ILQG_ROLL_PRIO
#define ILQG_ROLL_PRIO 1  // the integrating wave of a trial pass issues ahead of the row waves it shares SIMDs with (fp32 headline +1.7 %, fp64 +0.3 %)
#endif

#include "ilqg_lq.hpp"
#include "ilqg_lq_openloop.hpp"
#include "ilqg_lq_feedback2.hpp"
#include "ilqg_lq_feedback1w.hpp"
#include "ilqg_lq_generic.hpp"
#include "ilqg_stages.hpp"

namespace ilqg {

template <typename T>
struct SolveArgs {
  const T* x0;          // [B][n]
  T *xs, *us;           // [B][T][n], [B][T][m]   in: warm start, out: result (buffer 0)
  T *P, *alpha;         // [B][T][m*n], [B][T][m] in: warm start, out: result (buffer 0)
  T* total_costs;       // [B][N]
  int *iters, *status, *converged;
  const int* active;  // [B] or null: instances to skip (receding-horizon harness: dropped out of the loop)
  T* ws;                // workspace, ws_stride elements per instance
  size_t ws_stride;
  int fixed_iters;
  int batch;
  int ol_row;           // elements per open-loop scratch row (0 when the feedback sweep is used)
  int al_mode;          // 1: AugmentedLagrangianSolver::Solve around the inner iLQ solve
  ilqg_solver_params prm;
  long long* prof;      // optional [B][16] shader-clock cycles (diagnostics) or nullptr
  int first;            // trial kernel: 1 on the first launch of a solve (initialises the state)
  int* unfinished;      // [4] instances left waiting for: an LQ sweep, the exit path, a restart (AL), another pass
  // split passes only: the instances of this round (null: all of the batch, in order), where the decision kernel
  // lists the ones that need another pass, and the rows of one instance a workgroup of the row kernel takes
  const int* ids;
  int* ids_next;
  int round_count;      // instances this round's split kernels cover (the batch, or the length of `ids`)
  const T* forced_steps;  // [B][fixed_iters] or null: test mode, iteration q of instance b takes this step, no Armijo test
  int defer_forward;    // 1: the sweep's forward pass / ExpectedDecrease runs in the next trial pass, beside the rollout
  int prio_div;         // > 0: the batch is resident at once on this many CUs: the kernels rotate their wave priorities so
                        //      that the instances sharing a CU finish together (ilqg_lq.hpp)
  int compact;          // 1: [Q | l | R | r] travel from the row stage to the sweep as compact rows (ilqg_rows.hpp), kept
                        //    where the dense Q array would be
  int rows_cw;          // rows per chunk of the row stage (ilqg_rows.hpp): 64, 32 or 16, chosen by the launcher
  T* probe_pool;        // speculative line search: [listed instances][probe_k] entries of ProbeEntry::total elements
  int probe_k;          // step sizes probed per instance this round (<= kProbeCandidates)
  int outer_closed;     // 1: the augmented-Lagrangian outer loop is out of time (max_runtime): an inner solve that ends
                        //    now ends the instance's solve instead of starting another (augmented_lagrangian_solver.cpp:107-110)
  int clear_counters;   // 1: the sweep kernel's first workgroup clears the round counters — nothing counts between a
                        //    round's sweep and the next round's first kernel, and the next round then needs no fill
                        //    command in front of it (~5 us of every round of a free-running solve)
};


// Loop state of one instance's ILQSolver::Solve (the locals of src/ilq_solver.cpp:76-172 plus the
// AugmentedLagrangianSolver bookkeeping).  It lives in registers inside a kernel and in the
// workspace between the trial kernel and the LQ kernel.
// ST_PROBE: ST_ROLLOUT for an instance whose last probe rejected every candidate — it skips the regular pass of
// its round (which would try one more step size at the price of a rollout) and is probed again in the next one.
enum { ST_ROLLOUT = 0, ST_QUAD = 1, ST_LQ = 2, ST_INNER_DONE = 3, ST_DONE = 4, ST_PROBE = 5 };
enum { Q_COSTS = 0, Q_INIT = 1, Q_TRIAL = 2, Q_LIN = 3 };
template <typename T>
struct SolveState {
  int stage, qmode, initial, cur, sacc, num_iterations, bt, accepted_iters;
  int has_converged, ok, logged, inner_calls, al_success;
  int ed_pending;  // the sweep left its scratch rows: ExpectedDecrease is formed by the next trial pass (forward pass)
  int rejected;    // steps the line searches of this solve have rejected so far (diagnostics: mean back-tracks)
  int pad2;
  T acc_scale, step, last_merit, expected_decrease, max_err, mu;
};
constexpr int kStateElems = 32;  // >= sizeof(SolveState<T>) / sizeof(T) for float and double
static_assert(sizeof(SolveState<float>) <= kStateElems * sizeof(float), "state slot");
static_assert(sizeof(SolveState<double>) <= kStateElems * sizeof(double), "state slot");

// Per-instance workspace layout (in elements of T).
struct WsLayout {
  size_t xs1, us1, P1, al1, A, B, Q, l, R, r, lqscr, dx, mpart, cpart, ints, state, lambdas, wxs, wus, wP, wal, total;
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
    state = take(kStateElems);    // SolveState carried between the two kernels of a solve
    lambdas = take(size_t(num_constraints) * T);  // per-instance Constraint::lambdas_ (constraint.h:136)
    wxs = take(al_mode ? size_t(T) * n : 0);      // Problem's stored solution = warm start of the next inner solve
    wus = take(al_mode ? size_t(T) * m : 0);
    wP = take(al_mode ? size_t(T) * m * n : 0);
    wal = take(al_mode ? size_t(T) * m : 0);
    total = o;
  }
};

// One solve is a sequence of two kernels that alternate until every instance is done:
//
//   trial kernel  (W wavefronts per instance)  rollout of the warm start or of a line-search trial,
//                 linearise + quadraticise + merit/cost pieces of that trajectory, Armijo decision —
//                 repeated inside the kernel until a trial is accepted or the search gives up;
//   LQ kernel     (one workgroup per instance)  the coupled Riccati sweep at the accepted operating
//                 point + ExpectedDecrease, and the return path of ILQSolver::Solve / the
//                 AugmentedLagrangianSolver outer loop.
//
// They are separate kernels because their register needs differ by 2x (the MFMA sweep wants the whole
// 512-entry file of a SIMD, the trial stages fit in a quarter of it): fused, the sweep's allocation
// caps the kernel at one wave per SIMD and leaves nothing to run beside the serial rollout chain.
// Split, the trial kernel runs the rollout on wave 0 while the other waves of the workgroup already
// linearise / quadraticise the steps it has produced (the steps are independent of each other).
// The state is wave-uniform; pinning every field to the scalar register file keeps ~26 vector registers
// free across the kernel's tightest loops (a uniform value loaded through the vector memory path would
// otherwise sit in a VGPR for the whole kernel).
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ double uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <typename T>
__device__ __forceinline__ SolveState<T> state_load(const T* w, const WsLayout& L) {
  SolveState<T> s = *reinterpret_cast<const SolveState<T>*>(w + L.state);
  s.stage = uniform(s.stage); s.qmode = uniform(s.qmode); s.initial = uniform(s.initial); s.cur = uniform(s.cur);
  s.sacc = uniform(s.sacc); s.num_iterations = uniform(s.num_iterations); s.bt = uniform(s.bt);
  s.accepted_iters = uniform(s.accepted_iters); s.has_converged = uniform(s.has_converged); s.ok = uniform(s.ok);
  s.logged = uniform(s.logged); s.inner_calls = uniform(s.inner_calls); s.al_success = uniform(s.al_success);
  s.ed_pending = uniform(s.ed_pending);
  s.rejected = uniform(s.rejected);
  s.pad2 = 0;
  s.acc_scale = uniform(s.acc_scale); s.step = uniform(s.step); s.last_merit = uniform(s.last_merit);
  s.expected_decrease = uniform(s.expected_decrease); s.max_err = uniform(s.max_err); s.mu = uniform(s.mu);
  return s;
}
template <typename T>
__device__ __forceinline__ void state_store(T* w, const WsLayout& L, const SolveState<T>& s) {
  if (threadIdx.x == 0) *reinterpret_cast<SolveState<T>*>(w + L.state) = s;
}

template <typename T>
struct InstanceBuffers {
  T *w, *xs0, *us0, *P0, *al0;
  const WsLayout L;
  __device__ InstanceBuffers(const DevProblem& p, const SolveArgs<T>& sa, int b)
      : L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode) {
    w = sa.ws + size_t(b) * sa.ws_stride;
    // two operating-point buffers and two strategy buffers; buffer 0 is the caller's
    xs0 = sa.xs + size_t(b) * p.T * p.n;
    us0 = sa.us + size_t(b) * p.T * p.m;
    P0 = sa.P + size_t(b) * p.T * p.m * p.n;
    al0 = sa.alpha + size_t(b) * p.T * p.m;
  }
  __device__ T* XS(int i) const { return i ? w + L.xs1 : xs0; }
  __device__ T* US(int i) const { return i ? w + L.us1 : us0; }
  __device__ T* PB(int i) const { return i ? w + L.P1 : P0; }
  __device__ T* AL(int i) const { return i ? w + L.al1 : al0; }
  __device__ int* t_extreme() const { return reinterpret_cast<int*>(w + L.ints); }
};

template <typename T>
__device__ __forceinline__ int solve_max_iters(const SolveArgs<T>& sa) {
  // AugmentedLagrangianSolver builds its inner ILQSolver with unconstrained_solver_max_iters
  // (augmented_lagrangian_solver.h:80-84)
  return sa.fixed_iters > 0 ? sa.fixed_iters
                            : (sa.al_mode ? sa.prm.unconstrained_solver_max_iters : sa.prm.max_solver_iters);
}

// Row waves of a fused trial kernel of W wavefronts: wave 0 integrates, the others take the chunks of rows as they
// become ready (a single-wave workgroup does both, one after the other).
__host__ __device__ inline int trial_row_waves(int waves) { return waves > 1 ? waves - 1 : 1; }
// scratch of one row wave (ilqg_rows.hpp, `cw` rows per chunk); also holds the per-row partials of the reductions
// between passes
// (`xreg`: the chunk keeps its (x, u) rows in registers — the statically specialised row stage — and has no LDS image)
__host__ __device__ inline size_t trial_rows_elems(const DevProblem& p, int cw, bool xreg = false) {
  size_t e = xreg ? rows_lds_elems_xreg(p.rp_pslots, p.rp_lslots, cw) : rows_lds_elems(p.n, p.m, p.rp_pslots, p.rp_lslots, cw);
  const size_t red = size_t(p.T) * p.N * 2 + 8;
  if (e < red) e = red;
  return (e + 3) & ~size_t(3);
}
// scratch of a workgroup of the split row kernels (ilq_rows_kernel, ilq_probe_rows_kernel): one chunk, the state rows in
// registers where the shape allows it (`nx`: the instantiation's compile-time n, 0 for the run-time-dimensioned kernels)
__host__ __device__ inline size_t split_rows_elems(const DevProblem& p, int nx, int cw) {
  if (!rows_state_in_registers(nx, p.m)) return trial_rows_elems(p, cw);
  return (rows_lds_elems_xreg(p.rp_pslots, p.rp_lslots, cw) + 3) & ~size_t(3);
}
// ... and of a workgroup of the probing row kernel, which evaluates the merit function only (gradient slots)
__host__ __device__ inline size_t probe_rows_elems(const DevProblem& p, int nx, int cw) {
  const size_t image = rows_state_in_registers(nx, p.m) ? 0 : size_t(p.n + p.m) * cw;
  return (image + rows_lds_elems_grad(p.rp_pslots, p.rp_gslots, cw) + 3) & ~size_t(3);
}
// LDS of the trial kernel: [word maps | rollout scratch | row waves x row scratch | 4 ints]
template <typename T>
__host__ __device__ inline size_t trial_lds_bytes(const DevProblem& p, int waves, int cw, bool xreg = false) {
  const size_t re = (rollout_lds_elems(p.n, p.m) + 3) & ~size_t(3);
  return rows_maps_bytes(p) + (re + size_t(trial_row_waves(waves)) * trial_rows_elems(p, cw, xreg)) * sizeof(T) + 16;
}

// ---------------------------------------------------------------------------
// Return path of one ILQSolver::Solve call (stage INNER_DONE): final iterate back into the caller's
// buffers, and — in AL mode — the AugmentedLagrangianSolver bookkeeping that may start the next inner
// solve.  Executed by every thread of the trial kernel's workgroup.
// ---------------------------------------------------------------------------
template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void solve_exit_path(const DevProblem& p, const QuadTables<T>& tb, const SolveArgs<T>& sa,
                                                const InstanceBuffers<T>& ib, SolveState<T>& s, int b, T* sm) {
  const int n = NX > 0 ? NX : p.n, m = NX > 0 ? NP * MU : p.m;  // NX = 0: the run-time-dimensioned path
  const int Tn = p.T;
  const ilqg_solver_params& prm = sa.prm;
  const WsLayout& L = ib.L;
  T* const w = ib.w;
  T *const xs0 = ib.xs0, *const us0 = ib.us0, *const P0 = ib.P0, *const al0 = ib.al0;
  T* const lambdas = w + L.lambdas;
  const int t = threadIdx.x;
  // Copies with eight loads in flight per lane: a load-store pair per trip is one memory latency per element and lane,
  // and for the few dozen plans of a receding-horizon replan this kernel is a link of every round's chain (the
  // strategies alone are T m n words; config 5 as written: 90 -> 25 us per launch).
  auto copy = [&](T* dst, const T* src, int count, T scale, bool scaled) {
    constexpr int U = 8;
    const int nt = blockDim.x;
    for (int e0 = t; e0 < count; e0 += nt * U) {
      T v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * nt;
        v[u] = e < count ? src[e] : T(0);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * nt;
        if (e < count) dst[e] = scaled ? v[u] * scale : v[u];
      }
    }
  };
  __syncthreads();
      // ---- the log's final iterate goes back through buffer 0 (alpha carries the accepted step) ----
      if (s.cur == 1) {
        copy(xs0, w + L.xs1, Tn * n, T(1), false);
        copy(us0, w + L.us1, Tn * m, T(1), false);
      }
      if (s.sacc == 1) copy(P0, w + L.P1, Tn * m * n, T(1), false);
      copy(al0, ib.AL(s.sacc), Tn * m, s.acc_scale, true);
      __syncthreads();
      s.stage = ST_DONE;
      if (sa.al_mode) {  // AugmentedLagrangianSolver::Solve, src/augmented_lagrangian_solver.cpp:72-210
        s.logged += 1 + s.accepted_iters;  // SolverLog entries of this inner call (:94, :185)
        s.al_success = s.al_success && s.ok;
        if (s.inner_calls > 0 && !s.ok) {  // :166-178
          for (int e = t; e < p.num_constraints * Tn; e += blockDim.x)
            lambdas[e] *= T(prm.geometric_lambda_downscaling);
          s.mu *= T(prm.geometric_mu_downscaling);
          __syncthreads();
        }
        s.inner_calls++;
        if (p.num_constraints > 0 && s.logged < prm.max_solver_iters &&
            s.max_err > T(prm.constraint_error_tolerance) && !sa.outer_closed) {
          // ---- multiplier update at the final operating point (:116-140) ----
          T my_err = -dinf<T>();
          // A thread per (constraint slot, time step), in strides: the errors are independent of one another, and a
          // multiplier is touched by the steps whose TimeIndex is its own — a run of consecutive steps (the index is
          // monotone in k; the reference's float product makes a step now and then land on its predecessor's), which
          // the first step of the run applies in the reference's order.  (One thread per slot walking the steps was a
          // chain of T global round trips: most of this kernel's 135 us in config 5's replans.)
          auto time_index = [&](int k) {  // Constraint::IncrementLambda (constraint.h:98-102) at TimeIndex(t0 + dt*float(k))
            const double tt = 0.0 + p.dt * double(float(k));
            return int(static_cast<size_t>(tt / p.dt));
          };
          for (int e = t; e < p.num_constraints * Tn; e += blockDim.x) {
            const int cs = e / Tn, k = e - cs * Tn;
            int ti = 0;
            for (int q = 0; q < p.num_terms; q++)
              if (tb.terms[q].slot == cs) ti = q;
            const DevTerm c = tb.terms[ti];
            const bool on_state = c.role == ILQG_ROLE_STATE_CONSTRAINT;
            auto error_at = [&](int kk) {
              const T* v = on_state ? xs0 + size_t(kk) * n : us0 + size_t(kk) * m + p.uoff[c.arg];
              // FinalTimeConstraint::Evaluate (constraint/final_time_constraint.h:66-70): 0 before its threshold
              return kk < c.k_start ? T(0) : term_evaluate_leaf<T>(tb, ti, v, c.arg_dim);
            };
            const T err = error_at(k);
            my_err = err > my_err ? err : my_err;
            const int tidx = time_index(k);
            if (k > 0 && time_index(k - 1) == tidx) continue;  // applied by the first step of its run
            T lam = lambdas[cs * Tn + tidx];
            T step_err = err;
            for (int kk = k;;) {
              const T nl = lam + s.mu * step_err;
              lam = ((c.flags & ILQG_FLAG_EQUALITY) || nl > T(0)) ? nl : T(0);
              if (++kk >= Tn || time_index(kk) != tidx) break;
              step_err = error_at(kk);
            }
            lambdas[cs * Tn + tidx] = lam;
          }
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) {
            const T o = __shfl_xor(my_err, off, 64);
            my_err = o > my_err ? o : my_err;
          }
          if ((t & 63) == 0) sm[t >> 6] = my_err;  // (a workgroup of up to 64 waves)
          __syncthreads();
          s.max_err = sm[0];
          for (int q = 1; q < int(blockDim.x + 63) / 64; q++) s.max_err = sm[q] > s.max_err ? sm[q] : s.max_err;
          __syncthreads();
          s.mu *= T(prm.geometric_mu_scaling);  // :143
          // Problem::OverwriteSolution only after a successful inner solve (:151-154)
          if (s.ok) {
            copy(w + L.wxs, xs0, Tn * n, T(1), false);
            copy(w + L.wus, us0, Tn * m, T(1), false);
            copy(w + L.wP, P0, Tn * m * n, T(1), false);
            copy(w + L.wal, al0, Tn * m, T(1), false);
          } else {
            copy(xs0, w + L.wxs, Tn * n, T(1), false);
            copy(us0, w + L.wus, Tn * m, T(1), false);
            copy(P0, w + L.wP, Tn * m * n, T(1), false);
            copy(al0, w + L.wal, Tn * m, T(1), false);
          }
          __syncthreads();
          // ---- next ILQSolver::Solve call: fresh locals, persistent last_merit / t_extreme ----
          s.stage = ST_ROLLOUT;
          s.initial = 1;
          s.cur = 0;
          s.sacc = 0;
          s.acc_scale = T(1);
          s.num_iterations = 0;
          s.accepted_iters = 0;
          s.has_converged = 0;
          s.ok = 1;
        }
      }
      if (s.stage == ST_DONE) {
        if (sa.al_mode && p.num_constraints > 0 && s.max_err > T(prm.constraint_error_tolerance))
          s.al_success = 0;  // :188-191
        if (t == 0) {
          sa.iters[b] = sa.al_mode ? s.logged : s.num_iterations;
          sa.status[b] = (sa.al_mode ? s.al_success : s.ok) ? 1 : 0;
          sa.converged[b] = s.has_converged ? 1 : 0;
        }
      }
}

// Exit part: one workgroup per instance whose stage is INNER_DONE (its own small kernel, so that the
// augmented-Lagrangian bookkeeping does not sit in the register budget of the trial kernel).
template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void exit_part_instance(const DevProblem& p, const QuadTables<T>& tb, const SolveArgs<T>& sa,
                                                   int b, T* sm) {
  const InstanceBuffers<T> ib(p, sa, b);
  SolveState<T> s = state_load<T>(ib.w, ib.L);
  solve_exit_path<T, NX, NP, MU>(p, tb, sa, ib, s, b, sm);
  state_store<T>(ib.w, ib.L, s);
  if (threadIdx.x == 0 && s.stage == ST_ROLLOUT) atomicAdd(sa.unfinished + 2, 1);  // next inner solve (AL)
}

// ---------------------------------------------------------------------------
// Trial part.  Runs while the instance's stage is ROLLOUT or QUAD; W = wavefronts per instance.
//
// PHASE selects how much of one pass a kernel carries.  TRIAL_FUSED: everything, passes repeated inside the
// launch until the instance wants a sweep or its exit path (the default).  For problems whose linearise /
// quadraticise scratch leaves a CU with one or two instances (n = 24: 78 KB), the serial rollout of so few
// resident instances is what the launch waits for; there the pass is cut into three launches instead —
// TRIAL_ROLL (one wave per instance, LDS for the rollout only, so a CU holds many), the row kernel
// (rows_part_instance: one workgroup per block of rows, every row of the batch in flight) and TRIAL_DECIDE
// (reductions and the line-search decision) — and the host repeats them while instances ask for another pass.
// ---------------------------------------------------------------------------
enum { TRIAL_FUSED = 0, TRIAL_ROLL = 1, TRIAL_DECIDE = 2 };

// linearise / quadraticise scratch per wave of a trial-part kernel (TRIAL_DECIDE only needs the reductions' copy
// of the per-row partials, TRIAL_ROLL nothing)
template <typename T>
__host__ __device__ inline size_t trial_phase_quad_elems(const DevProblem& p, int phase, int cw, bool xreg = false) {
  if (phase == TRIAL_ROLL) return 0;
  if (phase == TRIAL_DECIDE) return (size_t(p.T) * p.N * 2 + 8 + 3) & ~size_t(3);
  return trial_rows_elems(p, cw, xreg);
}
template <typename T>
__host__ __device__ inline size_t trial_phase_lds_bytes(const DevProblem& p, int phase, int cw) {  // W = 1, no word maps
  // (the decision kernel integrates nothing: without the rollout's region a CU holds a wave slot's worth of its
  // workgroups — each one a chain of global round trips — instead of 18: B = 8192, fp64: 126 -> 120 us per launch)
  const size_t re = phase == TRIAL_DECIDE ? 0 : (rollout_lds_elems(p.n, p.m) + 3) & ~size_t(3);
  return (re + trial_phase_quad_elems<T>(p, phase, cw)) * sizeof(T) + 16;
}

// What the rows of a pass are asked for, from the instance's state (the pass's rollout, if any, is done).
template <typename T>
__device__ __forceinline__ QuadArgs<T> trial_quad_args(const DevProblem& p, const InstanceBuffers<T>& ib,
                                                       const SolveState<T>& s, bool compact) {
  const WsLayout& L = ib.L;
  T* const w = ib.w;
  const int qmode = s.qmode;
  const int at = (qmode == Q_COSTS || qmode == Q_INIT) ? s.cur : 1 - s.cur;
  QuadArgs<T> qa;
  qa.xs = ib.XS(at);
  qa.us = ib.US(at);
  qa.lambdas = p.num_constraints > 0 ? w + L.lambdas : nullptr;
  qa.mu = s.mu;
  qa.t_extreme = ib.t_extreme();
  qa.t_init = 0.0;
  const bool lin = qmode != Q_COSTS, quad = qmode == Q_INIT || qmode == Q_TRIAL;
  qa.A = (lin && !compact) ? w + L.A : nullptr;
  qa.Bm = (lin && !compact) ? w + L.B : nullptr;
  qa.Q = (quad && !compact) ? w + L.Q : nullptr;
  qa.l = (quad && !compact) ? w + L.l : nullptr;
  qa.R = (quad && !compact) ? w + L.R : nullptr;
  qa.r = (quad && !compact) ? w + L.r : nullptr;
  qa.compact = ((lin || quad) && compact) ? w + L.Q : nullptr;  // the dense Q array's space
  qa.compact_lin = lin && compact;
  qa.compact_quad = quad && compact;
  qa.merit_part = qmode == Q_TRIAL ? w + L.mpart : nullptr;
  qa.cost_part = (qmode == Q_COSTS || qmode == Q_TRIAL || qmode == Q_LIN) ? w + L.cpart : nullptr;
  qa.phacc = nullptr;
  return qa;
}

template <int ID> struct RowProgSel { typedef ProgStatic<ID> type; };
template <> struct RowProgSel<0> { typedef ProgDynamic type; };

// Row kernel of the split pass: one chunk of rows of instance b, one wave with its own scratch.
// PROGID: 0 = interpret the problem's row program; k = straight-line code for registered structure k (ilqg_rows.hpp)
template <typename T, int NX, int NP, int MU, int PROGID = 0>
__device__ __forceinline__ void rows_part_instance(const DevProblem& p, const short* maps, const SolveArgs<T>& sa,
                                                   int b, int chunk, T* sm) {
  const InstanceBuffers<T> ib(p, sa, b);
  const SolveState<T> s = state_load<T>(ib.w, ib.L);
  const QuadArgs<T> qa = trial_quad_args<T>(p, ib, s, sa.compact != 0);
  const int k0 = chunk * sa.rows_cw;
  const int nrows = p.T - k0 < sa.rows_cw ? p.T - k0 : sa.rows_cw;
  rows_chunk<T, NX, NP * MU, NP, rows_state_in_registers(NX, NP * MU), false, typename RowProgSel<PROGID>::type>(
      p, maps, qa, k0, nrows, sa.rows_cw, sm, int(threadIdx.x));
}

// ---------------------------------------------------------------------------
// Speculative line search (split passes).  A back-tracking instance tries its step sizes one pass at a time, and a
// pass of one instance costs the latency of a serial rollout however empty the chip is.  Given the sweep's
// strategies the candidates are independent, so the next kProbeCandidates step sizes of every listed instance are
// rolled out side by side into a small pool (trajectory + merit partials; the rows run in merit-only mode, nothing
// else is stored), and the instance's step is moved to the first candidate CheckArmijoCondition accepts — with the
// rejections counted as if they had been tried one by one.  The regular pass that follows evaluates that
// candidate again with all outputs and accepts it: same arithmetic, same decisions, fewer rounds.
// ---------------------------------------------------------------------------
constexpr int kProbeCandidates = 128;           // most step sizes probed per instance and round (a failing search walks
                                                // through all max_backtracking_steps of them — 100 in the examples — and a
                                                // round costs the latency of one rollout whatever it probes)
constexpr int kProbeStage = 32;                 // partials per candidate the pick kernel's loads run ahead of its additions
constexpr int kProbeEntries = 32768;            // pool size: candidates of all listed instances of one round
constexpr int kProbeRoundBudget = 8192;         // rollouts the first probing round of a tail may hold (doubling after): what
                                                // the chip integrates at once, two per wavefront at four waves per SIMD

struct ProbeEntry {
  size_t xs, us, mpart, merit, total;
  __host__ __device__ ProbeEntry(int n, int m, int N, int T) {
    auto al = [](size_t e) { return (e + 3) & ~size_t(3); };
    xs = 0;
    us = xs + al(size_t(T) * n);
    mpart = us + al(size_t(T) * m);
    merit = mpart + al(size_t(T) * N * 2);
    total = merit + 4;
  }
};

// Is instance state `s` in a line search that has already rejected a step, and is candidate j one the sequential
// loop would reach (ILQSolver::ModifyLQStrategies gives up after max_backtracking_steps rejections)?
template <typename T>
__device__ __forceinline__ bool probe_wanted(const SolveArgs<T>& sa, const SolveState<T>& s, int j) {
  return (s.stage == ST_ROLLOUT || s.stage == ST_PROBE) && !s.initial && sa.prm.linesearch && s.bt > 0 &&
         s.bt + j < sa.prm.max_backtracking_steps;
}
template <typename T>
__device__ __forceinline__ T probe_step(const SolveArgs<T>& sa, const SolveState<T>& s, int j) {
  T step = s.step;
  for (int i = 0; i < j; i++) step *= T(sa.prm.geometric_alpha_scaling);  // the products the loop forms, in order
  return step;
}

template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void probe_roll_instance(const DevProblem& p, const SolveArgs<T>& sa, int b, int slot, int j,
                                                    T* sm) {
  const InstanceBuffers<T> ib(p, sa, b);
  const SolveState<T> s = state_load<T>(ib.w, ib.L);
  if (!probe_wanted(sa, s, j)) return;
  const ProbeEntry E(p.n, p.m, p.N, p.T);
  T* const e = sa.probe_pool + (size_t(slot) * sa.probe_k + j) * E.total;
  const int snew = 1 - s.sacc;
  RolloutArgs<T> ra;
  ra.x0 = ib.XS(s.cur);
  ra.xs_ref = ib.XS(s.cur);
  ra.us_ref = ib.US(s.cur);
  ra.P = ib.PB(snew);
  ra.alpha = ib.AL(snew);
  ra.alpha_scale = probe_step(sa, s, j);
  ra.xs = e + E.xs;
  ra.us = e + E.us;
  if constexpr (NX > 0)
    rollout_instance<T, NX, NP * MU, (NX == 4 && NP == 2), (MU == 1), (NX == 3 && NP == 2 && MU == 1),
                     (NX == 4 * NP && MU == 2 && NP <= 2), dims_use_plain_rk4(NX, NP, MU)>(p, ra, sm, int(threadIdx.x),
                                                                                           nullptr, nullptr);
  else
    rollout_instance_rt<T>(p, ra, sm, int(threadIdx.x));  // the run-time-dimensioned path picks its integrator from the models
}

// Two candidates of one instance per wavefront (rollout_pair): j0 and j0 + 1 share the staged gains and references.
template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void probe_roll_pair(const DevProblem& p, const SolveArgs<T>& sa, int b, int slot, int j0, T* sm) {
  const InstanceBuffers<T> ib(p, sa, b);
  const SolveState<T> s = state_load<T>(ib.w, ib.L);
  const bool w0 = probe_wanted(sa, s, j0), w1 = j0 + 1 < sa.probe_k && probe_wanted(sa, s, j0 + 1);
  if (!w0 && !w1) return;
  const ProbeEntry E(p.n, p.m, p.N, p.T);
  const int snew = 1 - s.sacc;
  RolloutArgs<T> ra[2];
  for (int q = 0; q < 2; q++) {
    const int j = (q == 0 ? w0 : w1) ? j0 + q : (w0 ? j0 : j0 + 1);  // an idle half repeats the other candidate
    T* const e = sa.probe_pool + (size_t(slot) * sa.probe_k + j) * E.total;
    ra[q].x0 = ib.XS(s.cur);
    ra[q].xs_ref = ib.XS(s.cur);
    ra[q].us_ref = ib.US(s.cur);
    ra[q].P = ib.PB(snew);
    ra[q].alpha = ib.AL(snew);
    ra[q].alpha_scale = probe_step(sa, s, j);
    ra[q].xs = e + E.xs;
    ra[q].us = e + E.us;
  }
  rollout_pair<T, NX, NP * MU, (MU == 1)>(p, ra[0], ra[1], w0, w1, sm, int(threadIdx.x));
}

// C = 64 / N candidates of one instance per wavefront, a lane per (candidate, subsystem) (rollout_lanes): candidates
// j0 .. j0 + C - 1.  The wanted candidates are a prefix (probe_wanted is monotone in j); the other lanes repeat
// candidate j0 and store nothing.
template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void probe_roll_lanes(const DevProblem& p, const SolveArgs<T>& sa, int b, int slot, int j0, T* sm) {
  const InstanceBuffers<T> ib(p, sa, b);
  const SolveState<T> s = state_load<T>(ib.w, ib.L);
  if (j0 >= sa.probe_k || !probe_wanted(sa, s, j0)) return;
  const ProbeEntry E(p.n, p.m, p.N, p.T);
  const int snew = 1 - s.sacc;
  const int t = int(threadIdx.x);
  const int c = t / NP;
  const bool act = c < rollout_lanes_per_wave(NP) && j0 + c < sa.probe_k && probe_wanted(sa, s, j0 + c);
  const int j = act ? j0 + c : j0;
  T* const e = sa.probe_pool + (size_t(slot) * sa.probe_k + j) * E.total;
  RolloutArgs<T> ra;
  ra.x0 = ib.XS(s.cur);
  ra.xs_ref = ib.XS(s.cur);
  ra.us_ref = ib.US(s.cur);
  ra.P = ib.PB(snew);
  ra.alpha = ib.AL(snew);
  ra.alpha_scale = T(0);
  ra.xs = nullptr;
  ra.us = nullptr;
  rollout_lanes<T, NX, NP * MU, NP, (MU == 1)>(p, ra, probe_step(sa, s, j), e + E.xs, e + E.us, act, sm, t);
}

template <typename T, int NX, int NP, int MU, int PROGID = 0>
__device__ __forceinline__ void probe_rows_instance(const DevProblem& p, const short* maps, const SolveArgs<T>& sa,
                                                    int b, int slot, int j, int chunk, T* sm) {
  const InstanceBuffers<T> ib(p, sa, b);
  const SolveState<T> s = state_load<T>(ib.w, ib.L);
  if (!probe_wanted(sa, s, j)) return;
  const ProbeEntry E(p.n, p.m, p.N, p.T);
  T* const e = sa.probe_pool + (size_t(slot) * sa.probe_k + j) * E.total;
  QuadArgs<T> qa;
  qa.xs = e + E.xs;
  qa.us = e + E.us;
  qa.lambdas = p.num_constraints > 0 ? ib.w + ib.L.lambdas : nullptr;
  qa.mu = s.mu;
  qa.t_extreme = ib.t_extreme();
  qa.t_init = 0.0;
  qa.A = qa.Bm = qa.Q = qa.l = qa.R = qa.r = nullptr;
  qa.merit_part = e + E.mpart;
  qa.cost_part = nullptr;
  qa.phacc = nullptr;
  const int k0 = chunk * sa.rows_cw;
  const int nrows = p.T - k0 < sa.rows_cw ? p.T - k0 : sa.rows_cw;
  rows_chunk<T, NX, NP * MU, NP, rows_state_in_registers(NX, NP * MU), true, typename RowProgSel<PROGID>::type>(
      p, maps, qa, k0, nrows, sa.rows_cw, sm, int(threadIdx.x));
}

// The line-search bookkeeping of the candidates, in the order the loop would have met them
// (CheckArmijoCondition :350-362, the back-tracking branch of ModifyLQStrategies :333-347).
// `merits` (LDS, kProbeCandidates elements): lane j forms candidate j's merit value first — the per-row partials of its
// pool entry, summed in merit_reduce's order (one lane per candidate, not one workgroup: the sum is a serial chain either
// way, and a probing round has thousands of them; round 3 had a kernel of its own for it).  Null: read from the entries.
template <typename T>
__device__ __forceinline__ void probe_pick_instance(const DevProblem& p, const SolveArgs<T>& sa, int b, int slot,
                                                    T* merits = nullptr) {
  const InstanceBuffers<T> ib(p, sa, b);
  SolveState<T> s = state_load<T>(ib.w, ib.L);
  if (!probe_wanted(sa, s, 0)) return;
  const ilqg_solver_params& prm = sa.prm;
  const ProbeEntry E(p.n, p.m, p.N, p.T);
  const T* const e0 = sa.probe_pool + size_t(slot) * sa.probe_k * E.total;
  if (merits) {
    // Candidate j's merit value = its per-row partials summed in merit_reduce's order, one lane per candidate (a serial
    // chain either way; the candidates' entries are far apart, so every load of a lane is a cache line of its own).  What
    // it costs is the round trips, so the loads run a block of kProbeStage partials ahead of the additions.
    // (Measured and dropped in round 5: transposing the partials through LDS with contiguous half-wave reads — the
    // staging loop's own round trips made the kernel three times slower.)
    const int count = p.T * p.N * 2, skip = p.N * 2;  // the |l_i|^2 terms of k = 0 do not enter (:421)
    for (int j = threadIdx.x; j < sa.probe_k; j += blockDim.x) {
      if (s.bt + j >= prm.max_backtracking_steps) continue;
      const T* const mp = e0 + size_t(j) * E.total + E.mpart;
      T merit = T(0);
      // Elements [eb, eb + nvalid) in order with K loads in flight (indices past the end re-read the last element and
      // are not added; `head`: the odd elements below `skip` do not enter): one round trip for the head and one for
      // the tail, where a scalar loop is a round trip per element.
      auto batch = [&](auto Kc, int eb, int nvalid, bool head) {
        constexpr int K = decltype(Kc)::value;
        T v[K];
#pragma unroll
        for (int u = 0; u < K; u++) v[u] = mp[eb + u < count ? eb + u : count - 1];
#pragma unroll
        for (int u = 0; u < K; u++) {
          const bool use = u < nvalid && !(head && eb + u < skip && ((eb + u) & 1) != 0);
          merit = use ? merit + v[u] : merit;
        }
      };
      constexpr int kHead = 16;  // 2 N <= 16 elements of step 0, up to a multiple of four
      int e = ((skip + 3) & ~3) < count ? ((skip + 3) & ~3) : count;
      if (e > kHead) e = kHead;
      batch(std::integral_constant<int, kHead>{}, 0, e, true);
      for (; e < skip && e < count; e++)  // (only past kHead: never for N <= 8)
        if ((e & 1) == 0) merit += mp[e];
      // full blocks, the two register blocks taking turns (no copies between them)
      T x[kProbeStage], y[kProbeStage];
      const int nb = (count - e) / kProbeStage;
      auto load = [&](T (&dst)[kProbeStage], int at) {
#pragma unroll
        for (int u = 0; u < kProbeStage; u++) dst[u] = mp[at + u];
      };
      auto add = [&](const T (&src)[kProbeStage]) {
#pragma unroll
        for (int u = 0; u < kProbeStage; u++) merit += src[u];
      };
      if (nb > 0) load(x, e);
      for (int bi = 0; bi < nb;) {
        if (bi + 1 < nb) load(y, e + (bi + 1) * kProbeStage);
        add(x);
        if (++bi >= nb) break;
        if (bi + 1 < nb) load(x, e + (bi + 1) * kProbeStage);
        add(y);
        ++bi;
      }
      e += nb * kProbeStage;
      if (e < count) batch(std::integral_constant<int, kProbeStage>{}, e, count - e, false);
      merits[j] = T(0.5) * merit;
    }
    __syncthreads();
  }
  int tried = 0;
  bool found = false;
  T step = s.step, last_tried = s.step;
  for (int j = 0; j < sa.probe_k && s.bt + j < prm.max_backtracking_steps; j++) {
    const T merit = merits ? merits[j] : e0[size_t(j) * E.total + E.merit];
    const T scaled = T(prm.expected_decrease_fraction) * step * s.expected_decrease;
    if (s.last_merit - merit >= scaled) {
      found = true;
      break;
    }
    tried++;
    last_tried = step;
    step *= T(prm.geometric_alpha_scaling);
  }
  s.bt += tried;
  s.rejected += tried;
  if (found) {
    // The candidate's trajectory is the one the pass's rollout would produce (same function, same inputs): hand it
    // over instead of integrating it again, and enter the pass at its row stage.
    s.step = step;
    const T* const e = e0 + size_t(tried) * E.total;
    T* const xs_dst = ib.XS(1 - s.cur);
    T* const us_dst = ib.US(1 - s.cur);
    // (eight loads per lane in flight: a load-store pair per trip is a global round trip per 128 elements of a
    // trajectory, ~17 of them in a row for a lone instance)
    auto copy = [&](T* dst, const T* src, int count) {
      constexpr int U = 8;
      const int nt = blockDim.x;
      for (int i0 = threadIdx.x; i0 < count; i0 += nt * U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = src[i0 + u * nt < count ? i0 + u * nt : i0];
#pragma unroll
        for (int u = 0; u < U; u++)
          if (i0 + u * nt < count) dst[i0 + u * nt] = v[u];
      }
    };
    copy(xs_dst, e + E.xs, p.T * p.n);
    copy(us_dst, e + E.us, p.T * p.m);
    s.qmode = Q_TRIAL;
    s.stage = ST_QUAD;
  } else if (s.bt < prm.max_backtracking_steps) {
    s.step = step;  // the one after the last rejected: probed from there in the next round
    s.stage = ST_PROBE;
    if (threadIdx.x == 0 && sa.ids_next) sa.ids_next[atomicAdd(sa.unfinished + 3, 1)] = b;
  } else {          // :346-347 — out of back-tracking steps (the step stays the last one tried)
    s.step = last_tried;
    s.ok = 0;
    s.stage = ST_INNER_DONE;
    if (threadIdx.x == 0) atomicAdd(sa.unfinished + 1, 1);
  }
  state_store<T>(ib.w, ib.L, s);
}

// The state an instance enters a trial pass with: read back from its workspace, or — on the first launch of a solve
// (sa.first) — initialised here together with what the solve starts from.  Executed by every thread of the workgroup.
template <typename T>
__device__ __forceinline__ SolveState<T> trial_state_begin(const DevProblem& p, const SolveArgs<T>& sa,
                                                           const InstanceBuffers<T>& ib, int b, int n, int N, int m) {
  const WsLayout& L = ib.L;
  T* const w = ib.w;
  const int Tn = p.T, t = threadIdx.x;
  SolveState<T> s;
  if (sa.first) {
    int* const t_extreme = ib.t_extreme();
    T* const lambdas = w + L.lambdas;  // Constraint::lambdas_, zero-initialised (types.h:128)
    if (t < n) ib.xs0[t] = sa.x0[size_t(b) * n + t];  // xs[0] = x0 (src/ilq_solver.cpp:89-90)
    if (t < N) t_extreme[t] = 0;  // PlayerCost::time_of_extreme_cost_ starts at 0 (player_cost.h:70)
    for (int e = t; e < p.num_constraints * Tn; e += blockDim.x) lambdas[e] = T(0);
    if (sa.al_mode) {  // Problem's stored solution (what OverwriteSolution maintains)
      for (int e = t; e < Tn * n; e += blockDim.x) (w + L.wxs)[e] = ib.xs0[e];
      for (int e = t; e < Tn * m; e += blockDim.x) (w + L.wus)[e] = ib.us0[e];
      for (int e = t; e < Tn * m * n; e += blockDim.x) (w + L.wP)[e] = ib.P0[e];
      for (int e = t; e < Tn * m; e += blockDim.x) (w + L.wal)[e] = ib.al0[e];
    }
    s.stage = ST_ROLLOUT; s.qmode = Q_COSTS; s.initial = 1; s.cur = 0; s.sacc = 0;
    s.num_iterations = 0; s.bt = 0; s.accepted_iters = 0; s.has_converged = 0; s.ok = 1;
    s.logged = 0; s.inner_calls = 0; s.al_success = 1; s.ed_pending = 0; s.rejected = 0; s.pad2 = 0;
    // first == 2: the solver object has been called before on this workspace and its
    // last_merit_function_value_ (ilq_solver.h:189) is still what the previous call left
    const T carried = (sa.first == 2) ? state_load<T>(w, L).last_merit : dinf<T>();
    s.acc_scale = T(1); s.step = T(1); s.last_merit = carried; s.expected_decrease = dinf<T>();
    s.max_err = dinf<T>();
    s.mu = T(10);  // Constraint::mu_ = kDefaultMu (src/constraint.cpp:61) — one per instance
  } else {
    s = state_load<T>(w, L);
  }
  return s;
}

// The rollout of a pass whose instance is in ST_ROLLOUT: its inputs and outputs, and what the state becomes with it.
template <typename T>
__device__ __forceinline__ void trial_rollout_args(const SolveArgs<T>& sa, const InstanceBuffers<T>& ib, int b, int n,
                                                   SolveState<T>& s, RolloutArgs<T>& ra) {
  // initial: from the warm start (:100-104); later: trial point of the line search (:309-342)
  const int snew = 1 - s.sacc;
  ra.x0 = s.initial ? sa.x0 + size_t(b) * n : ib.XS(s.cur);
  ra.xs_ref = s.initial ? ib.XS(0) : ib.XS(s.cur);
  ra.us_ref = s.initial ? ib.US(0) : ib.US(s.cur);
  ra.P = s.initial ? ib.PB(0) : ib.PB(snew);
  ra.alpha = s.initial ? ib.AL(0) : ib.AL(snew);
  ra.alpha_scale = s.initial ? T(1) : s.step;
  ra.xs = s.initial ? ib.XS(1) : ib.XS(1 - s.cur);
  ra.us = s.initial ? ib.US(1) : ib.US(1 - s.cur);
  if (s.initial) {
    s.cur = 1;
    s.qmode = Q_COSTS;  // TotalCosts (:107) before quadraticising (:116): costs set t_extreme
  } else {
    s.qmode = sa.prm.linesearch ? Q_TRIAL : Q_LIN;
  }
}

// The rollout kernel of the split pass with TWO instances per wavefront (rollout_pair, ilqg_stages.hpp): what
// trial_part_instance<..., TRIAL_ROLL> does for one — the state in, the pass's rollout, the state out — for b0 and b1
// (b1 < 0: none) side by side.
template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void roll_pair_instances(const DevProblem& p, const SolveArgs<T>& sa, int b0, int b1, T* sm) {
  constexpr int n = NX, N = NP, m = NP * MU;
  const InstanceBuffers<T> ib0(p, sa, b0), ib1(p, sa, b1 < 0 ? b0 : b1);
  SolveState<T> s0, s1;
  RolloutArgs<T> r0, r1;
  // an instance this pass is not for (not in ST_ROLLOUT, or switched off by the caller's mask on the first launch) is
  // left alone, as ilq_roll_kernel leaves it
  auto enter = [&](const InstanceBuffers<T>& ib, int b, SolveState<T>& s, RolloutArgs<T>& r) -> bool {
    if (b < 0) return false;
    if (!sa.first) {
      if (reinterpret_cast<const SolveState<T>*>(ib.w + ib.L.state)->stage != ST_ROLLOUT) return false;
    } else if (sa.active && !sa.active[b]) {
      if (threadIdx.x == 0) reinterpret_cast<SolveState<T>*>(ib.w + ib.L.state)->stage = ST_DONE;
      return false;
    }
    s = trial_state_begin<T>(p, sa, ib, b, n, N, m);
    if (s.stage != ST_ROLLOUT) return false;
    trial_rollout_args<T>(sa, ib, b, n, s, r);
    return true;
  };
  const bool g0 = enter(ib0, b0, s0, r0), g1 = enter(ib1, b1, s1, r1);
  if (!g0 && !g1) return;
  __syncthreads();  // what the first launch wrote (xs[0]) is the rollout's to read
  rollout_pair<T, NX, NP * MU, (MU == 1)>(p, g0 ? r0 : r1, g1 ? r1 : r0, g0, g1, sm, int(threadIdx.x));
  if (g0) state_store<T>(ib0.w, ib0.L, s0);
  if (g1) state_store<T>(ib1.w, ib1.L, s1);
}

// PROGID: 0 = the row stage interprets the problem's row program; k = straight-line code for registered structure k
template <typename T, int NX, int NP, int MU, int W, int PHASE = TRIAL_FUSED, int PROGID = 0>
__device__ __forceinline__ void trial_part_instance(const DevProblem& p, const short* maps,
                                                    const SolveArgs<T>& sa, int b, T* sm) {
  static_assert(PHASE == TRIAL_FUSED || W == 1, "the split phases run one wave per instance");
  static_assert(NX > 0 || PHASE != TRIAL_FUSED, "the run-time-dimensioned path (NX = 0) runs the split passes");
  const int n = NX > 0 ? NX : p.n, N = NX > 0 ? NP : p.N, m = NX > 0 ? NP * MU : p.m;
  const int Tn = p.T;
  const ilqg_solver_params& prm = sa.prm;
  const InstanceBuffers<T> ib(p, sa, b);
  const WsLayout& L = ib.L;
  T* const w = ib.w;
  int* const t_extreme = ib.t_extreme();
  T* const lambdas = w + L.lambdas;  // Constraint::lambdas_, zero-initialised (types.h:128)
  T* const costs = sa.total_costs + size_t(b) * N;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;

  const size_t re = PHASE == TRIAL_DECIDE ? 0 : (rollout_lds_elems(n, m) + 3) & ~size_t(3);  // (trial_phase_lds_bytes)
  const size_t qe = trial_phase_quad_elems<T>(p, PHASE, sa.rows_cw, PROGID != 0 && PHASE == TRIAL_FUSED);
  constexpr int RW = W > 1 ? W - 1 : 1;     // row waves: all but the integrating wave 0
  const int rwave = W > 1 ? wave - 1 : 0;   // this wave's row scratch (-1: wave 0 of a multi-wave workgroup has none)
  T* const sm_roll = sm;
  T* const sm_quad = sm + re + size_t(rwave < 0 ? 0 : rwave) * qe;
  T* const sm_quad0 = sm + re;  // the first row wave's scratch doubles as reduction scratch between passes
  int* const flags = reinterpret_cast<int*>(sm + re + size_t(RW) * qe);  // [0] rows ready, [1] next chunk to claim
  T* const ed_slot = reinterpret_cast<T*>(flags + 2);  // deferred ExpectedDecrease, from the first row wave to everyone

  SolveState<T> s = trial_state_begin<T>(p, sa, ib, b, n, N, m);
  const int max_iters = solve_max_iters(sa);
  tl_stamp(sa.prof, b, 0, t == 0);
  const long long pr_start = clock64();
  long long qph[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // linquad phase profile of this wave (diagnostics)
  long long rph[4] = {0, 0, 0, 0};              // rollout phase profile (wave 0)

#pragma unroll 1
  while (s.stage == ST_ROLLOUT || s.stage == ST_QUAD) {
    __syncthreads();  // pass boundary: global-memory hand-off between waves
    const bool roll = PHASE != TRIAL_DECIDE && s.stage == ST_ROLLOUT;  // TRIAL_DECIDE: the pass's rollout is behind it
    RolloutArgs<T> ra;
    if (roll) trial_rollout_args<T>(sa, ib, b, n, s, ra);
    if (t == 0) {
      flags[0] = roll ? 0 : Tn;
      flags[1] = 0;
    }
    __syncthreads();
    tl_stamp(sa.prof, b, 1, t == 0);
    if (roll && wave == 0) {
#if ILQG_ROLL_PRIO
      __builtin_amdgcn_s_setprio(ILQG_ROLL_PRIO);
#endif
      if constexpr (NX > 0)
        rollout_instance<T, NX, NP * MU, (NX == 4 && NP == 2), (MU == 1), (NX == 3 && NP == 2 && MU == 1),
                         (NX == 4 * NP && MU == 2 && NP <= 2), dims_use_plain_rk4(NX, NP, MU)>(
                                         p, ra, sm_roll, lane, W > 1 ? &flags[0] : nullptr,
                                         (kProfile && sa.prof) ? rph : nullptr, kTimeline ? sa.prof : nullptr, b);
      else
        rollout_instance_rt<T>(p, ra, sm_roll, lane);
#if ILQG_ROLL_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    tl_stamp(sa.prof, b, 2, t == 0);
    if (roll && W == 1) {
      __syncthreads();
      if (t == 0) flags[0] = Tn;
      __syncthreads();
    }
    if (PHASE == TRIAL_ROLL) break;  // state (cur, qmode) goes to memory below; the row kernel takes over

    // ---- linearise / quadraticise the trajectory: every wave claims rows as they become ready ----
    const int qmode = s.qmode;
    if constexpr (PHASE == TRIAL_FUSED) {
    QuadArgs<T> qa = trial_quad_args<T>(p, ib, s, sa.compact != 0);
    qa.phacc = (kProfile && sa.prof) ? qph : nullptr;
    qa.tl = kTimeline ? sa.prof : nullptr;
    qa.tl_b = b;
    long long tq0 = (kProfile && sa.prof) ? clock64() : 0;
    // The sweep's forward pass (delta_xs, src/lq_feedback_solver.cpp:217-241) and ILQSolver::ExpectedDecrease
    // (:364-398), deferred to here: the first row wave runs them from the sweep's scratch rows while wave 0
    // integrates, before its first chunk overwrites the linearisation they read.
    static_assert(W <= 2, "the deferred forward pass reads the linearisation on the first row wave while no other row wave "
                          "may overwrite it yet: with more than one row wave set defer_forward = 0 or hold them back");
    if (s.ed_pending && rwave == 0) {
      LQArgs<T> fa{};
      fa.A = w + L.A;
      fa.scratch = w + L.lqscr;
      fa.x0 = nullptr;
      fa.dx = nullptr;
      fa.ed_out = ed_slot;
      fa.T_steps = Tn;
      if (sa.compact) {
        fa.compact = w + L.Q;
        fa.compact_tab = p.row_prog + p.rp_compact_off;
        fa.dt = p.dt;
      }
      constexpr int FWD_LDSE = 4 * 2 * ((NX * NX + LQCfg<T, NX, NP, MU>::SCR + 3) & ~3) + 2 * NX + 8;
      lq_forward_pass_body<T, NX, NP, MU, 64, FWD_LDSE>(fa, sm_quad, lane);
    }
    tl_stamp(sa.prof, b, 3, rwave == 0 && lane == 0);
    // A static row stage that keeps a chunk in registers (no LDS scratch: ilqg_rows.hpp) lets BOTH waves work on the
    // chunk that can only start when the rollout has ended: the integrating wave, idle from then on, takes one half of
    // its passes, the row wave the other — that chunk's time is what the launch ends with.
    typedef typename RowProgSel<PROGID>::type RowProg;
    constexpr bool SHARE_LAST = PROGID != 0 && W == 2 && ProgInRegisters<RowProg>::value && rows_state_in_registers(NX, NP * MU);
    if constexpr (SHARE_LAST) {
      const int cw = sa.rows_cw;
      const int nchunks = (Tn + cw - 1) / cw;
      const int rem = Tn % cw;
      constexpr unsigned first_half = static_prog_pass_split<typename RowProg::S>();
      if (wave == 0) qa.tl = nullptr;
#pragma unroll 1
      for (int c = (wave == 0 ? nchunks - 1 : 0); c < nchunks; c++) {
        const int k0 = rem == 0 ? c * cw : (c == 0 ? 0 : rem + (c - 1) * cw);  // the chunk that is not full comes first
        const int nrows = (rem != 0 && c == 0) ? rem : cw;
        if (wave != 0)
          while (progress_observe(&flags[0]) < k0 + nrows) __builtin_amdgcn_s_sleep(8);
        tl_stamp(sa.prof, b, 4 + 2 * (c < 3 ? c : 3), wave != 0 && lane == 0);
        const unsigned mask = c == nchunks - 1 ? (wave == 0 ? first_half : ~first_half) : ~0u;
        rows_chunk<T, NX, NP * MU, NP, true, false, RowProg>(p, maps, qa, k0, nrows, cw, sm_quad, lane, mask);
        tl_stamp(sa.prof, b, 5 + 2 * (c < 3 ? c : 3), wave != 0 && lane == 0);
      }
    } else
    if (rwave >= 0) {
      const int cw = sa.rows_cw;
      const int nchunks = (Tn + cw - 1) / cw;
      auto claim = [&]() {
        int c = 0;
        if (lane == 0) c = atomicAdd(&flags[1], 1);
        return __builtin_amdgcn_readfirstlane(c);
      };
      // The chunk that is not full comes FIRST: its rows are ready that much earlier in the rollout, so the row wave
      // starts sooner and the full chunks — which have to wait for the integration anyway — follow back to back.
      const int rem = Tn % cw;
#pragma unroll 1
      for (int c = claim(); c < nchunks; c = claim()) {
        const int k0 = rem == 0 ? c * cw : (c == 0 ? 0 : rem + (c - 1) * cw);
        const int nrows = (rem != 0 && c == 0) ? rem : cw;
        while (progress_observe(&flags[0]) < k0 + nrows) __builtin_amdgcn_s_sleep(8);
        if (kProfile && sa.prof) { const long long tq1 = clock64(); qph[6] += tq1 - tq0; }
        tl_stamp(sa.prof, b, 4 + 2 * (c < 3 ? c : 3), lane == 0);
        // (the register-held rows of the split kernels gain nothing for the interpreter here — measured, B = 1024: 1.45 ->
        // 1.435 M it/s; the static form reads its entries straight out of them)
        rows_chunk<T, NX, NP * MU, NP, (PROGID != 0 && rows_state_in_registers(NX, NP * MU)), false, typename RowProgSel<PROGID>::type>(
            p, maps, qa, k0, nrows, cw, sm_quad, lane);
        tl_stamp(sa.prof, b, 5 + 2 * (c < 3 ? c : 3), lane == 0);
        if (kProfile && sa.prof) { tq0 = clock64(); qph[7] += 1; }
      }
    }
    }

    if (PHASE == TRIAL_FUSED && s.ed_pending) {
      __syncthreads();
      s.expected_decrease = uniform(*ed_slot);
      s.ed_pending = 0;
      __syncthreads();
    }
    // ---- reductions and the line-search decision (wave-uniform, identical on every wave) ----
    tl_stamp(sa.prof, b, 12, t == 0);
    if (qmode == Q_COSTS) {
      costs_reduce<T>(p, w + L.cpart, costs, t_extreme, sm_quad0, int(qe));
      s.qmode = Q_INIT;
      s.stage = ST_QUAD;
    } else if (qmode == Q_INIT) {
      s.initial = 0;
      s.stage = (s.num_iterations < max_iters) ? ST_LQ : ST_INNER_DONE;
    } else {
      bool accepted = true;
      bool costs_ready = false;  // the cost totals were reduced beside the merit value and wait in LDS
      if (qmode == Q_TRIAL) {
        T merit_v = T(0);
        costs_ready = merit_costs_reduce<T>(p, w + L.mpart, w + L.cpart, t_extreme, sm_quad0, int(qe), &merit_v);
        const T merit = costs_ready ? uniform(merit_v) : uniform(merit_reduce<T>(p, w + L.mpart, sm_quad0, int(qe)));
        tl_stamp(sa.prof, b, 14, t == 0);
        const T scaled = T(prm.expected_decrease_fraction) * s.step * s.expected_decrease;
        accepted = (s.last_merit - merit >= scaled) || sa.forced_steps != nullptr;  // CheckArmijoCondition :350-362
        if (accepted) {
          const T diff = s.last_merit - merit;
          s.has_converged =
              (merit <= s.last_merit) && ((diff < T(0) ? -diff : diff) < T(prm.convergence_tolerance));
          s.last_merit = merit;
        }
      } else {
        __syncthreads();
      }
      if (accepted) {
        s.cur = 1 - s.cur;
        s.sacc = 1 - s.sacc;
        s.acc_scale = s.step;
        s.accepted_iters++;
        if (costs_ready)  // TotalCosts of the accepted iterate (:158)
          costs_commit<T>(p, sm_quad0, costs, t_extreme);
        else
          costs_reduce<T>(p, w + L.cpart, costs, t_extreme, sm_quad0, int(qe));
        s.stage = (s.num_iterations < max_iters && (sa.fixed_iters > 0 || !s.has_converged)) ? ST_LQ : ST_INNER_DONE;
      } else {
        s.bt++;
        s.rejected++;
        if (s.bt >= prm.max_backtracking_steps) {  // :346-347, :146-155 — keep the last accepted iterate
          s.ok = 0;
          s.stage = ST_INNER_DONE;
        } else {
          s.step *= T(prm.geometric_alpha_scaling);
          s.stage = ST_ROLLOUT;
        }
      }
    }
    tl_stamp(sa.prof, b, 15, t == 0);
    if (PHASE == TRIAL_DECIDE) break;  // another pass, if the instance needs one, is the host's to launch
    // hand-off (the host gave a list to fill): a rejected step leaves the fused kernel too — the rest of this line
    // search goes through the split passes, where the next step sizes are probed side by side
    if (PHASE == TRIAL_FUSED && sa.ids_next != nullptr && s.stage == ST_ROLLOUT && s.bt > 0) break;
  }
  state_store<T>(w, L, s);
  tl_stamp(sa.prof, b, 13, t == 0);
  if (PHASE == TRIAL_ROLL) return;
  // wants a sweep / wants the exit path / (split passes only) wants another pass
  if (t == 0) {
    const int slot = s.stage == ST_LQ ? 0 : (s.stage == ST_ROLLOUT || s.stage == ST_QUAD) ? 3 : 1;
    const int at = atomicAdd(sa.unfinished + slot, 1);
    if (slot == 3 && sa.ids_next) sa.ids_next[at] = b;
  }
  if (kProfile && t == 0 && sa.prof) sa.prof[size_t(b) * 96 + 1] += clock64() - pr_start;
  if (kProfile && t == 0 && sa.prof) {
#pragma unroll
    for (int i = 0; i < 4; i++) sa.prof[size_t(b) * 96 + 88 + i] += rph[i];
  }
  if (kProfile && lane == 0 && sa.prof && wave < 2) {
#pragma unroll
    for (int i = 0; i < 8; i++) sa.prof[size_t(b) * 96 + 64 + 8 * wave + i] += qph[i];
  }
}

// ---------------------------------------------------------------------------
// Iterate log (ilqg_iterate_log, include/ilqg.h): what SolverLog::AddSolverIterate deep-copies in the reference
// (src/ilq_solver.cpp:111,164).  An instance shows an accepted iterate at a kernel boundary exactly when its stage is
// LQ (the loop goes on) or INNER_DONE (it ends); iterate q of an inner solve goes to slot `logged so far` + q, so a
// second visit of the same state (an instance that waits several rounds for the batch's sweep, or one whose line
// search failed and still holds its last accepted iterate) rewrites the slot with the same values.
// Executed by every thread of a workgroup, one workgroup per instance, launched in front of the exit / sweep kernels.
// ---------------------------------------------------------------------------
template <typename T>
struct IterLog {
  T *xs, *us, *costs, *P, *alpha;
  int* count;
  int capacity;
};

template <typename T>
__device__ __forceinline__ void log_part_instance(const DevProblem& p, const SolveArgs<T>& sa, const IterLog<T>& lg, int b) {
  const InstanceBuffers<T> ib(p, sa, b);
  const SolveState<T> s = state_load<T>(ib.w, ib.L);
  if (s.stage != ST_LQ && s.stage != ST_INNER_DONE) return;
  const int slot = (sa.al_mode ? s.logged : 0) + s.accepted_iters;
  const int t = threadIdx.x, nt = blockDim.x;
  if (t == 0 && lg.count[b] < slot + 1) lg.count[b] = slot + 1;
  if (slot >= lg.capacity) return;
  const size_t Tn = p.T, n = p.n, m = p.m, N = p.N;
  const size_t at = size_t(b) * lg.capacity + slot;
  const T* xs = ib.XS(s.cur);
  const T* us = ib.US(s.cur);
  for (size_t e = t; e < Tn * n; e += nt) lg.xs[at * Tn * n + e] = xs[e];
  for (size_t e = t; e < Tn * m; e += nt) lg.us[at * Tn * m + e] = us[e];
  if (size_t(t) < N) lg.costs[at * N + t] = sa.total_costs[size_t(b) * N + t];
  if (lg.P) {
    const T* P = ib.PB(s.sacc);
    for (size_t e = t; e < Tn * m * n; e += nt) lg.P[at * Tn * m * n + e] = P[e];
  }
  if (lg.alpha) {  // the strategies the reference logs carry the accepted step (ScaleAlphas is destructive, :66-72)
    const T* al = ib.AL(s.sacc);
    for (size_t e = t; e < Tn * m; e += nt) lg.alpha[at * Tn * m + e] = al[e] * s.acc_scale;
  }
}

// The anytime exit of ILQSolver::Solve (src/ilq_solver.cpp:123-124): once the host's clock says the next iteration no
// longer fits max_runtime, the instances that would start one (stage LQ) take the loop's normal exit instead — success
// stays true, has_converged is whatever the last accepted step left.
template <typename T>
__device__ __forceinline__ void deadline_part_instance(const DevProblem& p, const SolveArgs<T>& sa, int b) {
  const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
  SolveState<T>* const st = reinterpret_cast<SolveState<T>*>(sa.ws + size_t(b) * sa.ws_stride + L.state);
  if (threadIdx.x == 0 && st->stage == ST_LQ) {
    st->stage = ST_INNER_DONE;
    atomicAdd(sa.unfinished + 1, 1);
  }
}

// ---------------------------------------------------------------------------
// LQ part: the Riccati sweep of one instance whose stage is LQ.  Kept free of everything else (the
// exit path lives in the trial kernel) so that the sweep's registers are all it has to hold.
// PW: the workgroup has one wave per player and runs the player-parallel MFMA feedback sweep;
// likewise the open-loop sweep (ilqg_lq_openloop.hpp); otherwise LQCfg::NT threads run the VALU feedback sweep.
// ---------------------------------------------------------------------------
// KIND: which sweep this kernel instantiation carries (one each, so that the register allocation of one does not
// pay for the others): LQ_VALU_FEEDBACK, LQ_PLAYER_WAVES (PW above) or LQ_OPEN_LOOP.
enum { LQ_VALU_FEEDBACK = 0, LQ_PLAYER_WAVES = 1, LQ_OPEN_LOOP = 2, LQ_PLAYER_WAVES_PACKED = 3, LQ_OPEN_LOOP_COMPACT = 4,
       LQ_SINGLE_WAVE = 5 };  // _SINGLE_WAVE: the one-tile feedback sweep with one wave per instance (ilqg_lq_feedback1w.hpp)
// _PACKED: ilqg_api.hip; _COMPACT: the open-loop sweep reading compact rows (its own instantiation: register budget)
template <typename T, int NX, int NP, int MU, int KIND>
__device__ __forceinline__ void lq_part_instance(const DevProblem& p, const SolveArgs<T>& sa, int b, T* sm) {
  const int Tn = p.T;
  const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
  T* const w = sa.ws + size_t(b) * sa.ws_stride;
  SolveState<T>* const st = reinterpret_cast<SolveState<T>*>(w + L.state);
  const int sacc = __builtin_amdgcn_readfirstlane(st->sacc);
  const long long pr_start = clock64();
  // LQ game at the current operating point (src/ilq_solver.cpp:136-143) + ExpectedDecrease (:303)
  LQArgs<T> la;
  la.A = w + L.A;
  la.Bm = w + L.B;
  la.Q = w + L.Q;
  la.l = w + L.l;
  la.R = w + L.R;
  la.r = w + L.r;
  la.x0 = nullptr;
  la.P = sacc ? sa.P + size_t(b) * Tn * p.m * p.n : w + L.P1;  // strategy buffer 1 - sacc
  la.alpha = sacc ? sa.alpha + size_t(b) * Tn * p.m : w + L.al1;
  constexpr bool kOL = KIND == LQ_OPEN_LOOP || KIND == LQ_OPEN_LOOP_COMPACT;
  const bool defer = sa.defer_forward && !kOL;
  la.dx = defer ? nullptr : w + L.dx;
  la.defer_forward = defer ? 1 : 0;
  la.scratch = w + L.lqscr;
  // where the sweep leaves the expected decrease: an LDS slot that is free once it ends (feedback sweeps), or
  // one past the open-loop sweep's own working set (the launch reserves it)
  constexpr int ed_slot = kOL ? OLCfg<T, NX, NP, MU>::LDS_ELEMS
                          : KIND == LQ_SINGLE_WAVE ? W1Cfg<T, NX, NP, MU>::ELEMS
                          : (KIND == LQ_PLAYER_WAVES && !LQCfg<T, NX, NP, MU>::MFMA_ONE_TILE) ? FB2Cfg<T, NX, NP, MU>::LDS_ELEMS
                                                                                              : LQCfg<T, NX, NP, MU>::oX;
  la.ed_out = defer ? nullptr : sm + ed_slot;
  la.T_steps = Tn;
  la.adaptive = 1;
  la.symmetric = 1;  // linquad_compute writes H(x,y) and H(y,x) from the same value
  if constexpr (kOL) {
    // block-diagonal A (one block per subsystem): every player brings a state block of its own
    bool blocks = true;
#pragma unroll
    for (int i = 0; i < NP; i++) blocks = blocks && p.xoff[i + 1] > p.xoff[i];
    if (blocks) {
      la.nsub = NP;
#pragma unroll
      for (int i = 0; i <= NP; i++) la.xoff[i] = p.xoff[i];
    }
  }
  la.prio_div = KIND == LQ_PLAYER_WAVES ? sa.prio_div : 0;
  if (sa.compact && (KIND == LQ_PLAYER_WAVES || KIND == LQ_OPEN_LOOP_COMPACT || KIND == LQ_SINGLE_WAVE)) {
    la.compact = w + L.Q;
    la.compact_tab = p.row_prog + p.rp_compact_off;
    la.dt = p.dt;
  }
  la.ph = (kProfile && sa.prof) ? sa.prof + size_t(b) * 96 + 8 : nullptr;
  la.tl = kTimeline ? sa.prof : nullptr;
  la.tl_b = b;
  tl_stamp(sa.prof, b, 16, threadIdx.x == 0);
  if constexpr (KIND == LQ_PLAYER_WAVES) {
    if constexpr (LQCfg<T, NX, NP, MU>::MFMA_ONE_TILE) {
      if (la.compact != nullptr && defer)
        lq_feedback_instance_mfma_pw<T, NX, NP, MU, true>(la, p.pairs, sm);
      else
        lq_feedback_instance_mfma<T, NX, NP, MU>(la, p.pairs, sm);
    } else
    lq_feedback_instance_mfma<T, NX, NP, MU>(la, p.pairs, sm);
  } else if constexpr (KIND == LQ_SINGLE_WAVE) {
    // (the launcher only picks this kind with compact rows; the expected decrease comes out of the sweep itself — the
    // adjoint recursion of ilqg_lq_feedback1w.hpp — unless the forward pass is deferred to the trial kernel)
    if constexpr (W1Cfg<T, NX, NP, MU>::SUPPORTED) {
      la.dx = nullptr;
      lq_feedback_instance_mfma_1w<T, NX, NP, MU>(la, p.pairs, sm);
    }
  } else if constexpr (KIND == LQ_OPEN_LOOP) {
    lq_openloop_instance<T, NX, NP, MU>(la, p.pairs, sm);  // SolverParams::open_loop (ilq_solver.h:76-81)
  } else if constexpr (KIND == LQ_OPEN_LOOP_COMPACT) {
    lq_openloop_instance<T, NX, NP, MU, true>(la, p.pairs, sm);
  } else {
    lq_feedback_instance<T, NX, NP, MU>(la, p.pairs, sm);
  }
  __syncthreads();
  tl_stamp(sa.prof, b, 19, threadIdx.x == 0);
  if (threadIdx.x == 0) {
    if (defer)
      st->ed_pending = 1;
    else
      st->expected_decrease = sm[ed_slot];
    st->num_iterations += 1;
    st->step = sa.forced_steps ? sa.forced_steps[size_t(b) * sa.fixed_iters + (st->num_iterations - 1)]
                               : T(sa.prm.initial_alpha_scaling);
    st->bt = 0;
    st->stage = ST_ROLLOUT;
    if (kProfile && sa.prof) sa.prof[size_t(b) * 96 + 2] += clock64() - pr_start;
  }
}

// The LQ part on the run-time-dimensioned sweeps (ilqg_lq_generic.hpp): dense rows in, strategies and the expected
// decrease out, no deferred forward pass.  One workgroup per instance whose stage is LQ.
template <typename T>
__device__ __forceinline__ void lq_part_generic(const DevProblem& p, const SolveArgs<T>& sa, int b, T* sm) {
  const int Tn = p.T;
  const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
  T* const w = sa.ws + size_t(b) * sa.ws_stride;
  SolveState<T>* const st = reinterpret_cast<SolveState<T>*>(w + L.state);
  const int sacc = __builtin_amdgcn_readfirstlane(st->sacc);
  GenDims gd;
  gd.n = p.n; gd.N = p.N; gd.m = p.m; gd.T = p.T;
  for (int i = 0; i < kMaxPlayers; i++) {
    gd.udim[i] = p.udim[i];
    gd.uoff[i] = p.uoff[i];
  }
  gd.uoff[kMaxPlayers] = p.uoff[kMaxPlayers];
  GenLQArgs<T> la;
  la.A = w + L.A; la.Bm = w + L.B; la.Q = w + L.Q; la.l = w + L.l; la.R = w + L.R; la.r = w + L.r;
  la.x0 = nullptr;
  la.P = sacc ? sa.P + size_t(b) * Tn * p.m * p.n : w + L.P1;  // strategy buffer 1 - sacc
  la.alpha = sacc ? sa.alpha + size_t(b) * Tn * p.m : w + L.al1;
  la.dx = w + L.dx;
  la.costates = nullptr;
  la.scratch = w + L.lqscr;
  // the expected decrease: one LDS slot past the sweep's own working set (the launch reserves it)
  la.ed_out = sm + (sa.prm.open_loop ? gen_openloop_lds_elems(p.n, p.N, p.m) : gen_feedback_lds_elems(p.n, p.N, p.m));
  la.adaptive = 1;
  const ParDevice par;
  if (sa.prm.open_loop)
    lq_openloop_generic<T>(gd, la, p.pairs, sm, par);
  else
    lq_feedback_generic<T>(gd, la, p.pairs, sm, par);
  __syncthreads();
  if (threadIdx.x == 0) {
    st->expected_decrease = *la.ed_out;
    st->num_iterations += 1;
    st->step = sa.forced_steps ? sa.forced_steps[size_t(b) * sa.fixed_iters + (st->num_iterations - 1)]
                               : T(sa.prm.initial_alpha_scaling);
    st->bt = 0;
    st->stage = ST_ROLLOUT;
  }
}

}  // namespace ilqg
