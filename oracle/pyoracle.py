"""ctypes access to oracle/liboracle.so — the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py, never by the product path (ilqgames_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from ilqgames_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle_capi.cpp", "ilqg_oracle.hpp")] + [
        os.path.join(_HERE, "..", "include", "ilqg.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-B", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_problem_create.restype = C.c_void_p
        _LIB.oracle_player_value.restype = C.c_double
        _LIB.oracle_min_eigenvalue.restype = C.c_double
    return _LIB


def _np(dtype):
    return np.float32 if dtype == abi.F32 else np.float64


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def lq_solve(dims, A, Bm, Q, l, R, r, pairs, x0=None, open_loop=False, want_dx=True, want_costates=False,
             threads=1):
    """LQFeedbackSolver::Solve / LQOpenLoopSolver::Solve on host arrays (layouts of include/ilqg.h)."""
    dt = _np(dims.dtype)
    B, T, n, N = dims.batch, dims.T, dims.n, dims.num_players
    m = sum(dims.udim[i] for i in range(N))
    arrs = [np.ascontiguousarray(a, dtype=dt) for a in (A, Bm, Q, l, R, r)]
    P = np.zeros((B, T, m * n), dt)
    alpha = np.zeros((B, T, m), dt)
    dx = np.zeros((B, T, n), dt) if want_dx else None
    co = np.zeros((B, T, N, n), dt) if want_costates else None
    x0a = None if x0 is None else np.ascontiguousarray(x0, dtype=dt)
    fn = lib().oracle_lq_openloop if open_loop else lib().oracle_lq_feedback
    rc = fn(C.byref(dims), *[_p(a) for a in arrs], abi.make_pairs(pairs), len(pairs), _p(x0a), _p(P), _p(alpha),
            _p(dx), _p(co), int(threads))
    if rc != 0:
        raise ValueError("oracle LQ solve: status %d" % rc)
    return P, alpha, dx, co


class OracleProblem:
    def __init__(self, spec):
        self.spec = spec
        self.desc, self._keep = spec.build(abi.F64)
        self.h = C.c_void_p(lib().oracle_problem_create(C.byref(self.desc)))
        self.n, self.m, self.N, self.T = spec.n, spec.m, len(spec.subsystems), spec.T
        self.pairs = spec.pairs()
        self.Rsz = sum(spec.udims[j] ** 2 for _, j in self.pairs)
        self.rsz = sum(spec.udims[j] for _, j in self.pairs)
        self.nc = spec.num_constraints

    def __del__(self):
        try:
            lib().oracle_problem_destroy(self.h)
        except Exception:
            pass

    def rollout(self, dtype, x0, xs_ref, us_ref, P, alpha, alpha_scale=None):
        dt = _np(dtype)
        B = x0.shape[0]
        xs = np.zeros((B, self.T, self.n), dt)
        us = np.zeros((B, self.T, self.m), dt)
        a = [np.ascontiguousarray(v, dtype=dt) for v in (x0, xs_ref, us_ref, P, alpha)]
        sc = None if alpha_scale is None else np.ascontiguousarray(alpha_scale, dtype=dt)
        lib().oracle_rollout(self.h, dtype, B, *[_p(v) for v in a], _p(sc), _p(xs), _p(us))
        return xs, us

    def linearize(self, dtype, xs, us):
        dt = _np(dtype)
        B = xs.shape[0]
        A = np.zeros((B, self.T, self.n * self.n), dt)
        Bm = np.zeros((B, self.T, self.n * self.m), dt)
        xs, us = np.ascontiguousarray(xs, dtype=dt), np.ascontiguousarray(us, dtype=dt)
        lib().oracle_linearize(self.h, dtype, B, _p(xs), _p(us), _p(A), _p(Bm))
        return A, Bm

    def quadraticize(self, dtype, xs, us, lambdas=None, mu=None, t_extreme=None):
        dt = _np(dtype)
        B = xs.shape[0]
        Q = np.zeros((B, self.T, self.N, self.n * self.n), dt)
        l = np.zeros((B, self.T, self.N, self.n), dt)
        R = np.zeros((B, self.T, self.Rsz), dt)
        r = np.zeros((B, self.T, self.rsz), dt)
        xs, us = np.ascontiguousarray(xs, dtype=dt), np.ascontiguousarray(us, dtype=dt)
        lam = None if lambdas is None else np.ascontiguousarray(lambdas, dtype=dt)
        mua = None if mu is None else np.ascontiguousarray(mu, dtype=dt)
        te = None if t_extreme is None else np.ascontiguousarray(t_extreme, dtype=np.int32)
        lib().oracle_quadraticize(self.h, dtype, B, _p(xs), _p(us), _p(lam), _p(mua), _p(te), _p(Q), _p(l), _p(R),
                                  _p(r))
        return Q, l, R, r

    def total_costs(self, dtype, xs, us, t_extreme=None):
        dt = _np(dtype)
        B = xs.shape[0]
        costs = np.zeros((B, self.N), dt)
        te = np.zeros((B, self.N), np.int32) if t_extreme is None else np.ascontiguousarray(t_extreme, np.int32).copy()
        xs, us = np.ascontiguousarray(xs, dtype=dt), np.ascontiguousarray(us, dtype=dt)
        lib().oracle_total_costs(self.h, dtype, B, _p(xs), _p(us), _p(costs), _p(te))
        return costs, te

    def solve(self, dtype, x0, xs=None, us=None, P=None, alpha=None, fixed_iters=0, merit_log_len=0, threads=1,
              augmented_lagrangian=False, forced_steps=None):
        """ILQSolver::Solve (or AugmentedLagrangianSolver::Solve) per instance.
        forced_steps [B][fixed_iters]: every iteration takes the given step size instead of running the line search.
        Returns dict with final op/strategies/costs/iters/status."""
        if augmented_lagrangian:
            fixed_iters = -1
        dt = _np(dtype)
        B = x0.shape[0]
        x0 = np.ascontiguousarray(x0, dtype=dt)
        xs = np.zeros((B, self.T, self.n), dt) if xs is None else np.ascontiguousarray(xs, dtype=dt).copy()
        us = np.zeros((B, self.T, self.m), dt) if us is None else np.ascontiguousarray(us, dtype=dt).copy()
        P = np.zeros((B, self.T, self.m * self.n), dt) if P is None else np.ascontiguousarray(P, dtype=dt).copy()
        alpha = np.zeros((B, self.T, self.m), dt) if alpha is None else np.ascontiguousarray(alpha, dtype=dt).copy()
        costs = np.zeros((B, self.N), dt)
        iters = np.zeros(B, np.int32)
        status = np.zeros(B, np.int32)
        conv = np.zeros(B, np.int32)
        rawP = np.zeros_like(P)
        rawA = np.zeros_like(alpha)
        ml = np.zeros((B, merit_log_len, 4), dt) if merit_log_len else None
        if forced_steps is not None:
            fs = np.ascontiguousarray(forced_steps, dtype=dt)
            assert fs.shape == (B, fixed_iters)
            lib().oracle_ilq_solve_forced(self.h, dtype, B, _p(x0), _p(xs), _p(us), _p(P), _p(alpha), _p(costs),
                                          _p(iters), _p(status), _p(conv), int(fixed_iters), _p(fs), _p(rawP),
                                          _p(rawA), _p(ml), int(merit_log_len))
        else:
            lib().oracle_ilq_solve(self.h, dtype, B, _p(x0), _p(xs), _p(us), _p(P), _p(alpha), _p(costs), _p(iters),
                                   _p(status), _p(conv), int(fixed_iters), _p(rawP), _p(rawA), _p(ml),
                                   int(merit_log_len), int(threads))
        return dict(xs=xs, us=us, P=P, alpha=alpha, costs=costs, iters=iters, status=status, converged=conv,
                    rawP=rawP, rawAlpha=rawA, log=ml)

    def receding_horizon_shift(self, dtype, x0, t0, planner_runtime, plan_t0, xs, us, P, alpha):
        """Problem::SetUpNextRecedingHorizon per instance.  Returns dict(xs, us, P, alpha, x0_next, first_step,
        new_plan_t0); inputs are not modified."""
        dt = _np(dtype)
        B = x0.shape[0]
        a = [np.ascontiguousarray(v, dtype=dt).copy() for v in (xs, us, P, alpha)]
        x0 = np.ascontiguousarray(x0, dtype=dt)
        x0n = np.zeros((B, self.n), dt)
        first = np.zeros(B, np.int32)
        npt = C.c_double(0.0)
        lib().oracle_receding_horizon_shift(self.h, dtype, B, _p(x0), C.c_double(t0), C.c_double(planner_runtime),
                                            C.c_double(plan_t0), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(x0n),
                                            _p(first), C.byref(npt))
        return dict(xs=a[0], us=a[1], P=a[2], alpha=a[3], x0_next=x0n, first_step=first, new_plan_t0=npt.value)

    # ---- receding-horizon harness (plans stored [B][cap][...] with per-instance length / start time) ----
    def new_plan(self, dtype, B, cap=None):
        """Empty SolutionSplicer state: dict(xs, us, P, alpha, len, t0)."""
        dt = _np(dtype)
        cap = self.T + 5 if cap is None else cap
        return dict(xs=np.zeros((B, cap, self.n), dt), us=np.zeros((B, cap, self.m), dt),
                    P=np.zeros((B, cap, self.m * self.n), dt), alpha=np.zeros((B, cap, self.m), dt),
                    len=np.zeros(B, np.int32), t0=np.zeros(B, np.float64))

    def plan_integrate(self, dtype, plan, t_from, t_to, must_contain, x, active):
        """MultiPlayerIntegrableSystem::Integrate(t0, t, x0, op, strategies); x and active are updated in place."""
        cap = plan["xs"].shape[1]
        lib().oracle_plan_integrate(self.h, dtype, x.shape[0], cap, _p(plan["xs"]), _p(plan["us"]), _p(plan["P"]),
                                    _p(plan["alpha"]), _p(plan["len"]), _p(plan["t0"]), C.c_double(t_from),
                                    C.c_double(t_to), C.c_double(must_contain), _p(x), _p(active))

    def receding_horizon_sync(self, dtype, plan, x, t, planner_runtime, active):
        """OverwriteSolution(plan) + SetUpNextRecedingHorizon per instance.  Returns the next solve's
        dict(xs, us, P, alpha, x0, t0, first_step); active is updated in place."""
        dt = _np(dtype)
        B = x.shape[0]
        cap = plan["xs"].shape[1]
        o = dict(xs=np.zeros((B, self.T, self.n), dt), us=np.zeros((B, self.T, self.m), dt),
                 P=np.zeros((B, self.T, self.m * self.n), dt), alpha=np.zeros((B, self.T, self.m), dt),
                 x0=np.zeros((B, self.n), dt), t0=np.zeros(B, np.float64), first_step=np.zeros(B, np.int32))
        lib().oracle_receding_horizon_sync(self.h, dtype, B, cap, _p(plan["xs"]), _p(plan["us"]), _p(plan["P"]),
                                           _p(plan["alpha"]), _p(plan["len"]), _p(plan["t0"]), _p(x), C.c_double(t),
                                           C.c_double(planner_runtime), _p(o["xs"]), _p(o["us"]), _p(o["P"]),
                                           _p(o["alpha"]), _p(o["x0"]), _p(o["t0"]), _p(o["first_step"]), _p(active))
        return o

    def solution_splice(self, dtype, plan, sol, solve_t0, converged=None, active=None):
        """SolutionSplicer construction (plan len 0) / Splice; plan is updated in place."""
        cap = plan["xs"].shape[1]
        B = plan["xs"].shape[0]
        t0 = np.ascontiguousarray(solve_t0, np.float64)
        lib().oracle_solution_splice(self.h, dtype, B, cap, _p(plan["xs"]), _p(plan["us"]), _p(plan["P"]),
                                     _p(plan["alpha"]), _p(plan["len"]), _p(plan["t0"]), _p(sol["xs"]), _p(sol["us"]),
                                     _p(sol["P"]), _p(sol["alpha"]), _p(t0), _p(converged), _p(active))

    def solve_resume(self, dtype, x0, xs, us, P, alpha, last_merit, augmented_lagrangian=False, threads=1):
        """Solve() of a solver object that has been called before: last_merit [B] (updated in place) carries
        ILQSolver::last_merit_function_value_."""
        dt = _np(dtype)
        B = x0.shape[0]
        x0 = np.ascontiguousarray(x0, dtype=dt)
        a = [np.ascontiguousarray(v, dtype=dt).copy() for v in (xs, us, P, alpha)]
        costs = np.zeros((B, self.N), dt)
        iters, status, conv = (np.zeros(B, np.int32) for _ in range(3))
        lib().oracle_solve_resume(self.h, dtype, B, _p(x0), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(costs),
                                  _p(iters), _p(status), _p(conv), int(augmented_lagrangian), _p(last_merit),
                                  int(threads))
        return dict(xs=a[0], us=a[1], P=a[2], alpha=a[3], costs=costs, iters=iters, status=status, converged=conv)

    def receding_horizon_simulate(self, dtype, x_init, final_time, planner_runtime, extra_time=0.25, solve_time=0.25,
                                  augmented_lagrangian=False, max_records=64, threads=1):
        """RecedingHorizonSimulator with a fixed simulated solve time.  Returns dict of per-record arrays
        [B][max_records][...], num_records [B], the final spliced plan and the final true state."""
        dt = _np(dtype)
        B = x_init.shape[0]
        R, T, n, m = max_records, self.T, self.n, self.m
        x_init = np.ascontiguousarray(x_init, dtype=dt)
        o = dict(num_records=np.zeros(B, np.int32), t_call=np.zeros((B, R)), x_measured=np.zeros((B, R, n), dt),
                 x0=np.zeros((B, R, n), dt), plan_t0=np.zeros((B, R)), first_step=np.zeros((B, R), np.int32),
                 xs=np.zeros((B, R, T, n), dt), us=np.zeros((B, R, T, m), dt), P=np.zeros((B, R, T, m * n), dt),
                 alpha=np.zeros((B, R, T, m), dt), iters=np.zeros((B, R), np.int32), ok=np.zeros((B, R), np.int32),
                 converged=np.zeros((B, R), np.int32), max_backtracks=np.zeros((B, R), np.int32),
                 plan=self.new_plan(dtype, B), x=np.zeros((B, n), dt))
        pl = o["plan"]
        lib().oracle_receding_horizon_simulate(
            self.h, dtype, B, _p(x_init), C.c_double(final_time), C.c_double(planner_runtime), C.c_double(extra_time),
            C.c_double(solve_time), int(augmented_lagrangian), R, _p(o["num_records"]), _p(o["t_call"]),
            _p(o["x_measured"]), _p(o["x0"]), _p(o["plan_t0"]), _p(o["first_step"]), _p(o["xs"]), _p(o["us"]),
            _p(o["P"]), _p(o["alpha"]), _p(o["iters"]), _p(o["ok"]), _p(o["converged"]), _p(o["max_backtracks"]),
            pl["xs"].shape[1],
            _p(pl["xs"]), _p(pl["us"]), _p(pl["P"]), _p(pl["alpha"]), _p(pl["len"]), _p(pl["t0"]), _p(o["x"]),
            int(threads))
        return o

    def strategy_costs(self, dtype, x0, xs, us, P, alpha, open_loop=False, euler=True):
        """ComputeStrategyCosts (src/compute_strategy_costs.cpp:61-106) per instance -> [B][N]."""
        dt = _np(dtype)
        a = [np.ascontiguousarray(v, dtype=dt) for v in (x0, xs, us, P, alpha)]
        costs = np.zeros((a[0].shape[0], self.N), dt)
        lib().oracle_nash(self.h, dtype, a[0].shape[0], _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]),
                          C.c_double(0.0), int(open_loop), int(euler), _p(costs), None, None, 1)
        return costs

    def check_local_nash(self, dtype, x0, xs, us, P, alpha, max_perturbation, open_loop=False, threads=8):
        """NumericalCheckLocalNashEquilibrium per instance -> (is_nash [B] int32, margin [B])."""
        dt = _np(dtype)
        a = [np.ascontiguousarray(v, dtype=dt) for v in (x0, xs, us, P, alpha)]
        B = a[0].shape[0]
        ok = np.zeros(B, np.int32)
        margin = np.zeros(B, dt)
        lib().oracle_nash(self.h, dtype, B, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]),
                          C.c_double(max_perturbation), int(open_loop), 1, None, _p(ok), _p(margin), int(threads))
        return ok, margin


    def check_sufficient_nash(self, dtype, xs, us):
        """CheckSufficientLocalNashEquilibrium per instance -> (ok [B] int32, smallest eigenvalue met [B])."""
        dt = _np(dtype)
        xs, us = np.ascontiguousarray(xs, dtype=dt), np.ascontiguousarray(us, dtype=dt)
        B = xs.shape[0]
        ok = np.zeros(B, np.int32)
        worst = np.zeros(B, np.float64)
        lib().oracle_sufficient_nash(self.h, dtype, B, _p(xs), _p(us), _p(ok), _p(worst))
        return ok, worst


    def dynamics(self, dtype, x, u, euler=False):
        x = np.ascontiguousarray(x, np.float64)
        u = np.ascontiguousarray(u, np.float64)
        xdot = np.zeros(self.n)
        xn = np.zeros(self.n)
        lib().oracle_dynamics(self.h, dtype, _p(x), _p(u), _p(xdot), _p(xn), int(euler))
        return xdot, xn

    def player_value(self, player, x, u, include_constraints=False, lam=0.0, mu=10.0, step=-1):
        x = np.ascontiguousarray(x, np.float64)
        u = np.ascontiguousarray(u, np.float64)
        return lib().oracle_player_value(self.h, int(player), _p(x), _p(u), int(include_constraints),
                                         C.c_double(lam), C.c_double(mu), int(step))


def polyline_closest_point(pts, q, dtype=abi.F32):
    pts = np.ascontiguousarray(np.asarray(pts, np.float32).reshape(-1))
    out = np.zeros(9)
    lib().oracle_polyline_closest_point(dtype, pts.ctypes.data_as(C.POINTER(C.c_float)), len(pts) // 2,
                                        C.c_double(q[0]), C.c_double(q[1]), _p(out))
    return dict(point=out[0:2], ssd=out[2], is_vertex=bool(out[3]), is_endpoint=bool(out[4]), segment=out[5:9])


def segment_closest_point(p1, p2, q):
    a = np.ascontiguousarray(np.array([p1[0], p1[1], p2[0], p2[1]], np.float32))
    out = np.zeros(5)
    lib().oracle_segment_closest_point(a.ctypes.data_as(C.POINTER(C.c_float)), C.c_double(q[0]), C.c_double(q[1]),
                                       _p(out))
    return dict(point=out[0:2], ssd=out[2], is_endpoint=bool(out[3]), side=bool(out[4]))


def min_eigenvalue(a):
    a = np.asfortranarray(np.asarray(a, np.float64))
    return lib().oracle_min_eigenvalue(a.shape[0], a.ctypes.data_as(C.c_void_p))
