"""ctypes loader for libilqg_hip.so — the product path.

There is NO CPU fallback: if the HIP library is missing or no gfx950 device is visible
every entry point raises.  torch is used only to own HBM buffers and streams.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# ILQG_HIP_LIB: an alternative build of the same library (profile / experiment builds of scripts/devbuild.py)
LIB_PATH = os.environ.get("ILQG_HIP_LIB") or os.path.join(_HERE, "libilqg_hip.so")
_LIB = None

EXPORTS = [
    "ilqg_lq_feedback_batch", "ilqg_lq_openloop_batch", "ilqg_default_solver_params", "ilqg_problem_create",
    "ilqg_problem_destroy", "ilqg_workspace_bytes", "ilqg_rollout_batch", "ilqg_linearize_batch",
    "ilqg_quadraticize_batch", "ilqg_problem_pairs", "ilqg_total_costs_batch", "ilqg_ilq_solve_batch",
    "ilqg_last_error", "ilqg_abi_version", "ilqg_device_info", "ilqg_selftest_mfma", "ilqg_al_solve_batch", "ilqg_plan_integrate_batch", "ilqg_receding_horizon_sync_batch",
    "ilqg_solution_splice_batch", "ilqg_solve_again_batch", "ilqg_strategy_costs_batch",
    "ilqg_check_local_nash_batch", "ilqg_check_sufficient_nash_batch",
    "ilqg_receding_horizon_shift_batch", "ilqg_default_solve_options", "ilqg_solve_batch_ex", "ilqg_solve_state_batch",
    "ilqg_set_scratch", "ilqg_problem_last_schedule", "ilqg_copy_bandwidth", "ilqg_problem_row_program", "ilqg_row_program_build",
]


class IlqgError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("ilqg status %d: %s" % (status, msg))
        self.status = status


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libilqg_hip.so is not built (run __graft_entry__.build()); "
                               "the HIP path has no CPU fallback")
        # torch ships its own libamdhip64; load (and initialise) it first so that both
        # torch and libilqg_hip.so bind to the same HIP runtime instance.
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
        _LIB = C.CDLL(LIB_PATH)
        _LIB.ilqg_last_error.restype = C.c_char_p
        if _LIB.ilqg_abi_version() != abi.ABI_VERSION:  # a stale .so next to newer struct mirrors corrupts silently
            raise RuntimeError("libilqg_hip.so has ABI version %d, the Python mirrors expect %d: rebuild "
                               "(__graft_entry__.build())" % (_LIB.ilqg_abi_version(), abi.ABI_VERSION))
    return _LIB


def _check(rc):
    if rc != 0:
        raise IlqgError(rc, lib().ilqg_last_error().decode())


def torch_dtype(dtype):
    import torch
    return torch.float32 if dtype == abi.F32 else torch.float64


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev(a, dtype):
    import torch
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.to(device="cuda", dtype=torch_dtype(dtype)).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch_dtype(dtype), device="cuda").contiguous()


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def copy_bandwidth_gbs(nbytes=1 << 30, reps=10):
    """Measured streaming-copy bandwidth of this GPU in GB/s (read + write bytes of ilqg_copy_bandwidth over HIP-event
    time, best of `reps` launches after one warm-up): the roofline's measured denominator."""
    import torch
    nbytes = int(nbytes) & ~15
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    src.fill_(1)
    _check(lib().ilqg_copy_bandwidth(_ptr(dst), _ptr(src), C.c_size_t(nbytes), _stream()))
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _check(lib().ilqg_copy_bandwidth(_ptr(dst), _ptr(src), C.c_size_t(nbytes), _stream()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    assert bool((dst[:: max(1, nbytes // 4096)] == 1).all())
    return 2.0 * nbytes / (best * 1e-3) / 1e9


def device_info():
    name = C.create_string_buffer(256)
    cus = C.c_int32(0)
    _check(lib().ilqg_device_info(name, 256, C.byref(cus)))
    return name.value.decode(), cus.value


def selftest_mfma(dtype, X, Y, Cm):
    """X^T Y + C (16x16, numpy, row/col indexed [i, j]) through the MFMA tile path."""
    import torch
    cm = lambda a: _dev(np.ascontiguousarray(np.asarray(a).T), dtype)  # noqa: E731  column-major image
    Xd, Yd, Cd = cm(X), cm(Y), cm(Cm)
    out = torch.empty_like(Xd)
    _check(lib().ilqg_selftest_mfma(dtype, _ptr(Xd), _ptr(Yd), _ptr(Cd), _ptr(out), _stream()))
    return out.cpu().numpy().T


def set_scratch(buffer=None):
    """ilqg_set_scratch: hand the stand-alone entry points a caller-owned device buffer (a torch tensor; None returns
    to the library's own grow-only allocation).  The caller keeps the tensor alive while it is installed."""
    if buffer is None:
        _check(lib().ilqg_set_scratch(None, C.c_size_t(0)))
    else:
        _check(lib().ilqg_set_scratch(C.c_void_p(buffer.data_ptr()), C.c_size_t(buffer.numel() * buffer.element_size())))


def lq_feedback(dims, A, Bm, Q, l, R, r, pairs, x0=None, want_dx=True, open_loop=False, want_costates=False):
    """ilqg_lq_feedback_batch / ilqg_lq_openloop_batch on device tensors (numpy inputs are uploaded).
    Returns (P, alpha, dx) or, with want_costates, (P, alpha, dx, costates [B][T][N][n])."""
    import torch
    dt = dims.dtype
    B, T, n, N = dims.batch, dims.T, dims.n, dims.num_players
    m = sum(dims.udim[i] for i in range(N))
    A, Bm, Q, l, R, r, x0 = [_dev(v, dt) for v in (A, Bm, Q, l, R, r, x0)]
    P = torch.empty((B, T, m * n), dtype=torch_dtype(dt), device="cuda")
    alpha = torch.empty((B, T, m), dtype=torch_dtype(dt), device="cuda")
    dx = torch.empty((B, T, n), dtype=torch_dtype(dt), device="cuda") if (want_dx or want_costates) else None
    co = torch.empty((B, T, N, n), dtype=torch_dtype(dt), device="cuda") if want_costates else None
    fn = lib().ilqg_lq_openloop_batch if open_loop else lib().ilqg_lq_feedback_batch
    _check(fn(C.byref(dims), _ptr(A), _ptr(Bm), _ptr(Q), _ptr(l), _ptr(R), _ptr(r),
                                        abi.make_pairs(pairs), len(pairs), _ptr(x0), _ptr(P), _ptr(alpha), _ptr(dx),
                                        _ptr(co), _stream()))
    return (P, alpha, dx, co) if want_costates else (P, alpha, dx)


def row_program_build(spec, dtype=abi.F64):
    """ilqg_row_program_build: (int32 words of the row program the library builds for `spec`, id of the registered
    structure it matches or 0) — host only, no device needed."""
    import numpy as np
    desc, keep = spec.build(dtype)
    n, sid = C.c_int32(0), C.c_int32(0)
    _check(lib().ilqg_row_program_build(C.byref(desc), None, 0, C.byref(n), C.byref(sid)))
    w = np.zeros(n.value, np.int32)
    _check(lib().ilqg_row_program_build(C.byref(desc), w.ctypes.data_as(C.c_void_p), n.value, C.byref(n), C.byref(sid)))
    del keep
    return w, int(sid.value)


class Problem:
    """Owns an ilqg_problem* (device tables of one reference `Problem`)."""

    def __init__(self, spec, dtype):
        self.spec, self.dtype = spec, dtype
        self.desc, self._keep = spec.build(dtype)
        self.h = C.c_void_p()
        _check(lib().ilqg_problem_create(C.byref(self.desc), C.byref(self.h)))
        self.n, self.m, self.N, self.T = spec.n, spec.m, len(spec.subsystems), spec.T
        arr = (abi.Pair * 64)()
        npairs = C.c_int32(0)
        _check(lib().ilqg_problem_pairs(self.h, arr, C.byref(npairs)))
        self.pairs = [(arr[q].i, arr[q].j) for q in range(npairs.value)]
        self.Rsz = sum(spec.udims[j] ** 2 for _, j in self.pairs)
        self.rsz = sum(spec.udims[j] for _, j in self.pairs)
        self._ws = None
        # ilqg_solve_options::single_wave_sweep of every solve through this object that does not say otherwise (None =
        # the library's choice, which depends on the batch size: tests that compare a slice with its batch bit for bit
        # pin it)
        self.single_wave_sweep = None

    def __del__(self):
        try:
            if self.h:
                lib().ilqg_problem_destroy(self.h)
        except Exception:
            pass

    def _empty(self, *shape):
        import torch
        return torch.empty(shape, dtype=torch_dtype(self.dtype), device="cuda")

    def rollout(self, x0, xs_ref, us_ref, P, alpha, alpha_scale=None):
        x0, xs_ref, us_ref, P, alpha, alpha_scale = [_dev(v, self.dtype) for v in
                                                     (x0, xs_ref, us_ref, P, alpha, alpha_scale)]
        B = x0.shape[0]
        xs, us = self._empty(B, self.T, self.n), self._empty(B, self.T, self.m)
        _check(lib().ilqg_rollout_batch(self.h, B, _ptr(x0), _ptr(xs_ref), _ptr(us_ref), _ptr(P), _ptr(alpha),
                                        _ptr(alpha_scale), _ptr(xs), _ptr(us), None, _stream()))
        return xs, us

    def linearize(self, xs, us):
        xs, us = _dev(xs, self.dtype), _dev(us, self.dtype)
        B = xs.shape[0]
        A, Bm = self._empty(B, self.T, self.n * self.n), self._empty(B, self.T, self.n * self.m)
        _check(lib().ilqg_linearize_batch(self.h, B, _ptr(xs), _ptr(us), _ptr(A), _ptr(Bm), None, _stream()))
        return A, Bm

    def quadraticize(self, xs, us, lambdas=None, mu=None, t_extreme=None):
        import torch
        xs, us, lambdas, mu = [_dev(v, self.dtype) for v in (xs, us, lambdas, mu)]
        te = None if t_extreme is None else torch.as_tensor(np.ascontiguousarray(t_extreme, dtype=np.int32),
                                                            device="cuda")
        B = xs.shape[0]
        Q, l = self._empty(B, self.T, self.N, self.n * self.n), self._empty(B, self.T, self.N, self.n)
        R, r = self._empty(B, self.T, self.Rsz), self._empty(B, self.T, self.rsz)
        _check(lib().ilqg_quadraticize_batch(self.h, B, _ptr(xs), _ptr(us), _ptr(lambdas), _ptr(mu), _ptr(te),
                                             _ptr(Q), _ptr(l), _ptr(R), _ptr(r), None, _stream()))
        return Q, l, R, r

    def total_costs(self, xs, us, t_extreme=None):
        import torch
        xs, us = _dev(xs, self.dtype), _dev(us, self.dtype)
        B = xs.shape[0]
        costs = self._empty(B, self.N)
        te = torch.zeros((B, self.N), dtype=torch.int32, device="cuda") if t_extreme is None else \
            torch.as_tensor(np.ascontiguousarray(t_extreme, dtype=np.int32), device="cuda")
        _check(lib().ilqg_total_costs_batch(self.h, B, _ptr(xs), _ptr(us), _ptr(costs), _ptr(te), None, _stream()))
        return costs, te

    def workspace_bytes(self, batch):
        b = C.c_uint64(0)
        _check(lib().ilqg_workspace_bytes(self.h, batch, C.byref(b)))
        return b.value

    def alloc_solve_buffers(self, batch):
        import torch
        z = lambda *s: torch.zeros(s, dtype=torch_dtype(self.dtype), device="cuda")  # noqa: E731
        return dict(xs=z(batch, self.T, self.n), us=z(batch, self.T, self.m), P=z(batch, self.T, self.m * self.n),
                    alpha=z(batch, self.T, self.m), costs=z(batch, self.N),
                    iters=torch.zeros(batch, dtype=torch.int32, device="cuda"),
                    status=torch.zeros(batch, dtype=torch.int32, device="cuda"),
                    converged=torch.zeros(batch, dtype=torch.int32, device="cuda"),
                    ws=torch.empty(self.workspace_bytes(batch), dtype=torch.uint8, device="cuda"))

    def solve(self, x0, bufs=None, fixed_iters=0, augmented_lagrangian=False, forced_steps=None, split_trial=None,
              handoff=None, probe=None, counted=None, resume=False, active=None, compact_rows=None, round_bursts=None,
              log_capacity=0, log_strategies=False, max_runtime=0.0, generic_kernels=None, probe_first=0, single_wave_sweep=None, adjoint_expected_decrease=None,
              deterministic=False, static_rows=None, padded_sweep=None, probe_lanes=None):
        """ilqg_solve_batch_ex. `bufs` (from alloc_solve_buffers) carries the warm start in and the solution out; zero
        warm start if omitted.  forced_steps [B][fixed_iters]: test mode, the given step sizes instead of the line
        search.  split_trial / handoff / probe / counted / compact_rows: None = let the library choose, True / False = force the
        schedule (same results either way).  log_capacity > 0: every iterate of every instance is logged
        (ilqg_iterate_log) into bufs["log"] = dict(xs, us, costs[, P, alpha], count).  max_runtime > 0: the anytime exit."""
        import torch
        x0 = _dev(x0, self.dtype)
        B = x0.shape[0]
        if bufs is None:
            bufs = self.alloc_solve_buffers(B)
        o = abi.SolveOptions()
        lib().ilqg_default_solve_options(C.byref(o))
        o.fixed_iters = int(fixed_iters)
        o.augmented_lagrangian = 1 if augmented_lagrangian else 0
        o.resume = 1 if resume else 0
        o.deterministic = 1 if deterministic else 0
        o.active = None if active is None else active.data_ptr()
        fs = None
        if forced_steps is not None:
            fs = _dev(forced_steps, self.dtype)
            assert tuple(fs.shape) == (B, fixed_iters)
            o.forced_steps = fs.data_ptr()
        tri = lambda v: abi.CHOICE_AUTO if v is None else (abi.CHOICE_ON if v else abi.CHOICE_OFF)  # noqa: E731
        o.split_trial, o.handoff, o.probe, o.counted = tri(split_trial), tri(handoff), tri(probe), tri(counted)
        o.compact_rows = tri(compact_rows)
        o.round_bursts = tri(round_bursts)
        o.generic_kernels = tri(generic_kernels)
        o.probe_first = int(probe_first)
        o.single_wave_sweep = tri(self.single_wave_sweep if single_wave_sweep is None else single_wave_sweep)
        o.adjoint_expected_decrease = tri(adjoint_expected_decrease)
        o.static_rows = tri(static_rows)
        o.padded_sweep = tri(padded_sweep)
        o.probe_lanes = tri(probe_lanes)
        o.max_runtime = float(max_runtime)
        il = None
        if log_capacity > 0:
            cap = int(log_capacity)
            log = dict(xs=self._empty(B, cap, self.T, self.n), us=self._empty(B, cap, self.T, self.m),
                       costs=self._empty(B, cap, self.N), count=torch.zeros(B, dtype=torch.int32, device="cuda"))
            if log_strategies:
                log["P"] = self._empty(B, cap, self.T, self.m * self.n)
                log["alpha"] = self._empty(B, cap, self.T, self.m)
            il = abi.IterateLog(log["xs"].data_ptr(), log["us"].data_ptr(), log["costs"].data_ptr(),
                                log["P"].data_ptr() if log_strategies else None,
                                log["alpha"].data_ptr() if log_strategies else None, log["count"].data_ptr(), cap)
            o.iterate_log = C.pointer(il)
            bufs["log"] = log
        _check(lib().ilqg_solve_batch_ex(self.h, B, _ptr(x0), _ptr(bufs["xs"]), _ptr(bufs["us"]), _ptr(bufs["P"]),
                                         _ptr(bufs["alpha"]), _ptr(bufs["costs"]), _ptr(bufs["iters"]),
                                         _ptr(bufs["status"]), _ptr(bufs["converged"]), _ptr(bufs["ws"]), C.byref(o),
                                         _stream()))
        return bufs

    def solve_state(self, bufs, augmented_lagrangian=False):
        """ilqg_solve_state_batch: dict(last_merit, expected_decrease, step, backtracks) of the last solve on bufs."""
        import torch
        B = bufs["iters"].shape[0]
        out = dict(last_merit=self._empty(B), expected_decrease=self._empty(B), step=self._empty(B),
                   backtracks=torch.zeros(B, dtype=torch.int32, device="cuda"))
        _check(lib().ilqg_solve_state_batch(self.h, B, _ptr(bufs["ws"]), int(augmented_lagrangian),
                                            _ptr(out["last_merit"]), _ptr(out["expected_decrease"]),
                                            _ptr(out["step"]), _ptr(out["backtracks"]), _stream()))
        return out

    def row_program(self):
        """ilqg_problem_row_program: (int32 words of the row program as built, id of the registered structure or 0)."""
        import numpy as np
        n, sid = C.c_int32(0), C.c_int32(0)
        _check(lib().ilqg_problem_row_program(self.h, None, 0, C.byref(n), C.byref(sid)))
        w = np.zeros(n.value, np.int32)
        _check(lib().ilqg_problem_row_program(self.h, w.ctypes.data_as(C.c_void_p), n.value, C.byref(n), C.byref(sid)))
        return w, int(sid.value)

    def last_schedule(self):
        """ilqg_problem_last_schedule: the ILQG_SCHEDULE_* bits of the last solve on this problem."""
        v = C.c_int32(0)
        _check(lib().ilqg_problem_last_schedule(self.h, C.byref(v)))
        return v.value

    def receding_horizon_shift(self, x0, t0, planner_runtime, plan_t0, bufs):
        """ilqg_receding_horizon_shift_batch: turns the solution in `bufs` (xs, us, P, alpha) into the warm start
        of the next receding-horizon solve, in place.  Returns (x0_next, first_step, new_plan_t0)."""
        import torch
        x0 = _dev(x0, self.dtype)
        B = x0.shape[0]
        x0_next = torch.empty_like(x0)
        first = torch.zeros(B, dtype=torch.int32, device="cuda")
        new_t0 = C.c_double(0.0)
        _check(lib().ilqg_receding_horizon_shift_batch(self.h, B, _ptr(x0), C.c_double(t0), C.c_double(planner_runtime),
                                                       C.c_double(plan_t0), _ptr(bufs["xs"]), _ptr(bufs["us"]),
                                                       _ptr(bufs["P"]), _ptr(bufs["alpha"]), _ptr(x0_next), _ptr(first),
                                                       C.byref(new_t0), _stream()))
        return x0_next, first, new_t0.value

    # ---- receding-horizon harness (examples/receding_horizon_simulator.h) ----
    def new_plan(self, batch, rows=None):
        """Empty SolutionSplicer state on the device: dict(xs, us, P, alpha, len, t0)."""
        import torch
        rows = self.T + 5 if rows is None else rows
        z = lambda *shape: torch.zeros(*shape, dtype=torch_dtype(self.dtype), device="cuda")
        return dict(xs=z(batch, rows, self.n), us=z(batch, rows, self.m), P=z(batch, rows, self.m * self.n),
                    alpha=z(batch, rows, self.m), len=torch.zeros(batch, dtype=torch.int32, device="cuda"),
                    t0=torch.zeros(batch, dtype=torch.float64, device="cuda"))

    def plan_integrate(self, plan, t_from, t_to, must_contain, x, active):
        """ilqg_plan_integrate_batch: x and active (device tensors) are updated in place."""
        _check(lib().ilqg_plan_integrate_batch(self.h, x.shape[0], plan["xs"].shape[1], _ptr(plan["xs"]),
                                               _ptr(plan["us"]), _ptr(plan["P"]), _ptr(plan["alpha"]),
                                               _ptr(plan["len"]), _ptr(plan["t0"]), C.c_double(t_from),
                                               C.c_double(t_to), C.c_double(must_contain), _ptr(x), _ptr(active),
                                               _stream()))

    def receding_horizon_sync(self, plan, x, t, planner_runtime, bufs, active):
        """ilqg_receding_horizon_sync_batch: fills bufs (xs, us, P, alpha) with the next solve's warm start.
        Returns (x0_next, solve_t0, first_step) as device tensors."""
        import torch
        B = x.shape[0]
        x0_next = torch.zeros_like(x)
        solve_t0 = torch.zeros(B, dtype=torch.float64, device="cuda")
        first = torch.zeros(B, dtype=torch.int32, device="cuda")
        _check(lib().ilqg_receding_horizon_sync_batch(
            self.h, B, plan["xs"].shape[1], _ptr(plan["xs"]), _ptr(plan["us"]), _ptr(plan["P"]), _ptr(plan["alpha"]),
            _ptr(plan["len"]), _ptr(plan["t0"]), _ptr(x), C.c_double(t), C.c_double(planner_runtime),
            _ptr(bufs["xs"]), _ptr(bufs["us"]), _ptr(bufs["P"]), _ptr(bufs["alpha"]), _ptr(x0_next), _ptr(solve_t0),
            _ptr(first), _ptr(active), _stream()))
        return x0_next, solve_t0, first

    def solution_splice(self, plan, bufs, solve_t0, converged=None, active=None):
        """ilqg_solution_splice_batch: constructs (plan len 0) or splices the solution in bufs into the plan."""
        _check(lib().ilqg_solution_splice_batch(
            self.h, plan["xs"].shape[0], plan["xs"].shape[1], _ptr(plan["xs"]), _ptr(plan["us"]), _ptr(plan["P"]),
            _ptr(plan["alpha"]), _ptr(plan["len"]), _ptr(plan["t0"]), _ptr(bufs["xs"]), _ptr(bufs["us"]),
            _ptr(bufs["P"]), _ptr(bufs["alpha"]), _ptr(solve_t0), None if converged is None else _ptr(converged),
            None if active is None else _ptr(active), _stream()))

    def solve_again(self, x0, bufs, augmented_lagrangian=False, active=None):
        """ilqg_solve_again_batch: the next Solve() of the solver object whose previous call used bufs['ws']."""
        x0 = _dev(x0, self.dtype)
        if self.single_wave_sweep is not None:  # a pinned schedule: the same call through the options struct
            return self.solve(x0, bufs, augmented_lagrangian=augmented_lagrangian, resume=True, active=active)
        _check(lib().ilqg_solve_again_batch(self.h, x0.shape[0], _ptr(x0), _ptr(bufs["xs"]), _ptr(bufs["us"]),
                                            _ptr(bufs["P"]), _ptr(bufs["alpha"]), _ptr(bufs["costs"]),
                                            _ptr(bufs["iters"]), _ptr(bufs["status"]), _ptr(bufs["converged"]),
                                            _ptr(bufs["ws"]), int(augmented_lagrangian),
                                            None if active is None else _ptr(active), _stream()))
        return bufs

    def receding_horizon_simulate(self, x_init, final_time, planner_runtime, extra_time=0.25, solve_time=0.25,
                                  augmented_lagrangian=False, max_records=64, on_record=None):
        """RecedingHorizonSimulator (src/receding_horizon_simulator.cpp:64-137) for a batch of initial states, with
        the wall clock replaced by a fixed simulated `solve_time` per solver call: every instance shares the
        clock, keeps its own spliced plan and drops out on its own (ContainsTime false, invalid times, or a failed
        first solve — the reference CHECKs success there).  Everything stays on the device; `on_record(r, info)`
        is called after every solver call with device tensors (info: t_call, active, x_measured, x0, solve_t0,
        first_step, bufs).  Returns dict(x, plan, active, num_records [B] tensor, calls)."""
        import torch
        x = _dev(x_init, self.dtype).clone()
        B = x.shape[0]
        dt = self.spec.dt
        bufs = self.alloc_solve_buffers(B)
        plan = self.new_plan(B)
        active = torch.ones(B, dtype=torch.int32, device="cuda")
        num_records = torch.ones(B, dtype=torch.int32, device="cuda")
        self.solve(x, bufs, augmented_lagrangian=augmented_lagrangian)
        solve_t0 = torch.zeros(B, dtype=torch.float64, device="cuda")
        if on_record:
            on_record(0, dict(t_call=0.0, active=active.clone(), x_measured=x, x0=x, solve_t0=solve_t0,
                              first_step=None, bufs=bufs))
        self.solution_splice(plan, bufs, solve_t0)
        active &= bufs["status"]  # CHECK(success) after the first call (:77)
        t = 0.0
        calls = 1
        while calls < max_records:
            t += extra_time
            if t >= final_time:
                break
            self.plan_integrate(plan, t - extra_time, t, t + planner_runtime + dt, x, active)
            x0_next, solve_t0, first = self.receding_horizon_sync(plan, x, t, planner_runtime, bufs, active)
            if int(active.sum().item()) == 0:
                break
            x0_all = torch.where(active.bool()[:, None], x0_next, bufs["xs"][:, 0, :])
            self.solve_again(x0_all, bufs, augmented_lagrangian=augmented_lagrangian, active=active)
            num_records += active
            if on_record:
                on_record(calls, dict(t_call=t, active=active.clone(), x_measured=x.clone(), x0=x0_all,
                                      solve_t0=solve_t0, first_step=first, bufs=bufs))
            calls += 1
            t += solve_time
            if t >= final_time:
                break
            self.plan_integrate(plan, t - solve_time, t, t, x, active)
            self.solution_splice(plan, bufs, solve_t0, converged=bufs["converged"], active=active)
        return dict(x=x, plan=plan, active=active, num_records=num_records, calls=calls)

    # ---- equilibrium checks ----
    def strategy_costs(self, x0, xs, us, P, alpha, open_loop=False, euler=True):
        """ilqg_strategy_costs_batch -> [B][N] device tensor."""
        a = [_dev(v, self.dtype) for v in (x0, xs, us, P, alpha)]
        costs = self._empty(a[0].shape[0], self.N)
        _check(lib().ilqg_strategy_costs_batch(self.h, a[0].shape[0], *[_ptr(v) for v in a], int(open_loop), int(euler),
                                               _ptr(costs), _stream()))
        return costs

    def check_local_nash(self, x0, xs, us, P, alpha, max_perturbation, open_loop=False):
        """ilqg_check_local_nash_batch -> (is_nash [B] int32, margin [B]) device tensors."""
        import torch
        a = [_dev(v, self.dtype) for v in (x0, xs, us, P, alpha)]
        B = a[0].shape[0]
        ok = torch.zeros(B, dtype=torch.int32, device="cuda")
        margin = self._empty(B)
        _check(lib().ilqg_check_local_nash_batch(self.h, B, *[_ptr(v) for v in a], C.c_double(max_perturbation),
                                                 int(open_loop), _ptr(ok), _ptr(margin), _stream()))
        return ok, margin

    def check_sufficient_nash(self, xs, us):
        """ilqg_check_sufficient_nash_batch -> is_psd [B] int32 device tensor."""
        import torch
        xs, us = _dev(xs, self.dtype), _dev(us, self.dtype)
        ok = torch.zeros(xs.shape[0], dtype=torch.int32, device="cuda")
        _check(lib().ilqg_check_sufficient_nash_batch(self.h, xs.shape[0], _ptr(xs), _ptr(us), _ptr(ok), _stream()))
        return ok
