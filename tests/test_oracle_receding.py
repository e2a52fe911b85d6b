"""CPU checks of the oracle's receding-horizon harness (oracle/ilqg_oracle.hpp, restating
src/problem.cpp:64-186, src/solution_splicer.cpp:56-129, src/multi_player_integrable_system.cpp:54-155 and
src/receding_horizon_simulator.cpp:64-137).  The reference has no tests for these files; what can be pinned
without it are the invariants its own CHECKs state, the agreement of the two restatements of
SetUpNextRecedingHorizon, and the row bookkeeping of the splicer."""
import numpy as np

from ilqgames_amd import abi, examples


def _spec():
    spec = examples.modified_three_player_intersection()
    spec.params.initial_alpha_scaling = 0.5
    spec.params.expected_decrease_fraction = 0.01
    spec.params.convergence_tolerance = 0.1
    return spec


def test_sync_on_a_T_row_plan_equals_the_shared_time_base_shift(oracle):
    """receding_horizon_sync (per-instance plans) and receding_horizon_shift (one time base for the batch) are two
    restatements of Problem::SetUpNextRecedingHorizon; on plans of exactly T rows they must coincide bit for bit."""
    spec = _spec()
    B = 4
    op = oracle.OracleProblem(spec)
    x0 = examples.jittered_x0(spec, B, seed=2)
    s0 = op.solve(abi.F64, x0, fixed_iters=3)
    plan = op.new_plan(abi.F64, B)
    op.solution_splice(abi.F64, plan, s0, np.zeros(B))
    for t, rt in [(0.33, 0.25), (0.0, 0.1), (1.07, 0.0), (2.5, 0.4)]:
        x = s0["xs"][:, int(t / spec.dt), :] + 0.05 * np.random.default_rng(3).standard_normal((B, op.n))
        act = np.ones(B, np.int32)
        a = op.receding_horizon_sync(abi.F64, plan, x, t, rt, act)
        b = op.receding_horizon_shift(abi.F64, x, t, rt, 0.0, s0["xs"], s0["us"], s0["P"], s0["alpha"])
        assert act.all()
        assert np.array_equal(a["first_step"], b["first_step"])
        assert np.all(a["t0"] == b["new_plan_t0"])
        assert np.array_equal(a["x0"], b["x0_next"])
        for k in ("xs", "us", "P", "alpha"):
            assert np.array_equal(a[k], b[k]), k
        assert np.all(np.abs(t + rt - a["t0"]) <= spec.dt + 1e-12)  # CHECK_LE at src/problem.cpp:123


def test_splice_keeps_five_rows_and_moves_the_start_time(oracle):
    spec = _spec()
    B = 6
    op = oracle.OracleProblem(spec)
    T = spec.T
    rng = np.random.default_rng(0)
    mk = lambda: dict(xs=rng.standard_normal((B, T, op.n)), us=rng.standard_normal((B, T, op.m)),
                      P=rng.standard_normal((B, T, op.m * op.n)), alpha=rng.standard_normal((B, T, op.m)))
    old, new = mk(), mk()
    plan = op.new_plan(abi.F64, B)
    op.solution_splice(abi.F64, plan, old, np.full(B, 1.5))
    assert np.all(plan["len"] == T) and np.all(plan["t0"] == 1.5)
    steps = np.array([0, 2, 5, 6, 30, 7])
    conv = np.array([1, 1, 1, 1, 1, 0], np.int32)
    op.solution_splice(abi.F64, plan, new, 1.5 + steps * spec.dt, converged=conv)
    for b, k in enumerate(steps):
        if not conv[b]:
            assert plan["len"][b] == T and np.array_equal(plan["xs"][b, :T], old["xs"][b])
            continue
        keep = min(k, 5)
        assert plan["len"][b] == T + keep
        assert abs(plan["t0"][b] - (1.5 + (k - keep) * spec.dt)) < 1e-12
        for key in ("xs", "us", "P", "alpha"):
            assert np.array_equal(plan[key][b, :keep], old[key][b, k - keep:k]), (b, key)
            assert np.array_equal(plan[key][b, keep:keep + T], new[key][b]), (b, key)
    # a solution that starts before the stored plan (the reference CHECK-aborts) leaves the plan alone
    before = {k: v.copy() for k, v in plan.items()}
    op.solution_splice(abi.F64, plan, old, plan["t0"] - 0.3, converged=np.ones(B, np.int32))
    for k in before:
        assert np.array_equal(before[k], plan[k])


def test_interval_integration_follows_the_plan_and_the_reference_quirk(oracle):
    """With zero gains and the plan's own controls, integrating along the plan from a plan state lands on plan
    states; when t0 sits exactly on the plan's start, Integrate(t0, t, ...) as written in the reference skips the
    first step (no IntegrateToNextTimeStep, whole steps begin at current_timestep + 1)."""
    spec = _spec()
    B = 2
    op = oracle.OracleProblem(spec)
    x0 = examples.jittered_x0(spec, B, seed=4)
    zeros = lambda *s: np.zeros(s)
    # a consistent plan: open-loop rollout of some controls
    us = 0.3 * np.random.default_rng(1).standard_normal((B, spec.T, op.m))
    xs, _ = op.rollout(abi.F64, x0, zeros(B, spec.T, op.n), us, zeros(B, spec.T, op.m * op.n), zeros(B, spec.T, op.m))
    plan = op.new_plan(abi.F64, B)
    sol = dict(xs=xs, us=us, P=zeros(B, spec.T, op.m * op.n), alpha=zeros(B, spec.T, op.m))
    op.solution_splice(abi.F64, plan, sol, np.full(B, 2.0))
    act = np.ones(B, np.int32)
    x = xs[:, 3, :].copy()  # state at t = 2.3, slightly inside step 3: partial step, whole steps, partial step
    op.plan_integrate(abi.F64, plan, 2.3 + 1e-9, 2.9, 2.9, x, act)
    assert act.all() and np.abs(x - xs[:, 9, :]).max() < 1e-6
    x = xs[:, 0, :].copy()  # t0 == plan start: steps 1 .. 5 only, i.e. 0.5 s of motion instead of 0.6 s
    op.plan_integrate(abi.F64, plan, 2.0, 2.6, 2.6, x, act)
    shifted = np.concatenate([us[:, 1:, :], us[:, -1:, :]], axis=1)  # controls of steps 1, 2, ...
    expect, _ = op.rollout(abi.F64, xs[:, 0, :], zeros(B, spec.T, op.n), shifted, zeros(B, spec.T, op.m * op.n),
                           zeros(B, spec.T, op.m))
    assert np.abs(x - expect[:, 5, :]).max() < 1e-9
    # a plan that does not contain `must_contain` drops the instance and leaves its state alone
    x = xs[:, 0, :].copy()
    act = np.ones(B, np.int32)
    op.plan_integrate(abi.F64, plan, 2.0, 2.6, 2.0 + spec.T * spec.dt + 0.01, x, act)
    assert not act.any() and np.array_equal(x, xs[:, 0, :])


def test_simulated_loop_records_are_consistent(oracle):
    spec = _spec()
    B = 4
    op = oracle.OracleProblem(spec)
    x0 = examples.jittered_x0(spec, B, seed=3)
    o = op.receding_horizon_simulate(abi.F64, x0, 3.0, 0.25, max_records=16, threads=4)
    assert o["num_records"].min() >= 4
    for b in range(B):
        R = o["num_records"][b]
        assert o["t_call"][b, 0] == 0.0 and o["first_step"][b, 0] == -1
        # the simulator's clock: 0.25 s of "extra time" before and 0.25 s of simulated solve time after each call
        assert np.allclose(np.diff(o["t_call"][b, 1:R]), 0.5) and abs(o["t_call"][b, 1] - 0.25) < 1e-12
        # every window starts within one step of (call time + planner runtime)   (src/problem.cpp:123)
        assert np.all(np.abs(o["t_call"][b, 1:R] + 0.25 - o["plan_t0"][b, 1:R]) <= spec.dt + 1e-9)
        # each solve starts from its stitched initial state
        assert np.array_equal(o["xs"][b, :R, 0, :], o["x0"][b, :R])
        # the ego block of the stitched state comes from the old plan, the others from the measured state's
        # forward integration: the measured state is within a few metres of the new initial state
        assert np.abs(o["x0"][b, 1:R] - o["x_measured"][b, 1:R]).max() < 10.0
    assert (o["plan"]["len"] >= spec.T).all() and (o["plan"]["len"] <= spec.T + 5).all()
    # same inputs, same answer (no hidden state between calls of the harness)
    o2 = op.receding_horizon_simulate(abi.F64, x0, 3.0, 0.25, max_records=16, threads=1)
    assert np.array_equal(o["xs"], o2["xs"]) and np.array_equal(o["iters"], o2["iters"])


def test_second_solve_uses_the_previous_merit_value(oracle):
    spec = _spec()
    spec.params.max_solver_iters = 40
    B = 4
    op = oracle.OracleProblem(spec)
    x0 = examples.jittered_x0(spec, B, seed=17)
    last = np.full(B, np.inf)
    zeros = [np.zeros(s) for s in ((B, op.T, op.n), (B, op.T, op.m), (B, op.T, op.m * op.n), (B, op.T, op.m))]
    r1 = op.solve_resume(abi.F64, x0, *zeros, last)
    ref = op.solve(abi.F64, x0)
    assert np.array_equal(r1["xs"], ref["xs"]) and np.isfinite(last).all()  # first call == a fresh solver
    # a tiny carried merit value makes every step of the next call look like an increase: the line search
    # exhausts its back-tracking budget on the first iteration and the call reports failure
    tiny = np.full(B, 1e-9)
    r2 = op.solve_resume(abi.F64, x0 + 0.2, r1["xs"], r1["us"], r1["P"], r1["alpha"], tiny)
    assert np.all(r2["status"] == 0) and np.all(r2["iters"] == 1) and np.all(tiny == 1e-9)


def test_strategy_costs_of_an_open_loop_plan_equal_the_total_costs_of_its_rollout(oracle):
    """ComputeStrategyCosts (src/compute_strategy_costs.cpp:61-106) with zero gains and zero alphas plays the
    operating point's own controls: with the default RK4 integration the accumulated per-step costs are
    ILQSolver::TotalCosts of the rolled-out trajectory (all players of this example are time-additive)."""
    spec = _spec()
    B = 3
    op = oracle.OracleProblem(spec)
    x0 = examples.jittered_x0(spec, B, seed=6)
    us = 0.2 * np.random.default_rng(2).standard_normal((B, spec.T, op.m))
    zP, za = np.zeros((B, spec.T, op.m * op.n)), np.zeros((B, spec.T, op.m))
    xs, _ = op.rollout(abi.F64, x0, np.zeros((B, spec.T, op.n)), us, zP, za)
    total, _ = op.total_costs(abi.F64, xs, us)
    got = op.strategy_costs(abi.F64, x0, xs, us, zP, za, open_loop=False, euler=False)
    assert np.allclose(got, total, rtol=1e-12)
    # Euler integration is a different trajectory, hence different costs; zero perturbation decides nothing
    assert not np.allclose(op.strategy_costs(abi.F64, x0, xs, us, zP, za, euler=True), total, rtol=1e-6)
    ok, margin = op.check_local_nash(abi.F64, x0, xs, us, zP, za, 0.0)
    assert np.all(ok == 1) and np.all(margin == 0.0)
    ok, margin = op.check_local_nash(abi.F64, x0, xs, us, zP, za, 0.05)
    assert np.all(ok == 0) and np.all(margin < 0)  # random controls are no equilibrium


def test_jacobi_min_eigenvalue_matches_numpy(oracle):
    """The oracle's eigenvalue routine (behind CheckSufficientLocalNashEquilibrium) against numpy.linalg.eigvalsh."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 4, 6, 14, 24):
        for _ in range(3):
            a = rng.standard_normal((n, n))
            a = a + a.T
            assert abs(oracle.min_eigenvalue(a) - np.linalg.eigvalsh(a)[0]) < 1e-10 * max(1.0, np.abs(a).max())
    g = rng.standard_normal((8, 3))
    assert abs(oracle.min_eigenvalue(g @ g.T)) < 1e-12  # rank-deficient PSD: smallest eigenvalue 0


def test_nearest_plan_state_of_two_player_unicycle_4d_is_by_position(oracle):
    """TwoPlayerUnicycle4D overrides DistanceBetween with the squared distance of (px, py) only
    (include/ilqgames/dynamics/two_player_unicycle_4d.h:141-147) — it does NOT inherit the whole-state default of
    multi_player_integrable_system.h:113.  Problem::SyncToExistingProblem (src/problem.cpp:105-110) therefore picks the
    plan state nearest in position.  The plan below is built so that the two metrics disagree: the state nearest in
    position (row 10) has a heading 1 rad away from the measured one, while row 3 matches the heading exactly and
    is 0.35 m away — position says 10, the whole-state norm would say 3.  (The expected index comes from the
    reference's rule, not from the device: tests/test_gpu_receding.py holds the device to the oracle.)"""
    spec = examples.two_player_unicycle_4d_scene()
    op = oracle.OracleProblem(spec)
    T, n, m = spec.T, op.n, op.m
    assert n == 4
    xs = np.zeros((1, T, n))
    xs[0, :, 0] = 0.05 * np.arange(T)   # px: 5 cm per row
    # v = 0 everywhere and zero controls / gains: whatever SetUpNextRecedingHorizon integrates, the state stays put
    xs[0, 3, 2] = 1.0                   # heading of row 3 only
    us = np.zeros((1, T, m))
    P = np.zeros((1, T, m * n))
    alpha = np.zeros((1, T, m))
    x = np.array([[0.5, 0.0, 1.0, 0.0]])  # on row 10's position, with row 3's heading
    out = op.receding_horizon_shift(abi.F64, x, 0.0, 0.0, 0.0, xs, us, P, alpha)
    d_pos = (xs[0, :, 0] - 0.5) ** 2 + xs[0, :, 1] ** 2
    d_all = d_pos + (xs[0, :, 2] - 1.0) ** 2 + xs[0, :, 3] ** 2
    assert int(np.argmin(d_pos)) == 10 and int(np.argmin(d_all)) == 3  # the construction
    assert int(out["first_step"][0]) == 10
