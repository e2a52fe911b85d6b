"""GPU parity of the run-time-dimensioned path (ilqg_lq_generic.hpp, rows_chunk / trial parts with NX = 0,
generic_solve in ilqg_api.hip): every shape the library holds no specialised instantiation of — players with different
control dimensions, un-instantiated state dimensions, any horizon — behind the same C-ABI entry points.

Checked against the oracle the way the specialised kernels are: the two sweeps on random games and on the reference-
generated fixtures, the stage kernels on a ConcatenatedDynamicalSystem(Dubins m = 1, Car5D m = 2), whole solves with
forced steps after every iteration, free-running and augmented-Lagrangian solves; and the same kernels forced onto a
shape that HAS a specialised instantiation (ilqg_solve_options::generic_kernels, ilqg_dims::sweep_formulation), where
both device paths must agree with the oracle.  Tolerances: fp64 1e-9 relative, fp32 as in test_gpu_parity.py."""
import numpy as np
import pytest

from ilqgames_amd import abi, examples
from helpers import dims_of, load_golden_lq, random_lq_game, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from ilqgames_amd import hip as h
    return h


def _np(t):
    return t.detach().cpu().numpy()


SHAPES = [  # (n, control dimensions): no specialised instantiation holds any of these
    (5, (2, 1, 2)), (7, (1, 2)), (9, (3, 1, 2, 1)), (3, (2,)), (13, (2, 2, 2, 2, 1)), (32, (2, 2, 2, 2, 2, 2, 2, 2)),
    (12, (2, 2, 2)), (20, (4, 4)),
]


@pytest.mark.parametrize("open_loop", [False, True], ids=["feedback", "open_loop"])
@pytest.mark.parametrize("n,ms", SHAPES, ids=["n%d_m%s" % (n, "".join(map(str, ms))) for n, ms in SHAPES])
@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
@pytest.mark.parametrize("formulation", ["auto", "run_time_dimensioned"])
def test_generic_sweeps_match_oracle(hip, oracle, n, ms, open_loop, dtype, formulation):
    """ilqg_lq_feedback_batch / ilqg_lq_openloop_batch on shapes without an instantiation.  auto: the game embedded in the
    smallest instantiated shape that holds it, on that shape's sweep ((5, (2,1,2)) in (6,3,2), (7, (1,2)) in (8,2,2),
    (12, (2,2,2)) in (14,3,2); the open-loop form when no costates are asked for), the run-time-dimensioned sweeps where
    there is none (four players and more, a control of three or four dimensions); ILQG_SWEEP_GENERIC: those sweeps always."""
    rng = np.random.default_rng(100 * n + len(ms) + (7 if open_loop else 0))
    T, B = (12 if n > 16 else 25), 4
    N = len(ms)
    pairs = ([(i, i) for i in range(N)] + [(i, (i + 1) % N) for i in range(N) if N > 1 and i % 2 == 0])[:16]
    g = random_lq_game(rng, n, list(ms), T, B, pairs=pairs)
    x0 = 0.3 * rng.standard_normal((B, n))
    d = abi.make_dims(n, list(ms), T, B, dtype, adaptive_regularization=not open_loop)
    if formulation != "auto":
        d.sweep_formulation = abi.SWEEP_GENERIC
    Pr, ar, dxr, cor = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], pairs, x0=x0,
                                       open_loop=open_loop, want_costates=True)
    tol = 1e-9 if dtype == abi.F64 else 5e-3
    for want_costates in (True, False):
        out = hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], pairs, x0=x0,
                              open_loop=open_loop, want_costates=want_costates)
        P, alpha, dx = out[:3]
        assert rel_err(_np(P), Pr) < tol and rel_err(_np(alpha), ar) < tol and rel_err(_np(dx), dxr) < tol
        if want_costates:
            assert rel_err(_np(out[3]), cor) < tol
        assert np.all(_np(P)[:, -1] == 0) and np.all(_np(alpha)[:, -1] == 0)
        if open_loop:
            assert np.all(_np(P) == 0)
    # without delta_x: strategies only
    P, alpha, _ = hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], pairs, x0=x0, open_loop=open_loop,
                                  want_dx=False)
    assert rel_err(_np(P), Pr) < tol and rel_err(_np(alpha), ar) < tol


@pytest.mark.parametrize("dims", [(14, 3, 2), (16, 3, 2), (24, 4, 2), (4, 2, 2), (3, 1, 1)])
@pytest.mark.parametrize("open_loop", [False, True], ids=["feedback", "open_loop"])
def test_generic_sweeps_agree_with_the_specialised_ones(hip, oracle, dims, open_loop):
    """ilqg_dims::sweep_formulation = ILQG_SWEEP_GENERIC on shapes that have a matrix-core instantiation: two device
    implementations of one recursion, both against the oracle (and therefore against each other)."""
    n, N, mu = dims
    rng = np.random.default_rng(17 * n + N)
    T, B = 25, 3
    g = random_lq_game(rng, n, [mu] * N, T, B)
    x0 = rng.standard_normal((B, n))
    d = dims_of(g, abi.F64, adaptive=not open_loop)
    Pr, ar, dxr, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0,
                                     open_loop=open_loop)
    outs = []
    for formulation in (abi.CHOICE_AUTO, abi.SWEEP_GENERIC):
        d.sweep_formulation = formulation
        P, alpha, dx = hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0,
                                       open_loop=open_loop)
        assert rel_err(_np(P), Pr) < 1e-9 and rel_err(_np(alpha), ar) < 1e-9 and rel_err(_np(dx), dxr) < 1e-9
        outs.append((_np(P), _np(alpha)))
    assert rel_err(outs[0][0], outs[1][0]) < 1e-9 and rel_err(outs[0][1], outs[1][1]) < 1e-9


def test_only_dimensions_past_the_header_limits_are_unsupported(hip):
    """n <= 32 (ILQG_MAX_XDIM), N <= 8, sum m_i <= 16 run; beyond that the status is ILQG_ERR_UNSUPPORTED."""
    rng = np.random.default_rng(0)
    g = random_lq_game(rng, 33, [2, 2], 5, 1)
    d = abi.make_dims(33, [2, 2], 5, 1, abi.F64)
    with pytest.raises(hip.IlqgError) as e:
        hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    assert e.value.status == abi.ERR_UNSUPPORTED
    spec = examples.three_player_overtaking()  # three Car6D ...
    for _ in range(3):                         # ... and three more: n = 36
        spec.add_player(abi.DYN_CAR_6D, 4.0)
    for i in range(3, 6):
        spec.quadratic(i, 1.0, 0, 0.0, control_of=i)
    with pytest.raises(hip.IlqgError) as e:
        hip.Problem(spec, abi.F64)
    assert e.value.status == abi.ERR_UNSUPPORTED


MIXED = ["mixed_dubins_car_scene", "mixed_dubins_car_scene_open_loop", "three_unicycle_scene"]


@pytest.mark.parametrize("scene", MIXED + ["mixed_dubins_car_scene_constrained"])
@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_stage_kernels_of_mixed_dimension_games_match_oracle(hip, oracle, scene, dtype):
    """ilqg_rollout_batch / linearize / quadraticize / total_costs on games whose players have different control
    dimensions (Dubins m = 1 beside Car5D m = 2) or an un-instantiated shape."""
    spec = examples.CONFIGS[scene]()
    B = 5
    rng = np.random.default_rng(5)
    x0 = examples.jittered_x0(spec, B, seed=2)
    prob, O = hip.Problem(spec, dtype), oracle.OracleProblem(spec)
    n, m, N, T = spec.n, spec.m, len(spec.subsystems), spec.T
    xs_ref, us_ref = np.zeros((B, T, n)), 0.05 * rng.standard_normal((B, T, m))
    P, alpha = 0.02 * rng.standard_normal((B, T, m * n)), 0.1 * rng.standard_normal((B, T, m))
    tol = 1e-9 if dtype == abi.F64 else 2e-4
    xs, us = prob.rollout(x0, xs_ref, us_ref, P, alpha)
    xr, ur = O.rollout(dtype, x0, xs_ref, us_ref, P, alpha)
    assert rel_err(_np(xs), xr) < tol and rel_err(_np(us), ur) < tol
    A, Bm = prob.linearize(xr, ur)
    Ar, Br = O.linearize(dtype, xr, ur)
    assert rel_err(_np(A), Ar) < tol and rel_err(_np(Bm), Br) < tol
    nc = spec.num_constraints
    lam = np.abs(rng.standard_normal((B, nc, T))) if nc else None
    mu = np.full(B, 10.0) if nc else None
    Q, l, R, r = prob.quadraticize(xr, ur, lam, mu)
    Qr, lr, Rr, rr = O.quadraticize(dtype, xr, ur, lam, mu)
    for got, want in ((Q, Qr), (l, lr), (R, Rr), (r, rr)):
        assert rel_err(_np(got), want) < (tol if dtype == abi.F64 else 2e-3)
    costs, _ = prob.total_costs(xr, ur)
    cr, _ = O.total_costs(dtype, xr, ur)
    assert rel_err(_np(costs), cr) < (tol if dtype == abi.F64 else 1e-3)


def _forced(oracle, spec, B, K, seed):
    rng = np.random.default_rng(seed)
    x0 = examples.jittered_x0(spec, B, seed=11)
    op = oracle.OracleProblem(spec)
    free = op.solve(abi.F64, x0, merit_log_len=K)
    a0 = float(spec.params.initial_alpha_scaling)
    acc = free["log"][:, :K, 2].astype(np.float64)
    acc = np.where(np.isfinite(acc) & (acc > 1e-6 * a0), acc, a0 / 256.0)
    steps = acc * 0.5 ** rng.choice([0, 1, 2, 5], p=[0.4, 0.3, 0.2, 0.1], size=acc.shape)
    return x0, op, steps, x0 + 1e-12 * rng.standard_normal(x0.shape)


def _compare_forced(hip, op, spec, dtype, x0, steps, x0_nudged, K, solve_kwargs, min_cover=0.7):
    """Every instance after every one of K forced-step iterations (the scheme of tests/test_gpu_forced.py: pairs whose
    conditioning amplifies a 1e-12 nudge of x0 past 1e-8 in the oracle itself are skipped and counted)."""
    B = x0.shape[0]
    prob = hip.Problem(spec, dtype)
    f64 = dtype == abi.F64
    tol_op, tol_st = (1e-9, 1e-9) if f64 else (2e-3, 1e-2)
    compared = 0
    for k in range(1, K + 1):
        ref = op.solve(dtype, x0, fixed_iters=k, forced_steps=steps[:, :k], merit_log_len=k)
        r64 = ref if f64 else op.solve(abi.F64, x0, fixed_iters=k, forced_steps=steps[:, :k])
        r64n = op.solve(abi.F64, x0_nudged, fixed_iters=k, forced_steps=steps[:, :k])
        out = prob.solve(x0, fixed_iters=k, forced_steps=steps[:, :k], **solve_kwargs)
        st = prob.solve_state(out)
        assert np.array_equal(_np(out["iters"]), ref["iters"]) and np.all(_np(out["status"]) == 1)
        dev = {q: _np(out[q]) for q in ("xs", "us", "P", "alpha", "costs")}
        for b in range(B):
            amp = max(rel_err(r64n[q][b], r64[q][b]) for q in ("xs", "us", "rawP", "alpha"))
            if amp > 1e-8 or not np.isfinite(ref["log"][b, k - 1, 0]):
                continue
            if not f64 and max(rel_err(ref[q][b], r64[q][b]) for q in ("rawP", "alpha")) > 2e-3:
                continue
            compared += 1
            where = "k=%d instance %d" % (k, b)
            assert rel_err(dev["xs"][b], ref["xs"][b]) < tol_op, where
            us_scale = float(np.max(np.abs(ref["us"][b]))) if f64 else max(1.0, float(np.max(np.abs(ref["us"][b]))))
            assert float(np.max(np.abs(dev["us"][b] - ref["us"][b]))) < tol_op * max(us_scale, 1e-30), where
            assert rel_err(dev["P"][b], ref["rawP"][b]) < tol_st, where
            assert rel_err(dev["alpha"][b], ref["alpha"][b]) < tol_st, where
            scale = max(1.0, float(np.max(np.abs(ref["xs"][b]))), float(np.max(np.abs(ref["costs"][b]))))
            assert float(np.max(np.abs(dev["costs"][b] - ref["costs"][b]))) < tol_op * scale, where
            merit_ref, ed_ref = ref["log"][b, k - 1, 0], ref["log"][b, k - 1, 1]
            assert abs(_np(st["last_merit"])[b] - merit_ref) <= tol_op * max(1.0, abs(merit_ref)), where
            assert abs(_np(st["expected_decrease"])[b] - ed_ref) <= tol_st * max(1.0, abs(ed_ref)), where
    assert compared >= min_cover * B * K, "only %d of %d (instance, iteration) pairs were well-conditioned" % (compared, B * K)


@pytest.mark.parametrize("scene", MIXED + ["three_unicycle_scene_open_loop"])
@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
@pytest.mark.parametrize("padded", [True, False], ids=["padded_sweep", "all_lds_sweep"])
def test_forced_step_solves_of_mixed_dimension_games_match_oracle_after_every_iteration(hip, oracle, scene, dtype, padded):
    """Both sweeps of the run-time-dimensioned solve: the specialised sweep of the shape the game embeds in
    (ilqg_solve_options::padded_sweep, the default for these shapes: (8, 2, (1, 2)) in (8, 2, 2), (12, 3, 2) in (14, 3, 2))
    and the all-LDS sweeps with run-time dimensions (csrc/ilqg_lq_generic.hpp)."""
    spec = examples.CONFIGS[scene]()
    K, B = 5, 8
    x0, op, steps, x0n = _forced(oracle, spec, B, K, seed=31)
    _compare_forced(hip, op, spec, dtype, x0, steps, x0n, K, dict(padded_sweep=padded))


def test_padded_sweep_is_reported_requested_and_refused(hip, oracle):
    """ilqg_problem_last_schedule says which sweep a run-time-dimensioned solve ran; ON sends a problem that was pushed onto
    the run-time-dimensioned kernels (generic_kernels) through the padded kernel of its own shape as well — no padding,
    the copies only — and is refused with ILQG_ERR_UNSUPPORTED where no instantiated shape holds the game (five players).
    Free-running solves under both sweeps take the same decisions on instances the oracle's own decisions are stable on."""
    import torch
    spec = examples.mixed_dubins_car_scene()
    x0 = examples.jittered_x0(spec, 12, seed=4)
    prob = hip.Problem(spec, abi.F64)
    a = prob.solve(x0)
    torch.cuda.synchronize()
    assert prob.last_schedule() & abi.SCHEDULE_GENERIC and prob.last_schedule() & abi.SCHEDULE_PADDED_SWEEP
    b = prob.solve(x0, padded_sweep=False)
    torch.cuda.synchronize()
    assert prob.last_schedule() & abi.SCHEDULE_GENERIC and not prob.last_schedule() & abi.SCHEDULE_PADDED_SWEEP
    O = oracle.OracleProblem(spec)
    ref, refn = O.solve(abi.F64, x0), O.solve(abi.F64, x0 + 1e-12 * np.random.default_rng(0).standard_normal(x0.shape))
    robust = [i for i in range(12) if ref["iters"][i] == refn["iters"][i] and ref["status"][i] == refn["status"][i] and
              rel_err(ref["xs"][i], refn["xs"][i]) < 1e-7]
    assert len(robust) >= 6
    for i in robust:
        assert _np(a["iters"])[i] == _np(b["iters"])[i] == ref["iters"][i], i
        assert rel_err(_np(a["xs"])[i], _np(b["xs"])[i]) < 1e-6 and rel_err(_np(a["xs"])[i], ref["xs"][i]) < 1e-6, i
    # an instantiated shape on the run-time-dimensioned kernels: AUTO keeps their own sweep, ON takes the padded kernel
    spec14 = examples.modified_three_player_intersection()
    spec14.params.expected_decrease_fraction, spec14.params.initial_alpha_scaling = 0.001, 0.1
    x14 = examples.jittered_x0(spec14, 6, seed=2)
    p14 = hip.Problem(spec14, abi.F64)
    g = p14.solve(x14, fixed_iters=3, generic_kernels=True)
    torch.cuda.synchronize()
    assert not p14.last_schedule() & abi.SCHEDULE_PADDED_SWEEP
    h = p14.solve(x14, fixed_iters=3, generic_kernels=True, padded_sweep=True)
    torch.cuda.synchronize()
    assert p14.last_schedule() & abi.SCHEDULE_PADDED_SWEEP
    r14 = oracle.OracleProblem(spec14).solve(abi.F64, x14, fixed_iters=3)
    for out in (g, h):
        assert rel_err(_np(out["xs"]), r14["xs"]) < 1e-7 and rel_err(_np(out["P"]), r14["rawP"]) < 1e-6
    # five players: no instantiated shape has five (ILQG_FOR_DIMS, csrc/ilqg_api.hip)
    s5 = abi.ProblemSpec(T=10)
    for _ in range(5):
        s5.add_player(abi.DYN_UNICYCLE_4D, 0.0)
    for i in range(5):
        s5.quadratic(i, 1.0, 4 * i + 3, 1.0)
        s5.quadratic(i, 1.0, -1, 0.0, control_of=i)
    s5.x0 = np.zeros(s5.n)
    p5 = hip.Problem(s5, abi.F64)
    x5 = 0.1 * np.random.default_rng(3).standard_normal((3, s5.n))
    out5 = p5.solve(x5, fixed_iters=2)   # AUTO: the all-LDS sweep
    torch.cuda.synchronize()
    assert not p5.last_schedule() & abi.SCHEDULE_PADDED_SWEEP
    r5 = oracle.OracleProblem(s5).solve(abi.F64, x5, fixed_iters=2)
    assert rel_err(_np(out5["xs"]), r5["xs"]) < 1e-9
    with pytest.raises(hip.IlqgError) as e:
        p5.solve(x5, fixed_iters=2, padded_sweep=True)
    assert e.value.status == abi.ERR_UNSUPPORTED


@pytest.mark.parametrize("scene", ["modified_three_player_intersection", "roundabout_merging"])
def test_generic_kernels_forced_onto_an_instantiated_shape_match_oracle(hip, oracle, scene):
    """ilqg_solve_options::generic_kernels = ON: BASELINE config 2's scene (feedback sweep) and config 4's (open-loop
    sweep) through the run-time-dimensioned kernels, against the oracle after every forced-step iteration."""
    spec = examples.CONFIGS[scene]()
    K, B = 4, 6
    x0, op, steps, x0n = _forced(oracle, spec, B, K, seed=37)
    _compare_forced(hip, op, spec, abi.F64, x0, steps, x0n, K, dict(generic_kernels=True), min_cover=0.6)


@pytest.mark.parametrize("scene", ["mixed_dubins_car_scene", "three_unicycle_scene_open_loop"])
def test_free_running_solves_of_mixed_dimension_games_match_oracle(hip, oracle, scene):
    """ILQSolver::Solve with its own line search.  Instances whose decisions the oracle itself does not reproduce from
    x0 + 1e-12 are left out (measured, as in test_gpu_parity.py); the rest must agree in iteration count, success,
    convergence and final iterate."""
    spec = examples.CONFIGS[scene]()
    B = 10
    x0 = examples.jittered_x0(spec, B, seed=5)
    rng = np.random.default_rng(1)
    O = oracle.OracleProblem(spec)
    ref = O.solve(abi.F64, x0)
    # two nudges: the size of fp64 round-off in x0, and the size of the device-vs-oracle tolerance itself
    nudged = [O.solve(abi.F64, x0 + e * rng.standard_normal(x0.shape)) for e in (1e-12, 1e-9)]
    out = hip.Problem(spec, abi.F64).solve(x0, log_capacity=int(spec.params.max_solver_iters) + 2)
    robust = [b for b in range(B) if all(ref["iters"][b] == r["iters"][b] and ref["status"][b] == r["status"][b] and
                                         ref["converged"][b] == r["converged"][b] and
                                         rel_err(ref["xs"][b], r["xs"][b]) < 1e-6 for r in nudged)]
    assert len(robust) >= B // 2, "scene too ill-conditioned to compare free-running solves: %s" % robust
    for b in robust:
        assert _np(out["iters"])[b] == ref["iters"][b] and _np(out["status"])[b] == ref["status"][b], b
        assert _np(out["converged"])[b] == ref["converged"][b], b
        assert rel_err(_np(out["xs"])[b], ref["xs"][b]) < 1e-6, b
        # alpha carries the last accepted step; at a converged iterate it is rounding noise (1e-13) whose step the two
        # runs need not agree on: compared on the scale of the controls
        assert np.max(np.abs(_np(out["alpha"])[b] - ref["alpha"][b])) < 1e-6 * max(1.0, np.max(np.abs(ref["us"][b]))), b
        np.testing.assert_allclose(_np(out["costs"])[b], ref["costs"][b], rtol=1e-6)
        # the iterate log of the run-time-dimensioned path: one entry per logged iterate, the last one = the result
        cnt = int(_np(out["log"]["count"])[b])
        assert cnt == ref["iters"][b] + 1 if ref["status"][b] else cnt >= 1
        assert np.array_equal(_np(out["log"]["xs"])[b, cnt - 1], _np(out["xs"])[b])


def test_augmented_lagrangian_solve_of_a_mixed_dimension_game_matches_oracle(hip, oracle):
    """AugmentedLagrangianSolver::Solve on the constrained Dubins + Car5D game (a control constraint on the player with
    ONE control, a state constraint, a proximity constraint)."""
    spec = examples.mixed_dubins_car_scene(constrained=True)
    spec.params.max_solver_iters = 25
    B = 8
    x0 = examples.jittered_x0(spec, B, seed=9)
    rng = np.random.default_rng(2)
    O = oracle.OracleProblem(spec)
    ref = O.solve(abi.F64, x0, augmented_lagrangian=True)
    refn = O.solve(abi.F64, x0 + 1e-12 * rng.standard_normal(x0.shape), augmented_lagrangian=True)
    out = hip.Problem(spec, abi.F64).solve(x0, augmented_lagrangian=True)
    robust = [b for b in range(B) if ref["iters"][b] == refn["iters"][b] and ref["status"][b] == refn["status"][b] and
              rel_err(ref["xs"][b], refn["xs"][b]) < 1e-7]
    assert len(robust) >= B // 2, robust
    for b in robust:
        assert _np(out["iters"])[b] == ref["iters"][b] and _np(out["status"])[b] == ref["status"][b], b
        assert rel_err(_np(out["xs"])[b], ref["xs"][b]) < 1e-6, b
        np.testing.assert_allclose(_np(out["costs"])[b], ref["costs"][b], rtol=1e-6)


@pytest.mark.parametrize("cfg", ["mixed_dubins_car_scene", "three_unicycle_scene", "mixed_dubins_car_scene_open_loop"])
def test_speculative_line_search_on_the_run_time_dimensioned_path_changes_nothing(hip, cfg):
    """The run-time-dimensioned solve lists its back-tracking instances and probes their next step sizes side by side
    (round 5: a failing 100-step line search of one instance used to cost the whole batch 100 serial passes).  Same
    arithmetic, same decisions: free-running and fixed-iteration solves come back bit for bit as without probing."""
    import torch
    spec = examples.CONFIGS[cfg]()
    x0 = examples.jittered_x0(spec, 96, seed=0)
    for kw in (dict(), dict(fixed_iters=5)):
        a = hip.Problem(spec, abi.F64).solve(x0, probe=False, **kw)
        b = hip.Problem(spec, abi.F64).solve(x0, probe=True, **kw)
        torch.cuda.synchronize()
        for k in ("iters", "status", "converged", "xs", "us", "P", "alpha", "costs"):
            assert torch.equal(a[k], b[k]), (cfg, kw, k)
    prob = hip.Problem(spec, abi.F64)
    prob.solve(x0[:4], fixed_iters=1)
    torch.cuda.synchronize()
    assert prob.last_schedule() & abi.SCHEDULE_GENERIC  # the scene really runs on the run-time-dimensioned kernels
