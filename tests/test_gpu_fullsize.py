"""Parity at BASELINE.json's full sizes, through properties that do not need the oracle to run 1024 games:
 * a batch is a set of independent games — what instance b gets must not depend on the rest of the batch
   (sub-batch and permutation invariance, bit for bit);
 * the LQ Nash sweep is affine in (l, r, x0): gains P ignore them, alpha and delta_x scale with them;
 * a random sample of the batch is compared with the oracle.
Config 2: three-player intersection, n=14, T=100, fp64, 1024 instances; config 3's per-GPU share: fp32, 8192."""
import numpy as np
import pytest

from ilqgames_amd import abi, examples
from helpers import dims_of, random_lq_game, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from ilqgames_amd import hip as h
    return h


def _np(t):
    return t.detach().cpu().numpy()


def _spec():
    spec = examples.modified_three_player_intersection()
    spec.params.initial_alpha_scaling = 0.1   # bench.py's workload
    spec.params.expected_decrease_fraction = 0.001
    return spec


KEYS = ("xs", "us", "P", "alpha", "costs", "iters", "status")


@pytest.mark.parametrize("dtype,B", [(abi.F64, 1024), (abi.F32, 8192)])
def test_full_batch_instances_are_independent(hip, dtype, B):
    spec = _spec()
    K = 4
    x0 = examples.jittered_x0(spec, B, seed=77)
    prob = hip.Problem(spec, dtype)
    # Which of its two one-tile feedback sweeps the library runs is a function of the batch size (the single-wave form
    # from five instances per CU on, ilqg_solve_options::single_wave_sweep): two schedules of one recursion that agree to
    # rounding, not bit for bit — so the slice is solved with the schedule the full batch ran.  Everything else the
    # library decides from the batch size (split passes, hand-off, probing, priorities) is bit-identical by construction.
    single = B >= 5 * hip.device_info()[1]
    full = {k: _np(v).copy() for k, v in prob.solve(x0, fixed_iters=K).items() if k in KEYS}
    assert np.isfinite(full["xs"]).all() and np.isfinite(full["P"]).all()
    assert np.all(full["iters"] == K) and set(full["status"].tolist()) <= {0, 1}
    # a slice of the batch solved on its own
    lo, hi = B // 3, B // 3 + 37
    part = prob.solve(x0[lo:hi], fixed_iters=K, single_wave_sweep=single)
    for k in KEYS:
        assert np.array_equal(_np(part[k]), full[k][lo:hi]), k
    # the whole batch in another order
    perm = np.random.default_rng(5).permutation(B)
    shuf = prob.solve(x0[perm], fixed_iters=K)
    for k in KEYS:
        assert np.array_equal(_np(shuf[k]), full[k][perm]), k


def test_full_batch_sample_matches_oracle_fp64(hip, oracle):
    spec = _spec()
    B, K = 1024, 4
    x0 = examples.jittered_x0(spec, B, seed=77)
    out = hip.Problem(spec, abi.F64).solve(x0, fixed_iters=K)
    pick = np.random.default_rng(9).choice(B, 24, replace=False)
    ref = oracle.OracleProblem(spec).solve(abi.F64, x0[pick], fixed_iters=K, merit_log_len=K, threads=8)
    bt = np.nan_to_num(ref["log"][:, :, 3], nan=0.0).max(axis=1)
    ok = np.where((bt <= 12) & (ref["status"] == 1))[0]  # well-conditioned line searches (test_gpu_parity._clean)
    assert len(ok) >= 12
    sel = pick[ok]
    assert np.array_equal(_np(out["iters"])[sel], ref["iters"][ok])
    assert rel_err(_np(out["xs"])[sel], ref["xs"][ok]) < 1e-7
    assert rel_err(_np(out["P"])[sel], ref["P"][ok]) < 1e-6
    assert rel_err(_np(out["alpha"])[sel], ref["alpha"][ok]) < 1e-6
    assert rel_err(_np(out["costs"])[sel], ref["costs"][ok]) < 1e-8


@pytest.mark.parametrize("open_loop", [False, True])
def test_full_size_lq_sweep_is_affine_in_its_linear_terms_fp64(hip, open_loop):
    """1024 random three-player games of the headline shape (n=14, m=2+2+2, T=100): scaling (l, r, x0) by c scales
    alpha and delta_x by c and leaves P alone (to rounding: 1e-11); zero linear terms and x0 give zero alpha, dx."""
    rng = np.random.default_rng(123)
    n, ms, T, B = 14, [2, 2, 2], 100, 1024
    g = random_lq_game(rng, n, ms, T, B)
    d = dims_of(g, abi.F64, adaptive=not open_loop)
    x0 = rng.standard_normal((B, n))

    def run(c):
        return [(_np(v) if v is not None else None) for v in
                hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], c * g["l"], g["R"], c * g["r"], g["pairs"], x0=c * x0,
                                open_loop=open_loop)]
    P1, a1, dx1 = run(1.0)
    P3, a3, dx3 = run(-2.5)
    P0, a0, dx0 = run(0.0)
    assert np.isfinite(a1).all() and np.isfinite(dx1).all()
    if not open_loop:
        assert rel_err(P3, P1) < 1e-13 and rel_err(P0, P1) < 1e-13 and np.abs(P1).max() > 1e-3
    assert rel_err(a3, -2.5 * a1) < 1e-11 and rel_err(dx3, -2.5 * dx1) < 1e-11
    assert not a0.any() and not dx0.any()


def test_full_batch_stage_outputs_have_the_reference_structure_fp64(hip):
    """Linearisation and quadraticisation of 1024 solved trajectories: A is block diagonal over the players'
    subsystems with B_i confined to player i's rows (ConcatenatedDynamicalSystem::Linearize,
    src/concatenated_dynamical_system.cpp:86-107), every Q_i and R_ij is exactly symmetric (each cost adds H(x,y) and
    H(y,x) from one value) with the regularisation on its diagonal, and the control blocks are those of the pair table."""
    import torch
    spec = _spec()
    B = 1024
    x0 = examples.jittered_x0(spec, B, seed=77)
    prob = hip.Problem(spec, abi.F64)
    out = prob.solve(x0, fixed_iters=2)
    A, Bm = prob.linearize(out["xs"], out["us"])
    n, m, N, T = prob.n, prob.m, prob.N, prob.T
    A = A.reshape(B, T, n, n).transpose(2, 3)   # column-major blocks -> [row, col]
    Bm = Bm.reshape(B, T, m, n).transpose(2, 3)
    xoff = [spec.xoff(i) for i in range(N)] + [n]
    uoff = np.concatenate([[0], np.cumsum(spec.udims)])
    for i in range(N):
        for j in range(N):
            if i != j:
                assert not A[:, :, xoff[i]:xoff[i + 1], xoff[j]:xoff[j + 1]].any()
                assert not Bm[:, :, xoff[i]:xoff[i + 1], uoff[j]:uoff[j + 1]].any()
    assert torch.all(torch.diagonal(A, dim1=2, dim2=3) == 1.0)  # (I + dt J): none of these models has J_kk != 0
    del A, Bm
    Q, l, R, r = prob.quadraticize(out["xs"], out["us"])
    Q = Q.reshape(B, T, N, n, n)
    assert torch.equal(Q, Q.transpose(3, 4))
    assert torch.all(torch.diagonal(Q, dim1=3, dim2=4) >= 10.0)  # sigma_x = 10 plus convex terms' diagonals
    off = 0
    for (i, j) in prob.pairs:
        mj = spec.udims[j]
        blk = R[:, :, off:off + mj * mj].reshape(B, T, mj, mj)
        assert torch.equal(blk, blk.transpose(2, 3))
        assert torch.all(torch.diagonal(blk, dim1=2, dim2=3) >= 10.0)
        off += mj * mj
    assert off == prob.Rsz and torch.isfinite(l).all() and torch.isfinite(r).all()


def test_full_batch_receding_horizon_step_is_per_instance_fp64(hip):
    """Config 5's batch (2048 plans): one receding-horizon step — integrate the state along the plan, re-anchor,
    warm-started solve, splice — gives every instance what it gets in a 29-instance slice of the batch, bit for bit,
    and the row bookkeeping obeys the reference's invariants for every instance."""
    import torch
    spec = examples.three_player_collision_avoidance_reachability()
    spec.params.max_solver_iters = 3
    B = 2048
    x0 = examples.jittered_x0(spec, B, seed=31)
    prob = hip.Problem(spec, abi.F64)
    # the sweep's schedule is a function of the batch size and the two schedules agree to rounding only: pinned to what
    # the full batch runs by itself, for the slice too (test_full_batch_instances_are_independent)
    prob.single_wave_sweep = B >= 5 * hip.device_info()[1]

    def step(x0_part):
        b = x0_part.shape[0]
        bufs = prob.alloc_solve_buffers(b)
        x = torch.as_tensor(x0_part, dtype=torch.float64, device="cuda").clone()
        prob.solve(x, bufs)
        plan = prob.new_plan(b)
        t0 = torch.zeros(b, dtype=torch.float64, device="cuda")
        prob.solution_splice(plan, bufs, t0)
        active = torch.ones(b, dtype=torch.int32, device="cuda")
        prob.plan_integrate(plan, 0.0, 0.25, 0.6, x, active)
        x0n, st0, first = prob.receding_horizon_sync(plan, x, 0.25, 0.25, bufs, active)
        prob.solve_again(x0n, bufs, active=active)
        conv = torch.ones(b, dtype=torch.int32, device="cuda")
        prob.solution_splice(plan, bufs, st0, converged=conv, active=active)
        return dict(x=_np(x), x0n=_np(x0n), st0=_np(st0), first=_np(first), active=_np(active), xs=_np(bufs["xs"]),
                    P=_np(bufs["P"]), len=_np(plan["len"]), t0=_np(plan["t0"]), pxs=_np(plan["xs"]))
    full = step(x0)
    lo, hi = 700, 729
    part = step(x0[lo:hi])
    for k in full:
        assert np.array_equal(part[k], full[k][lo:hi]), k
    assert full["active"].all()
    # SetUpNextRecedingHorizon's invariant (src/problem.cpp:123) and SolutionSplicer's row count (:100-103)
    assert np.all(np.abs(0.25 + 0.25 - full["st0"]) <= spec.dt + 1e-9)
    kept = np.minimum(np.floor(1e-4 + full["st0"] / spec.dt).astype(int), 5)
    assert np.array_equal(full["len"], spec.T + kept)
    assert np.all((full["first"] >= 0) & (full["first"] < spec.T))


def test_config5_as_written_full_size_fp64(hip):
    """BASELINE config 5 as written: ThreePlayerCollisionAvoidanceReachabilityExample, 2048 jittered instances,
    AugmentedLagrangianSolver::Solve at every replanning instant, up to 200 solver calls (RecedingHorizonSimulator
    with a fixed simulated solve time of 0.25 s, src/receding_horizon_simulator.cpp:64-137).  At this size the oracle
    cannot follow (tests/test_gpu_receding.py compares 64 instances x 22 calls with it); here the size-independent
    properties: an instance's run does not depend on the batch it is in — a slice holding the longest-running
    instances reproduces, bit for bit, what they had after 40 calls inside the full batch —, nobody re-enters the
    loop, everything stays finite, plan bookkeeping obeys the reference's invariants, and an instance is out after its
    first call exactly when that solve reported failure (the reference's CHECK(success), :77).  (Measured: about 3 %
    of the jittered instances pass their first solve — so does the oracle on a 64-instance sample, and the example's
    own initial state is not among them under these parameters; the last survivor leaves the loop at call 149.)"""
    import torch
    spec = examples.three_player_collision_avoidance_reachability()
    B, calls, probe_at = 2048, 200, 39
    x0 = examples.jittered_x0(spec, B, seed=5)
    prob = hip.Problem(spec, abi.F64)
    prob.single_wave_sweep = B >= 5 * hip.device_info()[1]  # one schedule of the sweep for the batch and its slice
    trace, snap = [], {}

    def on_record(r, info):
        trace.append((int(info["active"].sum().item()), _np(info["bufs"]["status"]).copy() if r == 0 else None))
        if r == probe_at:
            snap.update(xs=info["bufs"]["xs"].clone(), P=info["bufs"]["P"].clone(), x=info["x_measured"].clone(),
                        active=info["active"].clone())

    out = prob.receding_horizon_simulate(x0, 1e9, 0.25, augmented_lagrangian=True, max_records=calls, on_record=on_record)
    assert out["calls"] >= 100, "the loop should run for at least a hundred replanning instants"
    nrec = _np(out["num_records"])
    act = np.array([a for a, _ in trace])
    assert np.all(np.diff(act) <= 0), "an instance re-entered the loop"
    assert np.array_equal(nrec > 1, trace[0][1] == 1), "leaving after call 1 <=> the first solve failed"
    assert nrec.max() >= 100 and (nrec > probe_at).sum() >= 4, "some instances keep replanning"
    assert torch.isfinite(out["x"]).all() and torch.isfinite(out["plan"]["xs"]).all()
    plen = _np(out["plan"]["len"])
    assert np.all((plen >= spec.T) & (plen <= spec.T + 5))
    # batch independence: the longest runners (and some instances that drop out at once) on their own, 40 calls
    pick = np.unique(np.concatenate([np.argsort(-nrec)[:12], np.arange(40, 60)]))
    snap2 = {}

    def on_record2(r, info):
        if r == probe_at:
            snap2.update(xs=info["bufs"]["xs"].clone(), P=info["bufs"]["P"].clone(), x=info["x_measured"].clone(),
                         active=info["active"].clone())

    prob.receding_horizon_simulate(x0[pick], 1e9, 0.25, augmented_lagrangian=True, max_records=probe_at + 1,
                                   on_record=on_record2)
    assert torch.equal(snap2["active"], snap["active"][pick]) and snap2["active"].sum() >= 1
    live = snap2["active"].bool()
    for k in ("xs", "P", "x"):
        assert torch.equal(snap2[k][live], snap[k][pick][live]), k


def test_config4_full_batch_slice_is_reproduced_fp64(hip):
    """Config 4 at full size (roundabout merging, n=24, N=4, T=150, open-loop sweep, 4096 instances): a 21-instance
    slice solved on its own reproduces its share of the full batch bit for bit; everything finite."""
    import torch
    spec = examples.CONFIGS["roundabout_merging_T150"]()
    assert spec.params.open_loop and spec.T == 150 and spec.n == 24
    spec.params.initial_alpha_scaling = 0.1
    spec.params.expected_decrease_fraction = 0.001
    B, K = 4096, 2
    x0 = examples.jittered_x0(spec, B, seed=13)
    prob = hip.Problem(spec, abi.F64)
    full = prob.solve(x0, fixed_iters=K)
    lo, hi = 2000, 2021
    keep = {k: full[k][lo:hi].clone() for k in KEYS}
    assert torch.isfinite(full["xs"]).all() and torch.isfinite(full["alpha"]).all()
    assert torch.all(full["iters"] == K) and not full["P"].any()  # open-loop strategies carry no gains
    part = prob.solve(x0[lo:hi], fixed_iters=K)
    for k in KEYS:
        assert torch.equal(part[k], keep[k]), k
