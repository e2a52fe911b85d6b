"""GPU parity: HIP path (through the C ABI of libilqg_hip.so) vs the CPU oracle on the same
seeded inputs, and vs the committed golden vectors.  Tolerances are stated per test:
fp64 device vs fp64 oracle ~1e-9 relative (same algorithm, different summation order / FMA
contraction); fp32 device vs fp32 oracle ~1e-3 relative on P (conditioning of S after the
Gershgorin step x fp32 round-off)."""
import numpy as np
import pytest

from ilqgames_amd import abi, examples
from helpers import oracle_with_stability, dims_of, load_golden_lq, random_lq_game, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from ilqgames_amd import hip as h
    name, cus = h.device_info()
    assert "gfx950" in name, name
    return h


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_mfma_tile_layout(hip, dtype):
    """The accumulator-layout identities of ilqg_mfma.hpp, with ASYMMETRIC operands (a transposed
    read or write would pass a symmetric test)."""
    rng = np.random.default_rng(3)
    X, Y, Cm = (rng.standard_normal((16, 16)) for _ in range(3))
    out = hip.selftest_mfma(dtype, X, Y, Cm)
    ref = X.T @ Y + Cm
    assert rel_err(out, ref) < (1e-13 if dtype == abi.F64 else 1e-5)
    out_id = hip.selftest_mfma(dtype, np.eye(16), Y, 0 * Cm)
    assert np.array_equal(out_id.astype(np.float32 if dtype == abi.F32 else np.float64),
                          Y.astype(np.float32 if dtype == abi.F32 else np.float64))


@pytest.mark.parametrize("name", ["lq_feedback_random.npz", "lq_feedback_unicycle.npz", "lq_feedback_pointmass.npz"])
def test_lq_feedback_matches_reference_python_golden(hip, name):
    """Device sweep vs the reference's own numpy solver (fixtures of tests/golden/make_golden.py) — every fixture:
    lq_feedback_random.npz has n = 5 and control dimensions (2, 1, 2), a shape the run-time-dimensioned kernels take
    (ilqg_lq_generic.hpp); the other two run on their specialised instantiations."""
    g = load_golden_lq(name)
    d = dims_of(g, abi.F64, adaptive=False)
    P, alpha, _ = hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    assert rel_err(_np(P), g["P_ref"]) < 1e-9
    assert rel_err(_np(alpha), g["alpha_ref"]) < 1e-9
    assert np.all(_np(P)[:, -1] == 0) and np.all(_np(alpha)[:, -1] == 0)


@pytest.mark.parametrize("dims", [(14, 3, 2), (16, 3, 2), (15, 3, 2), (24, 4, 2), (18, 3, 2), (10, 2, 2), (6, 3, 2),
                                  (4, 2, 2), (6, 2, 1), (3, 2, 1), (3, 1, 1), (2, 2, 1), (17, 3, 2), (8, 2, 1)])
@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_lq_feedback_matches_oracle_random(hip, oracle, dims, dtype):
    n, N, mu = dims
    rng = np.random.default_rng(100 * n + N)
    T, B = 25, 5
    g = random_lq_game(rng, n, [mu] * N, T, B)
    d = dims_of(g, dtype, adaptive=True)
    x0 = rng.standard_normal((B, n))
    Pr, ar, dxr, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0)
    P, alpha, dx = hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0)
    tol = 1e-9 if dtype == abi.F64 else 2e-3
    assert rel_err(_np(P), Pr) < tol
    assert rel_err(_np(alpha), ar) < tol
    assert rel_err(_np(dx), dxr) < tol


def test_stand_alone_entry_points_run_in_caller_owned_scratch(hip, oracle):
    """ilqg_set_scratch: with a caller's buffer installed the stand-alone sweep (delta_x and costates asked for, which
    is what needs scratch) computes what it computes in the library's own allocation; a buffer that is too small is
    refused with the size that is needed, never grown behind the caller's back; NULL returns to the default."""
    import torch
    n, N, mu = 14, 3, 2
    rng = np.random.default_rng(5)
    T, B = 25, 5
    g = random_lq_game(rng, n, [mu] * N, T, B)
    d = dims_of(g, abi.F64, adaptive=True)
    x0 = rng.standard_normal((B, n))
    args = (d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    ref = [_np(v).copy() for v in hip.lq_feedback(*args, x0=x0, want_costates=True)]
    big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    small = torch.empty(256, dtype=torch.uint8, device="cuda")
    try:
        hip.set_scratch(big)
        big.fill_(0xAB)
        out = [_np(v).copy() for v in hip.lq_feedback(*args, x0=x0, want_costates=True)]
        for a, b in zip(out, ref):
            assert np.array_equal(a, b)
        assert (big != 0xAB).any()  # the sweep really worked in the caller's buffer
        hip.set_scratch(small)
        with pytest.raises(hip.IlqgError) as e:
            hip.lq_feedback(*args, x0=x0, want_costates=True)
        assert e.value.status == abi.ERR_INVALID and any(ch.isdigit() for ch in str(e.value))
    finally:
        hip.set_scratch(None)
    out = [_np(v).copy() for v in hip.lq_feedback(*args, x0=x0, want_costates=True)]
    for a, b in zip(out, ref):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("dims", [(14, 3, 2), (16, 3, 2), (24, 4, 2), (18, 3, 2), (10, 2, 2), (6, 3, 2), (2, 2, 1),
                                  (17, 3, 2), (8, 2, 1)])
@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_lq_openloop_matches_oracle_random(hip, oracle, dims, dtype):
    """ilqg_lq_openloop_batch vs the oracle's LQOpenLoopSolver restatement (alpha, delta_xs; P == 0)."""
    n, N, mu = dims
    rng = np.random.default_rng(7 * n + N)
    T, B = 20, 4
    g = random_lq_game(rng, n, [mu] * N, T, B)
    g["A"] = np.asarray(g["A"])
    d = dims_of(g, dtype)
    x0 = rng.standard_normal((B, n))
    Pr, ar, dxr, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0,
                                     open_loop=True)
    P, alpha, dx = hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0,
                                   open_loop=True)
    tol = 1e-8 if dtype == abi.F64 else 5e-3
    assert np.all(_np(P) == 0)
    assert rel_err(_np(alpha), ar) < tol
    assert rel_err(_np(dx), dxr) < tol


@pytest.mark.parametrize("dims,T", [((24, 4, 2), 256), ((6, 3, 2), 256), ((2, 2, 1), 256), ((10, 2, 2), 256), ((14, 3, 2), 255)])
def test_lq_openloop_long_horizons_match_oracle_fp64(hip, oracle, dims, T):
    """The longest horizon the library takes (kMaxT = 256).  The open-loop sweep's forward pass has two forms
    (csrc/ilqg_lq_openloop.hpp): the state recursion on one wave with alpha and the expected decrease parallel over the
    steps, which keeps every x_k in LDS, and the step-by-step pass for shapes whose state history does not fit beside
    nothing else (small working sets, long horizons).  The random games are not stable over hundreds of steps, so every
    time step is compared at its own magnitude."""
    n, N, mu = dims
    rng = np.random.default_rng(13 * n + T)
    B = 2
    g = random_lq_game(rng, n, [mu] * N, T, B)
    g["A"] = np.asarray(g["A"])
    d = dims_of(g, abi.F64)
    x0 = rng.standard_normal((B, n))
    _, ar, dxr, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0, open_loop=True)
    P, alpha, dx = hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0, open_loop=True)
    assert np.all(_np(P) == 0)
    for got, ref in ((_np(alpha), ar), (_np(dx), dxr)):
        assert np.all(np.isfinite(got))
        scale = np.maximum(np.max(np.abs(ref), axis=-1, keepdims=True), 1e-300)
        assert np.max(np.abs(got - ref) / scale) < 1e-7


@pytest.mark.parametrize("open_loop", [False, True])
@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
@pytest.mark.parametrize("dims", [(4, 2, 2), (14, 3, 2), (24, 4, 2)])
def test_lq_costates_match_oracle(hip, oracle, dims, dtype, open_loop):
    """The costates output of both LQ entry points (lq_solver.h:63-69): -Z_i[k+1] dx_k - zeta_i[k+1] for the feedback
    solver (lq_feedback_solver.cpp:223-227), A_k^T (M_i[k+1] x_{k+1} + m_i[k+1]) for the open-loop one
    (lq_open_loop_solver.cpp:171-176); zero at the last step in both."""
    n, N, mu = dims
    rng = np.random.default_rng(11 * n + N + (1 if open_loop else 0))
    T, B = 16, 3
    g = random_lq_game(rng, n, [mu] * N, T, B)
    g["A"] = np.asarray(g["A"])
    d = dims_of(g, dtype)
    x0 = rng.standard_normal((B, n))
    _, ar, dxr, cor = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0,
                                      open_loop=open_loop, want_costates=True)
    _, alpha, dx, co = hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0,
                                       open_loop=open_loop, want_costates=True)
    tol = 1e-8 if dtype == abi.F64 else 5e-3
    assert rel_err(_np(alpha), ar) < tol and rel_err(_np(dx), dxr) < tol
    assert np.all(_np(co)[:, -1] == 0) and np.all(cor[:, -1] == 0)
    assert np.max(np.abs(cor)) > 0
    assert rel_err(_np(co), cor) < tol


def test_lq_costates_need_delta_xs(hip):
    """Both reference solvers CHECK that delta_xs and costates come together (lq_feedback_solver.cpp:77-78,
    lq_open_loop_solver.cpp:83-84); the C ABI returns ILQG_ERR_INVALID."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(3)
    g = random_lq_game(rng, 4, [2, 2], 6, 1)
    d = dims_of(g, abi.F64)
    dev = lambda v: torch.as_tensor(np.asarray(v), dtype=torch.float64, device="cuda").contiguous()  # noqa: E731
    arrs = [dev(g[k]) for k in ("A", "Bm", "Q", "l", "R", "r")]
    P = torch.empty((1, 6, 16), dtype=torch.float64, device="cuda")
    al = torch.empty((1, 6, 4), dtype=torch.float64, device="cuda")
    co = torch.empty((1, 6, 2, 4), dtype=torch.float64, device="cuda")
    for fn in (hip.lib().ilqg_lq_feedback_batch, hip.lib().ilqg_lq_openloop_batch):
        rc = fn(C.byref(d), *[C.c_void_p(a.data_ptr()) for a in arrs], abi.make_pairs(g["pairs"]), len(g["pairs"]),
                None, C.c_void_p(P.data_ptr()), C.c_void_p(al.data_ptr()), None, C.c_void_p(co.data_ptr()), None)
        assert rc == abi.ERR_INVALID


@pytest.mark.parametrize("T", [100, 150], ids=["T100_reference_horizon", "T150_as_BASELINE_writes_config_4"])
def test_ilq_solve_open_loop_matches_oracle_fp64(hip, oracle, T):
    """BASELINE config 4: roundabout merging (n=24, 4 players) with SolverParams::open_loop, fp64 — at the reference's
    compile-time horizon (T = 100) and at the T = 150 BASELINE.json asks for, whole solves against the oracle."""
    spec = examples.roundabout_merging(T=T, open_loop=True)
    spec.params.expected_decrease_fraction = 0.001
    B, K = 4, 3
    x0 = examples.jittered_x0(spec, B, seed=5)
    O = oracle.OracleProblem(spec)
    prob = hip.Problem(spec, abi.F64)
    if T == 150:
        # Fifty more steps of an open-loop (feedback-free) rollout make the game ill-conditioned: the ORACLE ITSELF turns
        # a 1e-12 nudge of x0 into 1e-5 .. 1e-8 after one iteration, 1e-3 after two and O(1) after three (measured here,
        # per instance).  Two correct implementations differ by rounding (1e-16, i.e. 1e-4 of that nudge) times the same
        # amplification: the device is held to 1e-3 of the oracle's own nudged difference (floor 1e-9), for one and two
        # iterations; the line-search decisions must agree.
        rng = np.random.default_rng(0)
        x0n = x0 + 1e-12 * rng.standard_normal(x0.shape)
        for k in (1, 2):
            ref = O.solve(abi.F64, x0, fixed_iters=k, merit_log_len=k)
            refn = O.solve(abi.F64, x0n, fixed_iters=k)
            out = prob.solve(x0, fixed_iters=k)
            st = prob.solve_state(out)
            assert np.array_equal(_np(out["iters"]), ref["iters"]) and np.all(_np(out["P"]) == 0)
            for b in range(B):
                amp = max(rel_err(refn[q][b], ref[q][b]) for q in ("xs", "alpha"))
                tol = max(1e-9, 1e-3 * amp)
                assert rel_err(_np(out["xs"])[b], ref["xs"][b]) < tol, (k, b, amp)
                assert rel_err(_np(out["alpha"])[b], ref["alpha"][b]) < 10 * tol, (k, b, amp)
                assert abs(_np(st["step"])[b] - ref["log"][b, k - 1, 2]) <= 1e-6 * ref["log"][b, k - 1, 2], (k, b)
        return
    ref = O.solve(abi.F64, x0, fixed_iters=K, merit_log_len=K)
    out = prob.solve(x0, fixed_iters=K)
    ok = _clean(ref)
    assert len(ok) >= 2
    assert np.array_equal(_np(out["iters"])[ok], ref["iters"][ok])
    assert rel_err(_np(out["xs"])[ok], ref["xs"][ok]) < 1e-6
    assert rel_err(_np(out["alpha"])[ok], ref["alpha"][ok]) < 1e-5
    assert np.all(_np(out["P"]) == 0)
    assert rel_err(_np(out["costs"])[ok], ref["costs"][ok]) < 1e-7


@pytest.mark.parametrize("T", [2, 3, 4, 7])
@pytest.mark.parametrize("cfg", ["roundabout_merging", "modified_three_player_intersection"])
def test_ilq_solve_open_loop_short_horizons_match_oracle_fp64(hip, oracle, cfg, T):
    """The open-loop sweep pipelines its compact rows two steps ahead (staging DMA behind barrier 2, the tiles filled by
    the waiting waves behind barrier 1): the horizons at which the prologue and the steady state meet, n = 24 (2 x 2
    tiles) and n = 14 (one tile), two iterations against the oracle."""
    spec = examples.roundabout_merging(T=T, open_loop=True) if cfg == "roundabout_merging" else examples.CONFIGS[cfg](T=T)
    spec.params.open_loop = 1
    spec.params.expected_decrease_fraction = 0.001
    B, K = 3, 2
    x0 = examples.jittered_x0(spec, B, seed=2)
    ref = oracle.OracleProblem(spec).solve(abi.F64, x0, fixed_iters=K)
    out = hip.Problem(spec, abi.F64).solve(x0, fixed_iters=K)
    assert np.array_equal(_np(out["iters"]), ref["iters"])
    assert np.all(_np(out["P"]) == 0)
    assert rel_err(_np(out["xs"]), ref["xs"]) < 1e-8
    assert rel_err(_np(out["alpha"]), ref["alpha"]) < 1e-7
    assert rel_err(_np(out["costs"]), ref["costs"]) < 1e-8


def test_lq_feedback_partial_pairs_and_no_regularization(hip, oracle):
    """Only the (i,i) blocks plus one off-diagonal block; adaptive_regularization off."""
    rng = np.random.default_rng(5)
    pairs = [(0, 0), (1, 1), (2, 2), (0, 2)]
    g = random_lq_game(rng, 14, [2, 2, 2], 30, 3, pairs=pairs)
    d = dims_of(g, abi.F64, adaptive=False)
    Pr, ar, _, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], pairs)
    P, alpha, _ = hip.lq_feedback(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], pairs, want_dx=False)
    assert rel_err(_np(P), Pr) < 1e-9 and rel_err(_np(alpha), ar) < 1e-9


def test_lq_feedback_errors(hip):
    rng = np.random.default_rng(0)
    g = random_lq_game(rng, 4, [2, 2], 5, 1, pairs=[(0, 0), (0, 1)])  # player 1 has no R_11
    with pytest.raises(hip.IlqgError) as e:
        hip.lq_feedback(dims_of(g, abi.F64), g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    assert e.value.status == abi.ERR_INVALID
    g = random_lq_game(rng, 33, [2, 2], 5, 1)  # past ILQG_MAX_XDIM (any n <= 32 runs: tests/test_gpu_generic.py)
    with pytest.raises(hip.IlqgError) as e:
        hip.lq_feedback(dims_of(g, abi.F64), g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    assert e.value.status == abi.ERR_UNSUPPORTED


def _random_op(spec, rng, B):
    """A plausible operating point: jittered x0 rolled out under small random strategies."""
    T, n, m = spec.T, spec.n, spec.m
    x0 = examples.jittered_x0(spec, B, seed=int(rng.integers(1 << 30)))
    xs_ref = np.tile(x0[:, None, :], (1, T, 1)) + 0.05 * rng.standard_normal((B, T, n))
    us_ref = 0.1 * rng.standard_normal((B, T, m))
    # small gains: large random feedback makes the closed loop unstable and amplifies the
    # last-ulp differences between device libm and glibc by ~1e10 over 100 steps
    P = 0.002 * rng.standard_normal((B, T, m * n))
    alpha = 0.05 * rng.standard_normal((B, T, m))
    return x0, xs_ref, us_ref, P, alpha


@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_linearize_headline_system_matches_reference_python_golden(hip, dtype):
    """ilqg_linearize_batch on BASELINE config 2 / 3's own 14-state system against the (A, B_i) the reference's
    python/product_multiplayer_dynamical_system.py produced (tests/golden/product_dynamics_n14.npz): the device
    against a reference artefact directly, no oracle in between.  Both trajectories as one batch of two.
    fp64 1e-12 absolute; fp32 (the reference's C++ arithmetic) 2e-6 relative to the largest entry."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "product_dynamics_n14.npz"))
    spec = examples.modified_three_player_intersection()
    hp = hip.Problem(spec, dtype)
    xs = np.stack([g["xs_zero"], g["xs_random"]])
    us = np.stack([g["us_zero"], g["us_random"]])
    A_d, B_d = hp.linearize(xs, us)
    A_d = _np(A_d).astype(np.float64).reshape(2, spec.T, 14, 14).transpose(0, 1, 3, 2)   # column-major blocks
    B_d = _np(B_d).astype(np.float64).reshape(2, spec.T, 6, 14).transpose(0, 1, 3, 2)
    tol = 1e-12 if dtype == abi.F64 else 2e-6 * 5.0
    for b, traj in enumerate(("zero", "random")):
        assert np.abs(A_d[b] - g["A_" + traj]).max() < tol
        for i in range(3):
            assert np.abs(B_d[b][:, :, 2 * i:2 * i + 2] - g["B%d_%s" % (i, traj)]).max() < tol


@pytest.mark.parametrize("cfg", list(examples.CONFIGS))
@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_stage_kernels_match_oracle(hip, oracle, cfg, dtype):
    """rollout, linearize, quadraticize, total costs — each vs the oracle on the same inputs."""
    spec = examples.CONFIGS[cfg]()
    rng = np.random.default_rng(7)
    B = 4
    x0, xs_ref, us_ref, P, alpha = _random_op(spec, rng, B)
    scale = np.array([1.0, 0.5, 0.25, 0.1])
    op = oracle.OracleProblem(spec)
    hp = hip.Problem(spec, dtype)
    assert hp.pairs == op.pairs
    xs_o, us_o = op.rollout(dtype, x0, xs_ref, us_ref, P, alpha, scale)
    xs_d, us_d = hp.rollout(x0, xs_ref, us_ref, P, alpha, scale)
    tol = 1e-9 if dtype == abi.F64 else 5e-4  # fp64: device libm vs glibc, 100 chained steps; fp32: 100 chained RK4 steps of device-libm sin/cos/tan
    assert rel_err(_np(xs_d), xs_o) < tol
    assert rel_err(_np(us_d), us_o) < tol
    # downstream stages are compared at the ORACLE's operating point so errors do not chain
    A_o, B_o = op.linearize(dtype, xs_o, us_o)
    A_d, B_d = hp.linearize(xs_o, us_o)
    tol = 1e-12 if dtype == abi.F64 else 1e-5
    assert rel_err(_np(A_d), A_o) < tol and rel_err(_np(B_d), B_o) < tol
    nc = spec.num_constraints
    lam = np.abs(rng.standard_normal((B, max(nc, 1), spec.T))) if nc else None
    mu = np.array([10.0, 11.0, 12.1, 5.0]) if nc else None
    te = rng.integers(0, spec.T, size=(B, len(spec.subsystems))).astype(np.int32)
    Q_o, l_o, R_o, r_o = op.quadraticize(dtype, xs_o, us_o, lam, mu, te)
    Q_d, l_d, R_d, r_d = hp.quadraticize(xs_o, us_o, lam, mu, te)
    tol = 1e-9 if dtype == abi.F64 else 2e-3  # lane-boundary costs amplify ulp differences of hypot/sqrt
    for a, b in ((Q_d, Q_o), (l_d, l_o), (R_d, R_o), (r_d, r_o)):
        assert rel_err(_np(a), b) < tol
    c_o, te_o = op.total_costs(dtype, xs_o, us_o)
    c_d, te_d = hp.total_costs(xs_o, us_o)
    assert rel_err(_np(c_d), c_o) < (1e-10 if dtype == abi.F64 else 1e-4)
    assert np.array_equal(_np(te_d), te_o)


def test_problem_create_rejects_dynamics_the_kernels_do_not_cover(hip):
    """Descriptor validation of ilqg_problem_create: the shared-state kinds only occur as their row pairs, point
    masses only among point masses; the refusal is an error status with a message, never a silent fallback."""
    def spec_of(kinds):
        s = abi.ProblemSpec(T=10)
        for k in kinds:
            s.add_player(k, 1.0)
        for i in range(len(kinds)):
            s.quadratic(i, 1.0, -1, 0.0, control_of=i)
        s.x0 = np.zeros(s.n)
        return s
    for kinds in ((abi.DYN_POINT_MASS_2D, abi.DYN_UNICYCLE_4D), (abi.DYN_UNICYCLE_4D, abi.DYN_POINT_MASS_2D),
                  (abi.DYN_AIR_3D_EVADER, abi.DYN_DUBINS_CAR), (abi.DYN_PLANAR_DISTURBANCE, abi.DYN_UNICYCLE_4D_DISTURBED)):
        with pytest.raises(hip.IlqgError) as e:
            hip.Problem(spec_of(kinds), abi.F64)
        assert e.value.status == abi.ERR_UNSUPPORTED, kinds
        assert str(e.value)
    hip.Problem(spec_of((abi.DYN_POINT_MASS_2D, abi.DYN_POINT_MASS_2D)), abi.F64)  # the covered case builds
    # the plain-RK4 models in a shape whose instantiation does not carry that integrator ((10,2,2) exists, without it)
    # run on the run-time-dimensioned kernels, which pick the integrator from the models
    s5 = spec_of((abi.DYN_UNICYCLE_5D, abi.DYN_UNICYCLE_5D))
    p5 = hip.Problem(s5, abi.F64)
    x0 = np.array([[0.0, 0.0, 0.3, 2.0, 0.0, 3.0, 1.0, -0.2, 1.5, 0.0]])
    z = lambda *shape: np.zeros(shape)  # noqa: E731
    us_ref = 0.1 * np.ones((1, 10, 4))
    xs, _ = p5.rollout(x0, z(1, 10, 10), us_ref, z(1, 10, 40), z(1, 10, 4))
    assert np.all(np.isfinite(_np(xs))) and abs(float(_np(xs)[0, -1, 0])) > 1.0  # it moved
    # a time-dependent cost anywhere but among a player's state costs; speed indices outside the state
    s = spec_of((abi.DYN_CAR_5D, abi.DYN_CAR_5D))
    s.terms.append(dict(s.terms[0], kind=abi.COST_NOMINAL_PATH_LENGTH, idx=(0, 0, 0, 0)))  # a copy of a control cost
    with pytest.raises(hip.IlqgError) as e:
        hip.Problem(s, abi.F64)
    assert e.value.status == abi.ERR_INVALID
    s = spec_of((abi.DYN_CAR_5D, abi.DYN_CAR_5D))
    s.weighted_convex_proximity(0, 1.0, (0, 1), (5, 6), 4, 10, 3.0)  # n = 10: index 10 is past the state
    with pytest.raises(hip.IlqgError) as e:
        hip.Problem(s, abi.F64)
    assert e.value.status == abi.ERR_INVALID


def _clean(ref, max_bt=12):
    """Instances whose line search never went below step ~ alpha0 * 2^-12.  Deeper back-tracking
    means the Armijo test `last - merit >= frac*step*ED` is decided by the last bits of two
    ~1e5 merit values (the reference's own behaviour is not reproducible there),
    so accept/reject legitimately flips between any two correct implementations."""
    bt = np.nan_to_num(ref["log"][:, :, 3], nan=0.0)
    return np.where((bt.max(axis=1) <= max_bt) & (ref["status"] == 1))[0]


@pytest.mark.parametrize("cfg", ["modified_three_player_intersection", "three_player_intersection",
                                 "three_player_collision_avoidance_reachability", "two_player_unicycle_4d_scene",
                                 "two_player_reachability", "skeleton", "three_player_overtaking",
                                 "one_player_reachability", "dubins_origin", "air_3d", "modified_air_3d",
                                 "cost_zoo_scene", "dynamics_zoo_scene", "delayed_dubins_scene"])
def test_ilq_solve_matches_oracle_fp64(hip, oracle, cfg):
    """Whole iLQ loop, fp64, fixed iteration count.  Where the line search is well conditioned the
    device makes the oracle's accept/reject decisions, so trajectories, strategies and costs agree
    to fp64 accumulation error."""
    spec = examples.CONFIGS[cfg]()
    spec.params.initial_alpha_scaling = 0.1 if cfg != "modified_three_player_intersection" else 0.5
    spec.params.expected_decrease_fraction = 0.001
    # reachability's line search is noise-limited from iteration 2; so is the overtaking example's (steering-rate
    # weight 5e5 against O(1) costs)
    B, K = 12, (1 if ("reachability" in cfg or "overtaking" in cfg or cfg.endswith("air_3d")) else 6)
    x0 = examples.jittered_x0(spec, B, seed=11)
    ref = oracle.OracleProblem(spec).solve(abi.F64, x0, fixed_iters=K, merit_log_len=K)
    out = hip.Problem(spec, abi.F64).solve(x0, fixed_iters=K)
    ok = _clean(ref)
    assert len(ok) >= 2, "test instances are all ill-conditioned"
    assert np.array_equal(_np(out["iters"])[ok], ref["iters"][ok])
    assert np.array_equal(_np(out["status"])[ok], ref["status"][ok])
    same = (_np(out["status"]) == ref["status"]) & (_np(out["iters"]) == ref["iters"])
    # ... the agreement statistic over more instances than the twelve compared above (round 6: on twelve, one coin flip is
    # 8 % — scripts/diag/free_running_agreement.py on 96 instances of cost_zoo_scene: device vs oracle 0.80, oracle vs
    # itself 0.84-0.90, while the first twelve alone gave 0.50 on one build and 0.67 on the next)
    x0b = examples.jittered_x0(spec, 36, seed=12)
    refb = oracle.OracleProblem(spec).solve(abi.F64, x0b, fixed_iters=K, merit_log_len=K, threads=4)
    outb = hip.Problem(spec, abi.F64).solve(x0b, fixed_iters=K)
    sameb = (_np(outb["status"]) == refb["status"]) & (_np(outb["iters"]) == refb["iters"])
    agree = np.mean(np.concatenate([same, sameb]))
    # How many instances CAN end the same way is measured, not guessed: the oracle is run again from x0 nudged by
    # 1e-12; an instance whose two oracle runs end differently has a line search that is decided by rounding (deep
    # back-tracking or a failing search: accept / reject hangs on the last bits of two ~1e5 merit values), and two
    # correct implementations differ there as the oracle differs from itself.  Every scene is fully stable except
    # SkeletonExample (no regularisation, proximity cost switching on mid-horizon) and DubinsOrigin.  The device may
    # lose at most two more instances than the oracle loses against itself; the instance-by-instance comparison at
    # forced steps is test_gpu_forced.py.
    xall = np.concatenate([x0, x0b])
    nudged = oracle.OracleProblem(spec).solve(abi.F64, xall + 1e-12 * np.random.default_rng(5).standard_normal(xall.shape),
                                              fixed_iters=K, merit_log_len=K, threads=4)
    stable = (nudged["status"] == np.concatenate([ref["status"], refb["status"]])) & \
        (nudged["iters"] == np.concatenate([ref["iters"], refb["iters"]]))
    assert np.mean(stable) >= 0.5, "the scene is too ill-conditioned to test (%.2f of the oracle's own runs agree)" % np.mean(stable)
    # the device may lose at most a tenth of the instances more than the oracle loses against itself
    assert agree >= np.mean(stable) - 0.1, "too many instances end differently (%.2f agree, oracle vs itself %.2f)" % (agree, np.mean(stable))
    depth = np.nan_to_num(ref["log"][:, :, 3], nan=0.0).max(axis=1)
    for b in np.where(~same)[0]:
        assert depth[b] > 12 or ref["status"][b] == 0 or _np(out["status"])[b] == 0, \
            "instance %d ends differently after a shallow line search (depth %d)" % (b, depth[b])
    assert rel_err(_np(out["xs"])[ok], ref["xs"][ok]) < 1e-7
    assert rel_err(_np(out["us"])[ok], ref["us"][ok]) < 1e-7
    assert rel_err(_np(out["P"])[ok], ref["P"][ok]) < 1e-6      # north_star: P_t, alpha_t within 1e-6 rel-err
    assert rel_err(_np(out["alpha"])[ok], ref["alpha"][ok]) < 1e-6
    assert rel_err(_np(out["costs"])[ok], ref["costs"][ok]) < 1e-8
    # every instance, clean or not, must come back finite with a sane status word
    assert np.isfinite(_np(out["xs"])).all() and set(_np(out["status"]).tolist()) <= {0, 1}



@pytest.mark.parametrize("cfg", ["two_player_collision_avoidance_reachability", "two_player_collision"])
def test_ilq_iteration_on_an_unstable_closed_loop_fp64(hip, oracle, cfg):
    """TwoPlayerCollisionAvoidanceReachabilityExample (n=10, no regularisation): the first LQ solution's gains make
    the closed-loop rollout amplify rounding by ~2.5x per step (measured: 2e-15 at step 0, 7e-13 at step 5), so
    after 100 steps device and oracle trajectories are unrelated although nothing is wrong.  TwoPlayerCollisionExample
    (n=12, lane-boundary weights of 5e7, FinalTimeCost goals) behaves the same way.  What is well defined is
    compared: the strategies of the iteration (1e-9) and the first steps of the rollout."""
    spec = examples.CONFIGS[cfg]()
    spec.params.initial_alpha_scaling = 0.1
    spec.params.expected_decrease_fraction = 0.001
    B = 6
    x0 = examples.jittered_x0(spec, B, seed=11)
    ref = oracle.OracleProblem(spec).solve(abi.F64, x0, fixed_iters=1, merit_log_len=1)
    out = hip.Problem(spec, abi.F64).solve(x0, fixed_iters=1)
    assert np.array_equal(_np(out["iters"]), ref["iters"]) and np.array_equal(_np(out["status"]), ref["status"])
    assert rel_err(_np(out["P"]), ref["P"]) < 1e-9
    assert rel_err(_np(out["alpha"]), ref["alpha"]) < 1e-9
    assert rel_err(_np(out["xs"])[:, :8], ref["xs"][:, :8]) < 1e-9
    assert rel_err(_np(out["us"])[:, :8], ref["us"][:, :8]) < 1e-9
    assert np.isfinite(_np(out["xs"])).all()


@pytest.mark.parametrize("T,B", [(37, 3), (2, 1), (100, 65)])
def test_ilq_solve_odd_horizons_and_batches_fp64(hip, oracle, T, B):
    """Horizons that are not a multiple of anything the kernels stage in groups of (forward-pass groups of
    6-8 steps, two waves sharing the rows) and batches of 1 / 3 / 65 instances: same parity bar."""
    spec = examples.modified_three_player_intersection(T=T)
    spec.params.initial_alpha_scaling = 0.5
    spec.params.expected_decrease_fraction = 0.001
    K = 3
    x0 = examples.jittered_x0(spec, B, seed=5)
    ref = oracle.OracleProblem(spec).solve(abi.F64, x0, fixed_iters=K, merit_log_len=K)
    out = hip.Problem(spec, abi.F64).solve(x0, fixed_iters=K)
    ok = _clean(ref)
    assert len(ok) >= 1
    assert np.array_equal(_np(out["iters"])[ok], ref["iters"][ok])
    assert rel_err(_np(out["xs"])[ok], ref["xs"][ok]) < 1e-7
    assert rel_err(_np(out["P"])[ok], ref["P"][ok]) < 1e-6
    assert rel_err(_np(out["alpha"])[ok], ref["alpha"][ok]) < 1e-6
    assert rel_err(_np(out["costs"])[ok], ref["costs"][ok]) < 1e-8


def test_ilq_solve_fp32_tracks_fp64_oracle(hip, oracle):
    """The fp32 instantiation (the reference's own precision) of the whole loop against the fp64 oracle: two
    iterations from the same start stay within single-precision distance of it."""
    spec = examples.modified_three_player_intersection()
    spec.params.initial_alpha_scaling = 0.5
    spec.params.expected_decrease_fraction = 0.001
    B, K = 8, 2
    x0 = examples.jittered_x0(spec, B, seed=9)
    ref = oracle.OracleProblem(spec).solve(abi.F64, x0, fixed_iters=K, merit_log_len=K)
    out = hip.Problem(spec, abi.F32).solve(x0.astype(np.float32), fixed_iters=K)
    ok = _clean(ref)
    same = np.array([b for b in ok if _np(out["iters"])[b] == ref["iters"][b]])
    assert len(same) >= 2
    assert rel_err(_np(out["xs"])[same].astype(np.float64), ref["xs"][same]) < 2e-3
    assert rel_err(_np(out["costs"])[same].astype(np.float64), ref["costs"][same]) < 2e-2
    assert np.isfinite(_np(out["P"])).all()


def test_ilq_solve_free_running_matches_oracle_fp64(hip, oracle):
    """Reference semantics (convergence test + line-search failure) on the example's own params."""
    spec = examples.modified_three_player_intersection()
    B = 16
    x0 = examples.jittered_x0(spec, B, seed=3)
    # three draws: with one, instances 3 and 12 of this batch pass as stable although a second nudge of the same size
    # flips them in the oracle itself (measured: 11 stable after one draw, 7 after two or three)
    ref, stable = oracle_with_stability(oracle.OracleProblem(spec), abi.F64, x0, keys=("iters", "status", "converged"),
                                        merit_log_len=16, draws=3)
    out = hip.Problem(spec, abi.F64).solve(x0)
    # With the example's own expected_decrease_fraction = 0.9 the reference's line search fails early for most
    # instances (status 0, last accepted iterate returned) — reproduced.  A FAILED line search walks through all 100
    # step sizes, so somewhere on the way the Armijo test is decided by rounding noise and a correct implementation
    # may accept where another rejects — as the oracle does against itself from a 1e-12 nudge of x0.  The comparison
    # is on the instances whose outcome survives that nudge (helpers.oracle_with_stability): all of them but one must
    # end the same way on the device, and every one that does must match to the parity bar.
    same = (_np(out["status"]) == ref["status"]) & (_np(out["iters"]) == ref["iters"]) & \
        (_np(out["converged"]) == ref["converged"])
    assert stable.sum() >= 4, "the test instances are all decided by rounding"
    assert (same & stable).sum() >= stable.sum() - 1, (same, stable)
    ok = np.where(same & stable)[0]
    assert rel_err(_np(out["xs"])[ok], ref["xs"][ok]) < 1e-7
    assert rel_err(_np(out["P"])[ok], ref["P"][ok]) < 1e-6
    assert rel_err(_np(out["costs"])[ok], ref["costs"][ok]) < 1e-8


def test_augmented_lagrangian_solve_matches_oracle_fp64(hip, oracle):
    """ilqg_al_solve_batch vs the oracle's AugmentedLagrangianSolver restatement on the constrained
    three-player intersection (n=16, six ProximityConstraints), fp64, a 30-iterate log budget.
    Multiplier updates, mu schedule, warm restarts and failure down-scaling must all line up for the
    final iterate to agree; instances whose line searches are noise-limited are excluded as above."""
    spec = examples.three_player_intersection()
    spec.params.max_solver_iters = 30
    spec.params.unconstrained_solver_max_iters = 5
    B = 12
    x0 = examples.jittered_x0(spec, B, seed=21)
    # (three draws: with one, instance 9 passes as stable although another 1e-12 nudge moves it in the oracle itself)
    ref, stable = oracle_with_stability(oracle.OracleProblem(spec), abi.F64, x0, augmented_lagrangian=True, draws=3)
    out = hip.Problem(spec, abi.F64).solve(x0, augmented_lagrangian=True)
    same = (_np(out["iters"]) == ref["iters"]) & (_np(out["status"]) == ref["status"])
    # all but one of the instances whose outcome survives a 1e-12 nudge of x0 in the oracle itself end the same way on
    # the device, and each of those matches to the parity bar
    assert stable.sum() >= 3, "the test instances are all decided by rounding"
    assert (same & stable).sum() >= stable.sum() - 1, (same, stable, _np(out["iters"]), ref["iters"])
    good = np.where(same & stable)[0]
    for b in good:
        assert rel_err(_np(out["xs"])[b], ref["xs"][b]) < 1e-6, b
    assert rel_err(_np(out["costs"])[good], ref["costs"][good]) < 1e-6
    assert np.isfinite(_np(out["xs"])).all()


def test_augmented_lagrangian_max_runtime_bounds_the_outer_loop(hip):
    """AugmentedLagrangianSolver::Solve(success, max_runtime), src/augmented_lagrangian_solver.cpp:85-110: the first
    inner solve gets max_runtime / max_solver_iters, `elapsed` STARTS at that allowance and the outer loop runs only
    while elapsed < max_runtime - RuntimeUpperBound() (0.02 s until the loop timer holds two samples).  With a budget of
    1 ms neither an inner iteration nor an outer iteration fits: one logged iterate (the initial operating point), no
    restart — and success = 0 because the constraints are not met (:188-191).  A generous budget changes nothing."""
    spec = examples.three_player_intersection()
    spec.params.max_solver_iters = 30
    spec.params.unconstrained_solver_max_iters = 5
    x0 = examples.jittered_x0(spec, 6, seed=21)
    prob = hip.Problem(spec, abi.F64)
    tight = prob.solve(x0, augmented_lagrangian=True, max_runtime=1e-3)
    assert np.array_equal(_np(tight["iters"]), np.ones(6, dtype=np.int32)), _np(tight["iters"])
    assert not _np(tight["status"]).any()
    plain = hip.Problem(spec, abi.F64).solve(x0, augmented_lagrangian=True)
    relaxed = hip.Problem(spec, abi.F64).solve(x0, augmented_lagrangian=True, max_runtime=1e4)
    assert np.array_equal(_np(relaxed["iters"]), _np(plain["iters"]))
    assert np.array_equal(_np(relaxed["xs"]), _np(plain["xs"]))
    assert _np(plain["iters"]).min() > 1


def test_equality_flag_is_refused_on_non_affine_constraints(hip):
    """ILQG_FLAG_EQUALITY (Constraint::is_equality_) is only carried for the affine constraints; on a proximity
    constraint ilqg_problem_create refuses it instead of running an unclipped multiplier behind an inequality's gate."""
    spec = examples.three_player_intersection()
    t = next(t for t in spec.terms if t["constraint_slot"] >= 0)
    t["flags"] |= abi.FLAG_EQUALITY
    with pytest.raises(Exception, match="EQUALITY"):
        hip.Problem(spec, abi.F64)


@pytest.mark.parametrize("cfg,al,dtype", [("modified_three_player_intersection", False, abi.F64),
                                          ("modified_three_player_intersection", False, abi.F32),
                                          ("three_player_intersection", True, abi.F64),
                                          ("roundabout_merging", False, abi.F64),
                                          ("one_player_reachability", True, abi.F64),
                                          ("cost_zoo_scene", True, abi.F64), ("cost_zoo_scene", False, abi.F32)])
def test_split_trial_pass_is_the_fused_kernel_bit_for_bit(hip, cfg, al, dtype):
    """The three-launch form of the trial pass (rollout / rows / decision kernels, chosen by problem size or by
    ilqg_solve_options::split_trial) runs the same functions on the same data as the fused kernel: free-running solves — line
    searches with back-tracking, convergence exits, the augmented-Lagrangian restarts — must come back identical
    in every output, status word and iteration count."""
    spec = examples.CONFIGS[cfg]()
    spec.params.max_solver_iters = 12
    spec.params.unconstrained_solver_max_iters = 4
    B = 9
    x0 = examples.jittered_x0(spec, B, seed=3)
    outs = []
    # every pass in the fused kernel / fused first pass, back-tracking handed to split passes with the speculative
    # line search (the default for free-running solves) / split passes throughout / split passes without probing
    for split, handoff, probe in ((False, False, True), (False, True, True), (True, True, True), (True, True, False)):
        out = hip.Problem(spec, dtype).solve(x0, augmented_lagrangian=al, split_trial=split, handoff=handoff, probe=probe)
        outs.append({k: _np(v).copy() for k, v in out.items() if hasattr(v, "shape") and k != "ws"})
    fused = outs[0]
    for other in outs[1:]:
        assert set(fused) == set(other)
        for k in fused:
            assert np.array_equal(fused[k], other[k], equal_nan=True), k
    assert fused["iters"].max() >= 2


@pytest.mark.parametrize("cfg,al,dtype", [("modified_three_player_intersection", False, abi.F64),
                                          ("modified_three_player_intersection", False, abi.F32),
                                          ("three_player_intersection", True, abi.F64),
                                          ("three_player_intersection", False, abi.F32),
                                          ("three_player_collision_avoidance_reachability", False, abi.F64),
                                          ("roundabout_merging", False, abi.F64)])
def test_probing_rollouts_with_a_lane_per_subsystem_are_the_paired_ones_bit_for_bit(hip, cfg, al, dtype):
    """ilqg_solve_options::probe_lanes: the speculative line search's rollouts with 64 / N candidates of an instance per
    wavefront, a lane per (candidate, subsystem) walking the eight RK4 stages in sequence (rollout_lanes,
    sub_integrate_stages_seq), against the form with two candidates per wavefront and a lane per stage — and against no
    probing at all.  A probed trajectory is handed over in place of the regular pass's rollout, so every output of a
    free-running solve — line searches that back-track, fail, diverge past the fast trigonometric range — must come back
    identical.  (n = 14 own parameters: the line search fails at iteration 2; n = 16: ~10 % of the instances back-track
    tens of steps; the reachability scene: 35 rejected steps per iteration; n = 24: four players, 16 candidates per wave.)"""
    spec = examples.CONFIGS[cfg]()
    spec.params.max_solver_iters = 8
    spec.params.unconstrained_solver_max_iters = 4
    B = 40
    x0 = examples.jittered_x0(spec, B, seed=5)
    outs = []
    for kw in (dict(probe=True, probe_lanes=True), dict(probe=True, probe_lanes=False), dict(probe=False)):
        out = hip.Problem(spec, dtype).solve(x0, augmented_lagrangian=al, split_trial=True, **kw)
        outs.append({k: _np(v).copy() for k, v in out.items() if hasattr(v, "shape") and k != "ws"})
    st = hip.Problem(spec, dtype)
    o = st.solve(x0, augmented_lagrangian=al, split_trial=True, probe=True, probe_lanes=True)
    assert int(_np(st.solve_state(o, augmented_lagrangian=al)["backtracks"]).sum()) > B // 4   # the line searches did back-track
    for other in outs[1:]:
        for k in outs[0]:
            assert np.array_equal(outs[0][k], other[k], equal_nan=True), (k, cfg)


def test_augmented_lagrangian_with_a_polyline_constraint_fp64(hip, oracle):
    """Polyline2SignedDistanceConstraint through AugmentedLagrangianSolver (examples.cost_zoo_scene: a wall that bulges
    into player 1's lane makes the constraint active mid-horizon, next to the other cost kinds of that scene)."""
    spec = examples.cost_zoo_scene()
    spec.params.max_solver_iters = 30
    spec.params.unconstrained_solver_max_iters = 5
    B = 12
    x0 = examples.jittered_x0(spec, B, seed=21)
    # thirty chained inner solves that each end in a failed line search: three nudged oracle runs pick the instances
    # whose trajectories are reproducible at all (one run calls instance 3 stable, which then differs from itself by
    # 2e-4 under another 1e-13 nudge)
    ref, stable = oracle_with_stability(oracle.OracleProblem(spec), abi.F64, x0, draws=3, augmented_lagrangian=True)
    out = hip.Problem(spec, abi.F64).solve(x0, augmented_lagrangian=True)
    same = (_np(out["iters"]) == ref["iters"]) & (_np(out["status"]) == ref["status"])
    assert stable.sum() >= 3, "the test instances are all decided by rounding"
    assert (same & stable).sum() >= stable.sum() - 1, (same, stable, _np(out["iters"]), ref["iters"])
    good = np.where(same & stable)[0]
    for b in good:
        assert rel_err(_np(out["xs"])[b], ref["xs"][b]) < 1e-6, b
    assert rel_err(_np(out["costs"])[good], ref["costs"][good]) < 1e-6
    assert np.isfinite(_np(out["xs"])).all()


def test_augmented_lagrangian_single_player_dubins_fp64(hip, oracle):
    """OnePlayerReachabilityExample through AugmentedLagrangianSolver: one player, one control, two box constraints on
    it, a max-over-time cost — the N = 1, m = 1 corner of every kernel.  Log length, flags and costs must agree
    with the oracle on most instances (its line searches are noise-limited like the other reachability games)."""
    spec = examples.one_player_reachability()
    spec.params.max_solver_iters = 30
    spec.params.unconstrained_solver_max_iters = 5
    B = 8
    x0 = examples.jittered_x0(spec, B, seed=21)
    ref, stable = oracle_with_stability(oracle.OracleProblem(spec), abi.F64, x0, keys=("iters", "status", "converged"),
                                        augmented_lagrangian=True)
    out = hip.Problem(spec, abi.F64).solve(x0, augmented_lagrangian=True)
    same = (_np(out["iters"]) == ref["iters"]) & (_np(out["status"]) == ref["status"]) & \
        (_np(out["converged"]) == ref["converged"])
    assert stable.sum() >= 2, "the test instances are all decided by rounding"
    assert (same & stable).sum() >= stable.sum() - 1, (same, stable, _np(out["iters"]), ref["iters"])
    for b in np.where(same & stable)[0]:
        assert rel_err(_np(out["xs"])[b], ref["xs"][b]) < 1e-6, b
    assert np.isfinite(_np(out["xs"])).all()


@pytest.mark.parametrize("scene", ["modified_three_player_intersection", "dubins_origin", "delayed_dubins_scene"])
@pytest.mark.parametrize("t0,runtime", [(0.33, 0.25), (0.0, 0.1), (1.07, 0.0), (2.5, 0.4)])
def test_receding_horizon_shift_matches_oracle_fp64(hip, oracle, t0, runtime, scene):
    """Problem::SetUpNextRecedingHorizon on device vs the oracle's restatement: same nearest-state index, same
    shifted / zero-extended / re-propagated plan, same stitched initial state; then the warm-started solve from
    it reproduces the oracle's.  dubins_origin: an ego whose model inherits the default DistanceBetween (the squared
    norm of its whole state, heading included — single_player_dynamical_system.h:69) instead of a position metric;
    delayed_dubins_scene: the same with four states, through the plain-RK4 integrator."""
    spec = examples.CONFIGS[scene]()
    spec.params.initial_alpha_scaling = 0.5
    spec.params.expected_decrease_fraction = 0.001
    B = 6
    x0 = examples.jittered_x0(spec, B, seed=21)
    prob = hip.Problem(spec, abi.F64)
    bufs = prob.solve(x0, fixed_iters=3)
    plan = {k: _np(bufs[k]).copy() for k in ("xs", "us", "P", "alpha")}
    # a measured state near where the plan says the players are at t0, perturbed
    k_meas = int(t0 / spec.dt)
    rng = np.random.default_rng(4)
    x_meas = plan["xs"][:, k_meas, :] + 0.05 * rng.standard_normal((B, prob.n))
    ref = oracle.OracleProblem(spec).receding_horizon_shift(abi.F64, x_meas, t0, runtime, 0.0, plan["xs"], plan["us"],
                                                            plan["P"], plan["alpha"])
    x0n, first, new_t0 = prob.receding_horizon_shift(x_meas, t0, runtime, 0.0, bufs)
    assert np.array_equal(_np(first), ref["first_step"])
    assert abs(new_t0 - ref["new_plan_t0"]) < 1e-12 and abs(t0 + runtime - new_t0) <= spec.dt + 1e-9
    assert rel_err(_np(x0n), ref["x0_next"]) < 1e-12
    for k in ("xs", "us", "P", "alpha"):
        assert rel_err(_np(bufs[k]), ref[k]) < 1e-12, k
    # zero strategies in the re-propagated tail, as the reference leaves them
    f = int(ref["first_step"][0])
    if f > 0:
        assert np.all(_np(bufs["P"])[0, prob.T - f:] == 0) and np.all(_np(bufs["us"])[0, prob.T - f:] == 0)
    # the next solve, warm-started from the shifted plan, agrees with the oracle's
    nxt = oracle.OracleProblem(spec).solve(abi.F64, ref["x0_next"], xs=ref["xs"], us=ref["us"], P=ref["P"],
                                           alpha=ref["alpha"], fixed_iters=2, merit_log_len=2)
    out = prob.solve(x0n, bufs, fixed_iters=2)
    ok = _clean(nxt)
    assert len(ok) >= 2
    assert rel_err(_np(out["xs"])[ok], nxt["xs"][ok]) < 1e-7
    assert rel_err(_np(out["P"])[ok], nxt["P"][ok]) < 1e-6


def test_receding_horizon_shift_rejects_times_outside_the_plan(hip):
    spec = examples.modified_three_player_intersection()
    prob = hip.Problem(spec, abi.F64)
    x0 = examples.jittered_x0(spec, 2, seed=1)
    bufs = prob.solve(x0, fixed_iters=1)
    with pytest.raises(hip.IlqgError):
        prob.receding_horizon_shift(x0, -0.5, 0.1, 0.0, bufs)   # t0 before the plan (problem.cpp:70)
    with pytest.raises(hip.IlqgError):
        prob.receding_horizon_shift(x0, 9.95, 0.2, 0.0, bufs)   # t0 + runtime past the horizon (:69)


@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_instances_sharing_a_wavefront_in_the_split_rollout_do_not_see_each_other(hip, dtype):
    """The split rollout kernels integrate two trajectories per wavefront (rollout_pair, csrc/ilqg_stages.hpp): an
    instance's result must not depend on who shares its wavefront.  The n = 16 intersection with its own line-search
    parameters has instances whose rejected trial steps diverge (headings past the fast trigonometric range, where
    the library fall-back differs in the last bits) next to instances that do not — the pairing that showed a
    wavefront-wide fall-back vote changing the partner's last bits.  Every instance solved alone, in pairs in both
    orders, and in the whole batch must give the same bits."""
    spec = examples.CONFIGS["three_player_intersection"]()
    spec.params.max_solver_iters = 4
    B = 9
    x0 = examples.jittered_x0(spec, B, seed=3)
    keys = ("xs", "us", "P", "alpha", "costs", "iters", "status")

    def solve(rows):
        out = hip.Problem(spec, dtype).solve(x0[rows], split_trial=True, probe=False)
        return {k: _np(out[k]).copy() for k in keys}
    whole = solve(list(range(B)))
    for i in range(B):
        alone = solve([i])
        for k in keys:
            assert np.array_equal(alone[k][0], whole[k][i], equal_nan=True), (i, k)
    for a, b in ((2, 3), (3, 2), (6, 7), (0, 8)):
        pair = solve([a, b])
        for k in keys:
            assert np.array_equal(pair[k][0], whole[k][a], equal_nan=True), (a, b, k)
            assert np.array_equal(pair[k][1], whole[k][b], equal_nan=True), (a, b, k)


@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_rollout_trig_across_and_beyond_the_fast_range(hip, oracle, dtype):
    """The rollouts integrate with range-limited sin / cos / tan kernels (three-piece Cody-Waite reduction up to
    kTrigFastLimit) and take a reduction of their own beyond (csrc/ilqg_trig.hpp: the same form with a wide quadrant up to
    2^47, Payne-Hanek from there to the largest finite argument; fp32 beyond its limit goes through the double forms).
    Headline system (Car5D, Car5D, Unicycle4D): instance 0's unicycle heading crosses the limit in mid-horizon, instance
    1's car steering angle does (the tangent), instance 2's second car starts at three times the limit (the middle
    form), instance 3 never leaves the fast range and shares kernels with the others, instance 4's unicycle heads 1e15
    (Payne-Hanek; a heading that large only stays comparable while it does not change: zero turn rate).  The stand-alone
    rollout against the oracle — 1e-9 where the heading is constant, the heading's own resolution (ulp(1e9) = 1.2e-7 rad,
    accumulated over the horizon) where it moves —, then whole iterations on the fused trial kernel (one trajectory per
    wavefront) and the split one (two per wavefront): instance 3 against the oracle and bit for bit as when solved alone."""
    spec = examples.modified_three_player_intersection()
    spec.params.expected_decrease_fraction = 0.001
    spec.params.initial_alpha_scaling = 0.1
    f64 = dtype == abi.F64
    limit = 1.0e9 if f64 else 2.0e3  # kTrigFastLimit / kTrigFastLimitF, csrc/ilqg_trig.hpp
    B, T, n, m = 5, spec.T, spec.n, spec.m
    x0 = examples.jittered_x0(spec, B, seed=11)
    us_ref = np.zeros((B, T, m))
    # state: car (x, y, theta, phi, v) x 2, unicycle (x, y, theta, v); controls (phi rate, a) x 2, (omega, a)
    x0[0, 12] += limit - 3.0
    us_ref[0, :, 4] = 1.0           # + 10 rad over the horizon: crosses after ~ 30 steps
    # car 1's steering angle crosses after ~ 40 steps, on a stretch that holds no pole of the tangent
    # (1e9 = 318309886 pi + 0.577: poles at 1e9 - 2.148 and 1e9 + 0.994;  2e3 = 636 pi + 1.947: poles at 2e3 - 0.376, + 2.765)
    x0[1, 3] = limit - (0.5 if f64 else 0.05)
    us_ref[1, :, 0] = 0.11 if f64 else 0.012
    x0[2, 7] += 3.0 * limit         # car 2's heading beyond the range from the first step (its steering angle is zero)
    x0[4, 12] = 1.0e15              # beyond 2^47
    assert x0[2, 8] == 0.0
    hd = [2, 3, 7, 8, 12]
    rest = [i for i in range(n) if i not in hd]
    xs_ref = np.tile(x0[:, None, :], (1, T, 1))
    z = lambda *s: np.zeros(s)  # noqa: E731
    op = oracle.OracleProblem(spec)
    hp = hip.Problem(spec, dtype)
    xs_o, us_o = op.rollout(dtype, x0, xs_ref, us_ref, z(B, T, m * n), z(B, T, m))
    xs_d, us_d = hp.rollout(x0, xs_ref, us_ref, z(B, T, m * n), z(B, T, m))
    assert np.abs(xs_o[0, :, 12]).max() > limit + 5 and np.abs(xs_o[0, 0, 12]) < limit   # the crossing is in the horizon
    assert np.abs(xs_o[1, :, 3]).max() > limit + 0.05 and np.abs(xs_o[1, 0, 3]) < limit
    moving = 1e-4 if f64 else 2e-3   # a heading that moves at the limit, known to its ulp
    fixed = 1e-9 if f64 else 2e-3
    for b, tol in enumerate((moving, moving, fixed, fixed, fixed)):
        assert rel_err(_np(xs_d)[b][:, rest], xs_o[b][:, rest]) < tol, b
    assert np.array_equal(_np(xs_d)[4, :, 12], xs_o[4, :, 12]) and np.array_equal(_np(xs_d)[2, :, 7], xs_o[2, :, 7])
    assert np.abs(_np(xs_d)[:4][:, :, hd] - xs_o[:4][:, :, hd]).max() < (1e-3 if f64 else 5e-2)  # a heading driven by the tangent of a moving angle at the limit
    # instance 3 beside the others and alone: the same bits
    xs_a, _ = hp.rollout(x0[3:4], xs_ref[3:4], us_ref[3:4], z(1, T, m * n), z(1, T, m))
    assert np.array_equal(_np(xs_a)[0], _np(xs_d)[3])
    # whole iterations: fused trial kernel (one trajectory per wavefront) and the split form (two per wavefront)
    ref = op.solve(dtype, x0, fixed_iters=2)
    assert np.all(np.isfinite(ref["xs"][3]))
    for kw in (dict(), dict(split_trial=True)):
        out = hp.solve(x0, fixed_iters=2, **kw)
        xd = _np(out["xs"])
        assert rel_err(xd[3], ref["xs"][3]) < (1e-6 if f64 else 2e-2), kw
        alone = hp.solve(x0[3:4], fixed_iters=2, **kw)
        assert np.array_equal(_np(alone["xs"])[0], xd[3]), kw
