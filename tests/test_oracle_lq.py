"""Pins the CPU oracle's LQ sweeps (oracle/ilqg_oracle.hpp) against the reference.

 * golden vectors produced by the reference's own numpy solver python/solve_lq_game.py
   (tests/golden/make_golden.py),
 * test/test_lq_solver.cpp:292-317 (P[0] equals Lyapunov iterations of the coupled DARE),
 * test/test_lq_solver.cpp:319-387 (feedback / open-loop Nash properties, with and
   without linear cost terms) via direct perturbation of the LQ cost,
 * test/test_lq_solver.cpp:389-434 (single player: open-loop u0 ~ feedback u0).
"""
import numpy as np
import pytest

from ilqgames_amd import abi
from helpers import colmajor, dims_of, load_golden_lq, random_lq_game, rel_err


@pytest.mark.parametrize("name", ["lq_feedback_random.npz", "lq_feedback_unicycle.npz",
                                  "lq_feedback_pointmass.npz"])
def test_feedback_matches_reference_python_fp64(oracle, name):
    g = load_golden_lq(name)
    d = dims_of(g, abi.F64, adaptive=False)
    P, alpha, _, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    # fp64 Householder QR vs numpy lstsq: agreement to ~1e-10 relative (conditioning of S)
    assert rel_err(P, g["P_ref"]) < 1e-9
    assert rel_err(alpha, g["alpha_ref"]) < 1e-9
    assert np.all(P[:, -1] == 0) and np.all(alpha[:, -1] == 0)  # strategy.h:64-70, loop from T-2


@pytest.mark.parametrize("name", ["lq_feedback_random.npz", "lq_feedback_unicycle.npz"])
def test_feedback_fp32_close_to_reference_python(oracle, name):
    g = load_golden_lq(name, np.float32)
    d = dims_of(g, abi.F32, adaptive=False)
    P, alpha, _, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    # the reference's own arithmetic (Eigen fp32): ~1e-4 relative
    assert rel_err(P, g["P_ref"]) < 2e-4
    assert rel_err(alpha, g["alpha_ref"]) < 2e-4


def _pointmass(nominal, T=100):
    """TwoPlayerPointMass1D + ConstructCostsWithNominal, test/test_lq_solver.cpp:143-186,227-264,
    quadraticized at the zero operating point."""
    dt = 0.1
    A = np.eye(2) + np.array([[0.0, 1.0], [0.0, 0.0]]) * dt
    B = np.array([[0.05 * dt, 0.032 * dt], [1.0 * dt, 0.11 * dt]])
    w = np.array([[1.0, 0.1], [0.1, 1.0]])  # w[i][j] weight of R_ij ; state weights (1, 0.1)
    wq = [1.0, 0.1]
    pairs = [(0, 0), (0, 1), (1, 0), (1, 1)]
    g = dict(n=2, ms=[1, 1], T=T, N=2, pairs=pairs)
    g["A"] = np.tile(colmajor(A), (1, T, 1))
    g["Bm"] = np.tile(colmajor(B), (1, T, 1))
    g["Q"] = np.tile(np.stack([colmajor(wq[i] * np.eye(2)) for i in range(2)]), (1, T, 1, 1))
    g["l"] = np.tile(np.stack([wq[i] * (np.zeros(2) - nominal) for i in range(2)]), (1, T, 1, 1))
    g["R"] = np.tile(np.array([w[i][j] for i, j in pairs]), (1, T, 1))
    g["r"] = np.tile(np.array([w[i][j] * (0.0 - nominal) for i, j in pairs]), (1, T, 1))
    return g, A, B, wq, w


def _lyapunov(A, B1, B2, Q1, Q2, R11, R12, R21, R22, iters=100):
    """SolveLyapunovIterations, test/test_lq_solver.cpp:72-109."""
    Z1, Z2 = Q1.copy(), Q2.copy()
    P1 = np.linalg.solve(R11 + B1.T @ Z1 @ B1, B1.T @ Z1 @ A)
    P2 = np.linalg.solve(R22 + B2.T @ Z2 @ B2, B2.T @ Z2 @ A)
    for _ in range(iters):
        o1, o2 = P1, P2
        P1 = np.linalg.solve(R11 + B1.T @ Z1 @ B1, B1.T @ Z1 @ (A - B2 @ o2))
        P2 = np.linalg.solve(R22 + B2.T @ Z2 @ B2, B2.T @ Z2 @ (A - B1 @ o1))
        F = A - B1 @ P1 - B2 @ P2
        Z1 = F.T @ Z1 @ F + P1.T @ R11 @ P1 + P2.T @ R12 @ P2 + Q1
        Z2 = F.T @ Z2 @ F + P1.T @ R21 @ P1 + P2.T @ R22 @ P2 + Q2
    return P1, P2


@pytest.mark.parametrize("dtype", [abi.F32, abi.F64])
def test_feedback_matches_lyapunov_iterations(oracle, dtype):
    g, A, B, wq, w = _pointmass(0.0)
    d = dims_of(g, dtype, adaptive=True)  # the reference solver's default
    P, _, _, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    P0 = P[0, 0].reshape(2, 2, order="F")  # stacked (m x n)
    I1 = np.eye(1)
    P1, P2 = _lyapunov(A, B[:, :1], B[:, 1:], wq[0] * np.eye(2), wq[1] * np.eye(2), w[0][0] * I1, w[0][1] * I1,
                       w[1][0] * I1, w[1][1] * I1)
    assert np.max(np.abs(P0[0] - P1[0])) < 1e-4  # constants::kSmallNumber
    assert np.max(np.abs(P0[1] - P2[0])) < 1e-4


def _lq_costs(g, A, B, wq, w, nominal, x0, P, alpha, open_loop_eval, perturb=None):
    """Euler-free evaluation of the TRUE player costs of test_lq_solver's game under the strategy
    u_i = -P_i x - alpha_i (closed loop) or the recorded open-loop controls, optionally with a
    perturbation of one alpha entry — the check of NumericalCheckLocalNashEquilibrium
    (src/check_local_nash_equilibrium.cpp:60-133) specialised to this linear system."""
    T = g["T"]
    al = alpha.copy()
    if perturb is not None:
        k, row, eps = perturb
        al[k, row] += eps
    x = x0.copy()
    costs = np.zeros(2)
    xs, us = [], []
    for k in range(T):
        Pk = P[k].reshape(2, 2, order="F")
        u = -Pk @ x - al[k]
        xs.append(x.copy())
        us.append(u)
        x = A @ x + B @ u
    return xs, us


def _total_costs(xs, us, wq, w, nominal):
    c = np.zeros(2)
    for x, u in zip(xs, us):
        for i in range(2):
            c[i] += 0.5 * wq[i] * np.sum((x - nominal) ** 2)
            for j in range(2):
                c[i] += 0.5 * w[i][j] * (u[j] - nominal) ** 2
    return c


@pytest.mark.parametrize("nominal", [0.0, 0.5])
def test_feedback_solution_is_feedback_nash(oracle, nominal):
    """test/test_lq_solver.cpp:319-345: unilateral alpha perturbations (closed loop) never help."""
    g, A, B, wq, w = _pointmass(nominal)
    d = dims_of(g, abi.F64, adaptive=True)
    P, alpha, _, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    P, alpha = P[0], alpha[0]
    x0 = np.ones(2)
    xs, us = _lq_costs(g, A, B, wq, w, nominal, x0, P, alpha, False)
    base = _total_costs(xs, us, wq, w, nominal)
    for k in range(0, g["T"] - 1, 7):
        for i in range(2):
            for eps in (0.1, -0.1):
                xs2, us2 = _lq_costs(g, A, B, wq, w, nominal, x0, P, alpha, False, perturb=(k, i, eps))
                assert _total_costs(xs2, us2, wq, w, nominal)[i] >= base[i] - 1e-9


def test_openloop_solution_is_openloop_nash(oracle):
    """test/test_lq_solver.cpp:347-387: with the other player's control sequence frozen, a
    perturbation of one's own open-loop control never helps."""
    nominal = 0.5
    g, A, B, wq, w = _pointmass(nominal)
    d = dims_of(g, abi.F64)
    x0 = np.ones(2)
    P, alpha, dx, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0[None],
                                      open_loop=True)
    assert np.all(P == 0)
    T = g["T"]
    us = [-alpha[0, k] for k in range(T)]

    def roll(us):
        x = x0.copy()
        xs = []
        for k in range(T):
            xs.append(x.copy())
            x = A @ x + B @ us[k]
        return xs
    xs = roll(us)
    assert np.allclose(np.array(xs), dx[0], atol=1e-9)  # delta_xs is the optimal open-loop state
    base = _total_costs(xs, us, wq, w, nominal)
    for k in range(0, T - 1, 7):
        for i in range(2):
            for eps in (0.1, -0.1):
                us2 = [u.copy() for u in us]
                us2[k][i] += eps
                assert _total_costs(roll(us2), us2, wq, w, nominal)[i] >= base[i] - 1e-9


def test_single_player_openloop_equals_feedback_first_control(oracle):
    """test/test_lq_solver.cpp:389-434 (double integrator, B = 0.041 I, Q = I, R = I)."""
    T, dt = 100, 0.1
    A = np.eye(2)
    A[0, 1] = dt
    B = dt * 0.41 * np.eye(2)
    g = dict(n=2, ms=[2], T=T, N=1, pairs=[(0, 0)])
    g["A"] = np.tile(colmajor(A), (1, T, 1))
    g["Bm"] = np.tile(colmajor(B), (1, T, 1))
    g["Q"] = np.tile(colmajor(np.eye(2)), (1, T, 1, 1))
    g["l"] = np.zeros((1, T, 1, 2))
    g["R"] = np.tile(colmajor(np.eye(2)), (1, T, 1))
    g["r"] = np.zeros((1, T, 2))
    x0 = np.ones((1, 2))
    d = dims_of(g, abi.F64, adaptive=True)
    Pf, af, _, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0)
    Po, ao, _, _ = oracle.lq_solve(d, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"], x0=x0,
                                   open_loop=True)
    u_fb = -Pf[0, 0].reshape(2, 2, order="F") @ x0[0] - af[0, 0]
    u_ol = -ao[0, 0]
    assert np.max(np.abs(u_ol - u_fb)) < 0.01 * np.max(np.abs(u_fb))


def test_gershgorin_and_linear_terms_change_the_answer(oracle):
    """Guards the pieces the python oracle cannot pin (r_ij, adaptive regularisation): they must at
    least be live code paths with the documented effect (lq_feedback_solver.cpp:154-176)."""
    rng = np.random.default_rng(0)
    g = random_lq_game(rng, 6, [2, 2, 2], 12, 2)
    d0 = dims_of(g, abi.F64, adaptive=False)
    d1 = dims_of(g, abi.F64, adaptive=True)
    P0, a0, _, _ = oracle.lq_solve(d0, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    P1, a1, _, _ = oracle.lq_solve(d1, g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
    Pz, az, _, _ = oracle.lq_solve(d0, g["A"], g["Bm"], g["Q"], g["l"], g["R"], 0 * g["r"], g["pairs"])
    assert np.allclose(P0, Pz)            # r only moves alpha
    assert not np.allclose(a0, az)
    assert np.isfinite(P1).all() and np.isfinite(a1).all()


def test_missing_diagonal_block_is_an_error(oracle):
    """CHECK at lq_feedback_solver.cpp:139-140 -> ILQG_ERR_INVALID at the boundary."""
    rng = np.random.default_rng(0)
    g = random_lq_game(rng, 4, [2, 2], 5, 1, pairs=[(0, 0), (0, 1)])
    with pytest.raises(ValueError):
        oracle.lq_solve(dims_of(g, abi.F64), g["A"], g["Bm"], g["Q"], g["l"], g["R"], g["r"], g["pairs"])
