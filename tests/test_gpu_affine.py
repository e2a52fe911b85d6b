"""GPU parity of the reference's two dense constraints — AffineScalarConstraint, AffineVectorConstraint
(include/ilqgames/constraint/affine_scalar_constraint.h:54-100, affine_vector_constraint.h:52-112) — on the device
(ROP_AFFINE ops of the row program, csrc/ilqg_rows.hpp::rows_affine; multiplier update of the exit path).

The oracle's restatement of the two classes is pinned by the reference's own check of them (test/test_quadraticization.cpp:
305-316, re-expressed in tests/test_oracle_models.py: analytic derivatives against numerical ones of the augmented
Lagrangian); here the device is compared with the oracle: the quadraticisation stage at random operating points and
multipliers (both precisions; an inequality and an EQUALITY scalar constraint on the state, a vector constraint on a
control vector), whole solves after every forced-step iteration (specialised kernels and the run-time-dimensioned ones),
and AugmentedLagrangianSolver::Solve."""
import numpy as np
import pytest

from ilqgames_amd import abi, examples
from helpers import rel_err
from test_gpu_generic import _compare_forced, _forced, _np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from ilqgames_amd import hip as h
    return h


@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_affine_constraints_quadraticise_like_the_oracle(hip, oracle, dtype):
    spec = examples.affine_constraint_scene()
    B = 6
    rng = np.random.default_rng(3)
    prob, O = hip.Problem(spec, dtype), oracle.OracleProblem(spec)
    n, m, T, nc = spec.n, spec.m, spec.T, spec.num_constraints
    assert nc == 3
    x0 = examples.jittered_x0(spec, B, seed=1)
    xs, us = O.rollout(dtype, x0, np.zeros((B, T, n)), 0.3 * rng.standard_normal((B, T, m)),
                       0.01 * rng.standard_normal((B, T, m * n)), 0.1 * rng.standard_normal((B, T, m)))
    # multipliers of every sign for the equality constraint (slot 1), non-negative ones for the inequalities, some at
    # zero where the inactive-inequality gate of Constraint::Mu applies (constraint.h:112-117)
    lam = np.abs(rng.standard_normal((B, nc, T)))
    lam[:, 1] = rng.standard_normal((B, T))
    lam[:, 0, ::3] = 0.0
    mu = rng.uniform(5.0, 20.0, B)
    Q, l, R, r = prob.quadraticize(xs, us, lam, mu)
    Qr, lr, Rr, rr = O.quadraticize(dtype, xs, us, lam, mu)
    tol = 1e-9 if dtype == abi.F64 else 2e-3
    for got, want in ((Q, Qr), (l, lr), (R, Rr), (r, rr)):
        assert rel_err(_np(got), want) < tol
    # the dense blocks are really there: player 1's Q couples px1 with px2, player 1's R_11 is a full 2 x 2 block
    Q0 = Qr[0, 5, 0].reshape(n, n, order="F")
    assert abs(Q0[0, 5]) > 0 and abs(Rr[0, 5, 1]) > 0


@pytest.mark.parametrize("kwargs", [{}, dict(generic_kernels=True)], ids=["specialised", "generic"])
def test_solves_with_affine_constraints_match_oracle_after_every_forced_iteration(hip, oracle, kwargs):
    spec = examples.affine_constraint_scene()
    K, B = 4, 8
    x0, op, steps, x0n = _forced(oracle, spec, B, K, seed=51)
    _compare_forced(hip, op, spec, abi.F64, x0, steps, x0n, K, kwargs, min_cover=0.6)


def test_augmented_lagrangian_solve_with_affine_constraints_matches_oracle(hip, oracle):
    """AugmentedLagrangianSolver::Solve: multiplier updates of an inequality, an equality (not clipped at zero,
    constraint.h:98-102) and a vector constraint; instances whose decisions the oracle itself does not reproduce from a
    nudged x0 are left out."""
    spec = examples.affine_constraint_scene()
    spec.params.max_solver_iters = 20
    B = 8
    x0 = examples.jittered_x0(spec, B, seed=4)
    rng = np.random.default_rng(2)
    O = oracle.OracleProblem(spec)
    ref = O.solve(abi.F64, x0, augmented_lagrangian=True)
    nudged = [O.solve(abi.F64, x0 + e * rng.standard_normal(x0.shape), augmented_lagrangian=True) for e in (1e-12, 1e-9)]
    out = hip.Problem(spec, abi.F64).solve(x0, augmented_lagrangian=True)
    robust = [b for b in range(B) if all(ref["iters"][b] == r["iters"][b] and ref["status"][b] == r["status"][b] and
                                         rel_err(ref["xs"][b], r["xs"][b]) < 1e-6 for r in nudged)]
    assert len(robust) >= B // 2, robust
    for b in robust:
        assert _np(out["iters"])[b] == ref["iters"][b] and _np(out["status"])[b] == ref["status"][b], b
        assert rel_err(_np(out["xs"])[b], ref["xs"][b]) < 1e-6, b
        np.testing.assert_allclose(_np(out["costs"])[b], ref["costs"][b], rtol=1e-6)
