"""GPU parity of the equilibrium checks (ilqg_strategy_costs_batch, ilqg_check_local_nash_batch) against the oracle's
restatements of src/compute_strategy_costs.cpp:61-106 and src/check_local_nash_equilibrium.cpp:60-133, fp64.
Costs agree to accumulation error (1e-10 relative); the verdict is compared wherever the margin is not within
rounding of zero."""
import numpy as np
import pytest

from ilqgames_amd import abi, examples
from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from ilqgames_amd import hip as h
    return h


def _np(t):
    return t.detach().cpu().numpy()


def _solved(oracle, cfg, B, iters):
    spec = examples.CONFIGS[cfg]()
    spec.params.initial_alpha_scaling = 0.5 if "intersection" not in cfg or cfg.startswith("modified") else 0.1
    spec.params.expected_decrease_fraction = 0.001
    op = oracle.OracleProblem(spec)
    x0 = examples.jittered_x0(spec, B, seed=5)
    r = op.solve(abi.F64, x0, fixed_iters=iters)
    return spec, op, x0, r


@pytest.mark.parametrize("cfg", ["modified_three_player_intersection", "three_player_collision_avoidance_reachability",
                                 "two_player_unicycle_4d_scene", "roundabout_merging", "two_player_reachability",
                                 "one_player_reachability", "air_3d", "modified_air_3d",
                                 # the kinds no reference example uses (time-dependent costs, Car7D / Unicycle5D /
                                 # DelayedDubinsCar on the plain RK4, the norm / orientation / curvature costs)
                                 "cost_zoo_scene", "dynamics_zoo_scene", "delayed_dubins_scene", "weighted_proximity_scene"])
@pytest.mark.parametrize("open_loop,euler", [(False, True), (True, True), (False, False)])
def test_strategy_costs_match_oracle_fp64(hip, oracle, cfg, open_loop, euler):
    spec, op, x0, r = _solved(oracle, cfg, 3, 2)
    ref = op.strategy_costs(abi.F64, x0, r["xs"], r["us"], r["P"], r["alpha"], open_loop=open_loop, euler=euler)
    out = hip.Problem(spec, abi.F64).strategy_costs(x0, r["xs"], r["us"], r["P"], r["alpha"], open_loop=open_loop,
                                                    euler=euler)
    # A replay of an unstable closed loop can overflow (Air3D's Euler replay does), run away to 1e20+ (one instance of
    # dynamics_zoo_scene after two iterations) or merely amplify rounding by ten orders of magnitude (one of
    # cost_zoo_scene): nothing to compare there.  Which instances those are is measured — the oracle replays from x0
    # nudged by 1e-12 and must reproduce its own costs to 1e-9.
    nudged = op.strategy_costs(abi.F64, x0 + 1e-12 * np.random.default_rng(9).standard_normal(x0.shape), r["xs"], r["us"],
                               r["P"], r["alpha"], open_loop=open_loop, euler=euler)
    with np.errstate(invalid="ignore", over="ignore"):
        fin = np.isfinite(ref).all(axis=1) & np.isfinite(nudged).all(axis=1)
        fin &= np.nan_to_num(np.abs(nudged - ref) / np.maximum(np.abs(ref), 1e-30), nan=1.0).max(axis=1) < 1e-9
        # ... and a trajectory that has left the scene by ten orders of magnitude evaluates sines and cosines of
        # arguments ~1e11, where two correct libms already differ
        fin &= np.abs(np.nan_to_num(ref)).max(axis=1) < 1e12
    assert fin.any()
    assert rel_err(_np(out)[fin], ref[fin]) < 1e-10


@pytest.mark.parametrize("cfg,eps", [("modified_three_player_intersection", 1e-2), ("two_player_unicycle_4d_scene", 1e-2),
                                     ("three_player_collision_avoidance_reachability", 1e-1)])
@pytest.mark.parametrize("open_loop", [False, True])
def test_check_local_nash_matches_oracle_fp64(hip, oracle, cfg, eps, open_loop):
    """1 + 2 m (T-1) Euler rollouts per instance in one launch: same margins as the oracle's loop, same verdicts
    wherever the margin is not a rounding-level quantity.  Half of the instances carry the solver's strategies, the
    other half zero strategies (far from any equilibrium)."""
    B = 4
    spec, op, x0, r = _solved(oracle, cfg, B, 8)
    for k in ("us", "P", "alpha"):
        r[k][B // 2:] = 0.0
    r["xs"][B // 2:] = x0[B // 2:, None, :]
    ok_ref, mg_ref = op.check_local_nash(abi.F64, x0, r["xs"], r["us"], r["P"], r["alpha"], eps, open_loop=open_loop)
    ok, mg = hip.Problem(spec, abi.F64).check_local_nash(x0, r["xs"], r["us"], r["P"], r["alpha"], eps,
                                                        open_loop=open_loop)
    nominal = np.abs(op.strategy_costs(abi.F64, x0, r["xs"], r["us"], r["P"], r["alpha"], open_loop=open_loop)).max(axis=1)
    tol = 1e-9 * np.maximum(1.0, nominal)  # a margin is the difference of two costs of this size
    assert np.all(np.abs(_np(mg) - mg_ref) <= tol), (_np(mg), mg_ref)
    decided = np.abs(mg_ref) > 10 * tol
    assert decided.sum() >= B // 2
    assert np.array_equal(_np(ok)[decided], ok_ref[decided])


def test_check_local_nash_with_zero_perturbation_is_trivially_true(hip, oracle):
    spec, op, x0, r = _solved(oracle, "modified_three_player_intersection", 2, 3)
    ok, mg = hip.Problem(spec, abi.F64).check_local_nash(x0, r["xs"], r["us"], r["P"], r["alpha"], 0.0)
    assert np.all(_np(ok) == 1) and np.all(_np(mg) == 0.0)


def test_equilibrium_checks_fp32_track_the_fp32_oracle(hip, oracle):
    """fp32 (the reference's precision): costs to 1e-4, margins to 1e-3 of the nominal cost, verdicts where decided."""
    B = 4
    spec, op, x0, r = _solved(oracle, "modified_three_player_intersection", B, 6)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    args = [f32(x0)] + [f32(r[k]) for k in ("xs", "us", "P", "alpha")]
    prob = hip.Problem(spec, abi.F32)
    ref = op.strategy_costs(abi.F32, *args)
    assert rel_err(_np(prob.strategy_costs(*args)), ref) < 1e-4
    ok_ref, mg_ref = op.check_local_nash(abi.F32, *args, 5e-2)
    ok, mg = prob.check_local_nash(*args, 5e-2)
    tol = 1e-3 * np.abs(ref).max(axis=1)
    assert np.all(np.abs(_np(mg) - mg_ref) <= tol), (_np(mg), mg_ref)
    decided = np.abs(mg_ref) > 10 * tol
    assert np.array_equal(_np(ok)[decided], ok_ref[decided])


@pytest.mark.parametrize("open_loop", [False, True])
def test_strategy_costs_respect_final_time_costs_fp64(hip, oracle, open_loop):
    """FinalTimeCost terms (cost/final_time_cost.h:55-88) count from their threshold step on; in the open-loop form
    state costs are taken at the NEXT step's time (PlayerCost::EvaluateOffset), which moves the switch-on by one."""
    spec = examples.two_player_unicycle_4d_scene()
    spec.final_time(4.95, spec.quadratic(0, 40.0, 1, 3.0))
    spec.final_time(2.0, spec.quadratic(1, 7.0, 0, -2.0))
    op = oracle.OracleProblem(spec)
    x0 = examples.jittered_x0(spec, 3, seed=5)
    r = op.solve(abi.F64, x0, fixed_iters=2)
    ref = op.strategy_costs(abi.F64, x0, r["xs"], r["us"], r["P"], r["alpha"], open_loop=open_loop)
    out = hip.Problem(spec, abi.F64).strategy_costs(x0, r["xs"], r["us"], r["P"], r["alpha"], open_loop=open_loop)
    assert rel_err(_np(out), ref) < 1e-10
    plain = examples.two_player_unicycle_4d_scene()
    base = oracle.OracleProblem(plain).strategy_costs(abi.F64, x0, r["xs"], r["us"], r["P"], r["alpha"], open_loop=open_loop)
    assert np.all(ref > base)  # the gated terms did contribute


@pytest.mark.parametrize("cfg", ["modified_three_player_intersection", "three_player_intersection",
                                 "three_player_collision_avoidance_reachability", "two_player_reachability", "skeleton",
                                 "roundabout_merging"])
@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_check_sufficient_nash_matches_oracle(hip, oracle, cfg, dtype):
    """CheckSufficientLocalNashEquilibrium: the device decides positive semidefiniteness (to 1e-4) of every Q_i, R_ij
    by a shifted Cholesky factorisation, the oracle computes the smallest eigenvalue; same verdict wherever that
    eigenvalue is not within 1e-6 (fp32: 1e-5) of the -1e-4 margin.  Reachability players' signed-distance costs
    have indefinite Hessians, the intersection's costs are convex."""
    B = 6
    spec, op, x0, r = _solved(oracle, cfg, B, 1)
    ok_ref, worst = op.check_sufficient_nash(dtype, r["xs"], r["us"])
    ok = hip.Problem(spec, dtype).check_sufficient_nash(r["xs"], r["us"])
    decided = np.abs(worst + 1e-4) > (1e-6 if dtype == abi.F64 else 1e-5)
    assert decided.sum() >= B // 2
    assert np.array_equal(_np(ok)[decided], ok_ref[decided]), (_np(ok), ok_ref, worst)
