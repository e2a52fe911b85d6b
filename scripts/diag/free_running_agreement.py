"""Diagnostic (GPU box): how often a free-running fixed-iteration solve ends the same way (iterations, status) on the
device as on the oracle, against how often the oracle agrees with itself from x0 nudged by 1e-12 — on more instances
than tests/test_gpu_parity.py::test_ilq_solve_matches_oracle_fp64 uses.  python scripts/diag/free_running_agreement.py [cfg ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ilqgames_amd import abi, examples, hip  # noqa: E402
from oracle import pyoracle  # noqa: E402

cfgs = sys.argv[1:] or ["cost_zoo_scene", "skeleton", "dubins_origin", "three_player_intersection"]
for cfg in cfgs:
    spec = examples.CONFIGS[cfg]()
    spec.params.initial_alpha_scaling = 0.1
    spec.params.expected_decrease_fraction = 0.001
    B, K = 96, 6
    x0 = examples.jittered_x0(spec, B, seed=11)
    O = pyoracle.OracleProblem(spec)
    ref = O.solve(abi.F64, x0, fixed_iters=K, merit_log_len=K, threads=16)
    out = hip.Problem(spec, abi.F64).solve(x0, fixed_iters=K)
    it = out["iters"].cpu().numpy()
    st = out["status"].cpu().numpy()
    same = (st == ref["status"]) & (it == ref["iters"])
    rng = np.random.default_rng(5)
    stable = []
    for _ in range(4):
        nd = O.solve(abi.F64, x0 + 1e-12 * rng.standard_normal(x0.shape), fixed_iters=K, merit_log_len=K, threads=16)
        stable.append(np.mean((nd["status"] == ref["status"]) & (nd["iters"] == ref["iters"])))
    # where they end the same way, how close are the trajectories?
    xs = out["xs"].cpu().numpy()
    err = [np.max(np.abs(xs[b] - ref["xs"][b])) / max(1.0, np.max(np.abs(ref["xs"][b]))) for b in np.nonzero(same)[0]]
    print("%-46s device vs oracle agree %.3f | oracle vs nudged oracle %s | first 12: %.3f | median rel-err where equal %.2e" % (
        cfg, np.mean(same), " ".join("%.3f" % s for s in stable), np.mean(same[:12]), np.median(err) if err else float("nan")))
