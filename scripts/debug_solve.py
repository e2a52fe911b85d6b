"""Prints HIP-vs-oracle differences of the whole iLQ solve (diagnostic, not a test)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from ilqgames_amd import abi, examples, hip
from oracle import pyoracle
from helpers import rel_err
_np = lambda t: t.detach().cpu().numpy()
for cfg in ["modified_three_player_intersection", "three_player_intersection", "three_player_collision_avoidance_reachability"]:
    for K in (1, 2, 3, 6):
        spec = examples.CONFIGS[cfg]()
        spec.params.initial_alpha_scaling = 0.1 if cfg != "modified_three_player_intersection" else 0.5
        spec.params.expected_decrease_fraction = 0.001
        B = 6
        x0 = examples.jittered_x0(spec, B, seed=11)
        ref = pyoracle.OracleProblem(spec).solve(abi.F64, x0, fixed_iters=K, merit_log_len=8)
        out = hip.Problem(spec, abi.F64).solve(x0, fixed_iters=K)
        print(cfg, "K", K, "iters", _np(out["iters"]), ref["iters"], "status", _np(out["status"]), ref["status"])
        print("   xs %.2e us %.2e P %.2e alpha %.2e costs %.2e" % tuple(rel_err(_np(out[k]), ref[k]) for k in ("xs", "us", "P", "alpha", "costs")))
        print("   bt ref", ref["log"][:, :K, 3].tolist())
spec = examples.modified_three_player_intersection()
x0 = examples.jittered_x0(spec, 8, seed=3)
ref = pyoracle.OracleProblem(spec).solve(abi.F64, x0, merit_log_len=8)
out = hip.Problem(spec, abi.F64).solve(x0)
print("free", _np(out["iters"]), ref["iters"], _np(out["status"]), ref["status"], _np(out["converged"]), ref["converged"])
print("   xs %.2e costs %.2e" % (rel_err(_np(out["xs"]), ref["xs"]), rel_err(_np(out["costs"]), ref["costs"])))
print(ref["log"][:, :3, :].tolist())
