import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqgames_amd import abi, examples, hip
from oracle import pyoracle as oracle
_np = lambda t: t.detach().cpu().numpy()
for cfg in ["modified_three_player_intersection", "three_player_intersection", "three_player_collision_avoidance_reachability", "two_player_unicycle_4d_scene", "two_player_reachability", "skeleton", "three_player_overtaking", "one_player_reachability", "dubins_origin", "air_3d", "modified_air_3d"]:
    spec = examples.CONFIGS[cfg]()
    spec.params.initial_alpha_scaling = 0.1 if cfg != "modified_three_player_intersection" else 0.5
    spec.params.expected_decrease_fraction = 0.001
    B, K = 12, (1 if ("reachability" in cfg or "overtaking" in cfg or cfg.endswith("air_3d")) else 6)
    x0 = examples.jittered_x0(spec, B, seed=11)
    ref = oracle.OracleProblem(spec).solve(abi.F64, x0, fixed_iters=K, merit_log_len=K)
    out = hip.Problem(spec, abi.F64).solve(x0, fixed_iters=K)
    same = (_np(out["status"]) == ref["status"]) & (_np(out["iters"]) == ref["iters"])
    bt = np.nan_to_num(ref["log"][:, :, 3], nan=0.0).max(axis=1)
    print(cfg, "agree %.2f" % same.mean(), "disagreeing depth:", bt[~same], "status ref", ref["status"][~same], "dev", _np(out["status"])[~same], "iters ref", ref["iters"][~same], "dev", _np(out["iters"])[~same])
