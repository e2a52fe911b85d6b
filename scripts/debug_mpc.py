import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import pyoracle as oracle
from ilqgames_amd import abi, examples, hip
from helpers import rel_err
name = sys.argv[1] if len(sys.argv) > 1 else "three_player_intersection"
spec = examples.CONFIGS[name]()
spec.params.initial_alpha_scaling = 0.5
spec.params.expected_decrease_fraction = 0.01
spec.params.convergence_tolerance = 0.1
if name == "three_player_intersection":
    spec.params.max_solver_iters = 60
B = 8
x0 = examples.jittered_x0(spec, B, seed=3); x0[0] = spec.x0
op = oracle.OracleProblem(spec)
ref = op.receding_horizon_simulate(abi.F64, x0, 4.0, 0.25, max_records=16, threads=8)
prob = hip.Problem(spec, abi.F64)
recs = []
_np = lambda t: t.detach().cpu().numpy()
def on_record(r, info):
    recs.append(dict(first=None if info["first_step"] is None else _np(info["first_step"]).copy(), xs=_np(info["bufs"]["xs"]).copy(),
                     iters=_np(info["bufs"]["iters"]).copy(), status=_np(info["bufs"]["status"]).copy(), converged=_np(info["bufs"]["converged"]).copy(),
                     x0=_np(info["x0"]).copy(), active=_np(info["active"]).copy()))
out = prob.receding_horizon_simulate(x0, 4.0, 0.25, max_records=16, on_record=on_record)
for b in range(B):
    R = int(ref["num_records"][b])
    print("inst", b, "ref R", R, "dev R", int(_np(out["num_records"])[b]))
    for r in range(min(R, len(recs))):
        d = recs[r]
        print("  r", r, "iters", d["iters"][b], ref["iters"][b, r], "ok", d["status"][b], ref["ok"][b, r], "conv", d["converged"][b], ref["converged"][b, r],
              "first", None if d["first"] is None else d["first"][b], ref["first_step"][b, r], "x0err %.2e" % rel_err(d["x0"][b], ref["x0"][b, r]),
              "xserr %.2e" % rel_err(d["xs"][b], ref["xs"][b, r]), "act", d["active"][b])
