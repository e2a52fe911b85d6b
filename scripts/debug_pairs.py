"""Split trial pass vs fused kernel (diagnostic): which schedule parts from the fused kernel, and in what."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from ilqgames_amd import abi, examples, hip
_np = lambda t: t.detach().cpu().numpy()
cfg, al = (sys.argv[1] if len(sys.argv) > 1 else "three_player_intersection"), (len(sys.argv) <= 3 or sys.argv[3] == "al")
spec = examples.CONFIGS[cfg]()
spec.params.max_solver_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
spec.params.unconstrained_solver_max_iters = int(sys.argv[4]) if len(sys.argv) > 4 else 4
B = 9
x0 = examples.jittered_x0(spec, B, seed=3)
if os.environ.get('ONLY'):
    x0 = x0[[int(v) for v in os.environ['ONLY'].split(',')]]
    B = x0.shape[0]
outs = []
for split, handoff, probe in ((False, False, True), (True, True, False)):
    out = hip.Problem(spec, abi.F64).solve(x0, augmented_lagrangian=al, split_trial=split, handoff=handoff, probe=probe)
    outs.append({k: _np(v).copy() for k, v in out.items() if hasattr(v, "shape") and k != "ws"})
for i, o in enumerate(outs[1:]):
    print("split/no probe: iters", o["iters"].tolist(), "status", o["status"].tolist())
    for k in outs[0]:
        a, b = outs[0][k].astype(np.float64), o[k].astype(np.float64)
        if not np.array_equal(a, b, equal_nan=True):
            d = np.abs(a - b).reshape(B, -1).max(axis=1)
            print("   ", k, "differs; per instance", d)
print("fused iters", outs[0]["iters"].tolist(), "status", outs[0]["status"].tolist())
