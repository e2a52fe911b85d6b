"""Where do probe_lanes = ON / OFF solves differ (diagnostic)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
from ilqgames_amd import abi, examples, hip
cfg = sys.argv[1] if len(sys.argv) > 1 else "modified_three_player_intersection"
spec = examples.CONFIGS[cfg]()
spec.params.max_solver_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = 40
x0 = examples.jittered_x0(spec, B, seed=5)
outs = []
for kw in (dict(probe=True, probe_lanes=True), dict(probe=True, probe_lanes=False)):
    out = hip.Problem(spec, abi.F64).solve(x0, split_trial=True, **kw)
    torch.cuda.synchronize()
    outs.append({k: v.cpu().numpy().copy() for k, v in out.items() if hasattr(v, "shape") and k != "ws"})
a, b = outs
print("iters", a["iters"][:12], b["iters"][:12])
for k in ("xs", "us"):
    d = np.abs(a[k] - b[k])
    d = np.where(np.isnan(d), 0, d)
    print(k, "max diff", d.max(), "nan mismatch", int((np.isnan(a[k]) != np.isnan(b[k])).sum()))
    bad = np.argwhere(d > 0)
    if len(bad):
        inst = sorted(set(bad[:, 0]))
        print(" instances differing:", inst[:20])
        i = inst[0]
        bi = bad[bad[:, 0] == i]
        print(" first instance", i, "first step", bi[:, 1].min(), "columns", sorted(set(bi[bi[:, 1] == bi[:, 1].min()][:, 2])))
        k0 = bi[:, 1].min()
        print("  a:", a[k][i, k0], "\n  b:", b[k][i, k0])
