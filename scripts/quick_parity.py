"""Quick device-vs-oracle check of the n = 14 headline game (diagnostic; the parity tests proper are tests/test_gpu_*.py)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from ilqgames_amd import abi, examples, hip
from oracle import pyoracle
spec = examples.modified_three_player_intersection()
spec.params.expected_decrease_fraction = 0.001
spec.params.initial_alpha_scaling = 0.1
x0 = examples.jittered_x0(spec, 6, seed=3)
ok = True
for dtype, tol in ((abi.F64, 1e-9), (abi.F32, 2e-3)):
    for kw in (dict(fixed_iters=4), dict(fixed_iters=4, compact_rows=False), dict(fixed_iters=4, split_trial=True), dict()):
        out = hip.Problem(spec, dtype).solve(x0, **kw)
        torch.cuda.synchronize()
        okw = {k: v for k, v in kw.items() if k == "fixed_iters"}
        ref = pyoracle.OracleProblem(spec).solve(dtype, x0, **okw)
        e = {k: float(np.max(np.abs(out[k].cpu().numpy() - ref[k])) / max(np.max(np.abs(ref[k])), 1e-30)) for k in ("xs", "us", "P", "alpha", "costs")}
        same = np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
        good = same and all(v < tol for v in e.values()) if kw else same
        ok &= bool(good)
        print("dtype %d %s: iters equal %s  rel-err %s  %s" % (dtype, kw, same, " ".join("%s %.1e" % kv for kv in e.items()), "OK" if good else "FAIL"))
print("QUICK PARITY", "PASS" if ok else "FAIL")
