"""Quick device-vs-oracle check of the n = 24 open-loop game (config 4's scene), compact rows on / off (diagnostic)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from ilqgames_amd import abi, examples, hip
from oracle import pyoracle
spec = examples.roundabout_merging()
x0 = examples.jittered_x0(spec, 5, seed=3)
ok = True
for dtype, tol in ((abi.F64, 1e-8), (abi.F32, 5e-3)):
    outs = {}
    for kw in (dict(fixed_iters=3), dict(fixed_iters=3, compact_rows=False)):
        out = hip.Problem(spec, dtype).solve(x0, **kw)
        torch.cuda.synchronize()
        ref = pyoracle.OracleProblem(spec).solve(dtype, x0, fixed_iters=3)
        e = {k: float(np.max(np.abs(out[k].cpu().numpy() - ref[k])) / max(np.max(np.abs(ref[k])), 1e-30)) for k in ("xs", "us", "alpha", "costs")}
        same = np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
        good = same and all(v < tol for v in e.values())
        ok &= bool(good)
        outs[str(kw)] = out
        print("dtype %d %s: iters equal %s rel-err %s %s" % (dtype, kw, same, " ".join("%s %.1e" % kv for kv in e.items()), "OK" if good else "FAIL"))
    a, b = list(outs.values())
    same_bits = all(torch.equal(a[k], b[k]) for k in ("xs", "us", "alpha", "P"))
    print("  compact vs dense bit-identical:", same_bits)
    ok &= same_bits
print("QUICK PARITY OL", "PASS" if ok else "FAIL")
