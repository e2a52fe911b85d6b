import os, sys, subprocess, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from ilqgames_amd import abi, hip
from oracle import pyoracle
out = tempfile.mkdtemp()
subprocess.check_call([os.path.join(R, "tests", "host", "_bin", "host_solve_demo"), out], timeout=600)
spec = abi.ProblemSpec.from_dump(open(os.path.join(out, "scene_rh.txt")).read())
O = pyoracle.OracleProblem(spec)
for b in range(3):
    lines = open(os.path.join(out, "rh_batch_%d.txt" % b)).read().splitlines()
    x0 = None
    for ln in lines:
        if ln.startswith("x "):
            x0 = np.array([float(v) for v in ln.split()[1:]]); break
    print("instance", b, "device calls", lines[0])
    rng = np.random.default_rng(1)
    xs = [x0] + [x0 + s * rng.standard_normal(x0.shape) for s in (1e-12, 1e-12, 1e-12, 1e-13, 1e-11, 1e-10)]
    X = np.stack(xs)
    ref = O.solve(abi.F64, X, merit_log_len=12)
    print("  oracle (x0, then nudged): iters", ref["iters"], "status", ref["status"], "conv", ref["converged"])
    dev = hip.Problem(spec, abi.F64).solve(X)
    torch.cuda.synchronize()
    print("  device                  : iters", dev["iters"].cpu().numpy(), "status", dev["status"].cpu().numpy(), "conv", dev["converged"].cpu().numpy())
print("--- batch of the three x0 (as the demo) vs alone")
x0s = []
for b in range(3):
    for ln in open(os.path.join(out, "rh_batch_%d.txt" % b)).read().splitlines():
        if ln.startswith("x "):
            x0s.append(np.array([float(v) for v in ln.split()[1:]])); break
X3 = np.stack(x0s)
P = hip.Problem(spec, abi.F64)
d3 = P.solve(X3); torch.cuda.synchronize()
print("  batch of 3: iters", d3["iters"].cpu().numpy(), "status", d3["status"].cpu().numpy())
for b in range(3):
    d1 = hip.Problem(spec, abi.F64).solve(X3[b:b+1]); torch.cuda.synchronize()
    print("  alone", b, ": iters", d1["iters"].cpu().numpy(), "status", d1["status"].cpu().numpy(),
          "xs equal to batch:", bool(torch.equal(d1["xs"][0], d3["xs"][b])))
o3 = O.solve(abi.F64, X3)
print("  oracle    : iters", o3["iters"], "status", o3["status"])
import re
print("  demo first-call lines:", [open(os.path.join(out, "rh_batch_%d.txt" % b)).read().splitlines()[2][:60] for b in range(3)])
