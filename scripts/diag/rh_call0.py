"""Diagnostic (GPU box): the first solver call of the three receding-horizon batch instances of tests/host/host_solve_demo
on the device against the oracle — outcome, final loop state, and what the oracle's own outcomes are under nudges of
x0.  python scripts/diag/rh_call0.py"""
import collections
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ilqgames_amd import abi, hip  # noqa: E402
from oracle import pyoracle  # noqa: E402
from test_host_mirror import _parse_rh, _as_host_floats  # noqa: E402

out = tempfile.mkdtemp()
subprocess.check_call([os.path.join(ROOT, "tests", "host", "_bin", "host_solve_demo"), out], stdout=subprocess.DEVNULL)
spec = abi.ProblemSpec.from_dump(open(os.path.join(out, "scene_rh.txt")).read())
O = pyoracle.OracleProblem(spec)
prob = hip.Problem(spec, abi.F64)
for b in range(3):
    logs = _parse_rh(os.path.join(out, "rh_batch_%d.txt" % b))
    x0 = _as_host_floats(logs[0]["xs"][0])
    print("== instance", b, "device (mirror) call 0: iters", logs[0]["iters"], "converged", logs[0]["converged"])
    for sw in (None,):
        bufs = prob.solve(x0[None, :], log_capacity=16)
        st = prob.solve_state(bufs)
        print("   device solve: iters", int(bufs["iters"][0]), "status", int(bufs["status"][0]), "converged",
              int(bufs["converged"][0]), "last_merit %.17g ED %.6g step %.6g backtracks %d" % (
                  float(st["last_merit"][0]), float(st["expected_decrease"][0]), float(st["step"][0]), int(st["backtracks"][0])))
    ref = O.solve(abi.F64, x0[None, :], merit_log_len=16)
    print("   oracle solve: iters", int(ref["iters"][0]), "status", int(ref["status"][0]), "converged", int(ref["converged"][0]))
    for q in range(16):
        if not np.isnan(ref["log"][0, q, 0]):
            print("      oracle it %d merit %.17g ED %.9g step %.6g bt %d" % ((q,) + tuple(ref["log"][0, q])))
    rng = np.random.default_rng(5)
    hist = collections.Counter()
    for scale in [1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8] * 40:
        r = O.solve(abi.F64, (x0 + scale * rng.standard_normal(x0.shape))[None, :], merit_log_len=16)
        hist[(int(r["iters"][0]), int(r["status"][0]), int(r["converged"][0]))] += 1
    print("   oracle outcomes (iters, status, converged) under 240 nudges:", dict(hist))
    # the device from nudged states as well
    hist = collections.Counter()
    xs = np.stack([x0 + scale * rng.standard_normal(x0.shape) for scale in [1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8] * 40])
    bufs = prob.solve(xs)
    for i in range(xs.shape[0]):
        hist[(int(bufs["iters"][i]), int(bufs["status"][i]), int(bufs["converged"][i]))] += 1
    print("   device outcomes under 240 nudges:", dict(hist))
