"""Diagnostic (GPU box): where a solve with the static row stage first differs from the interpreter's.
python scripts/diag/static_vs_interp.py roundabout_merging [f64|f32]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ilqgames_amd import abi, examples, hip  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "roundabout_merging"
dtype = abi.F32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else abi.F64
spec = examples.CONFIGS[scene]()
B = 8
x0 = examples.jittered_x0(spec, B, seed=11)
prob = hip.Problem(spec, dtype)
print("static id", prob.row_program()[1])
for K in (1, 2, 3):
    for kw in (dict(), dict(probe=False), dict(split_trial=True)):
        a = prob.solve(x0, fixed_iters=K, static_rows=True, **kw)
        torch.cuda.synchronize()
        sched = prob.last_schedule()
        a = {q: a[q].cpu().numpy().copy() for q in ("xs", "us", "P", "alpha", "costs", "iters", "status")}
        b = prob.solve(x0, fixed_iters=K, static_rows=False, **kw)
        torch.cuda.synchronize()
        b = {q: b[q].cpu().numpy().copy() for q in a}
        sa = prob.solve_state(prob.solve(x0, fixed_iters=K, static_rows=True, **kw))
        sb = prob.solve_state(prob.solve(x0, fixed_iters=K, static_rows=False, **kw))
        print("K", K, kw, "schedule", sched, {q: (float(np.max(np.abs(a[q].astype(np.float64) - b[q]))) if a[q].size else 0.0) for q in a},
              "backtracks", sa["backtracks"].cpu().numpy().tolist(), sb["backtracks"].cpu().numpy().tolist(),
              "merit diff", float((sa["last_merit"] - sb["last_merit"]).abs().max()))
