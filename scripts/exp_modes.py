"""Times one fixed-iteration batched solve under a given schedule (diagnostic; run under rocprofv3 for per-kernel times).

  python scripts/exp_modes.py --batch 8192 --dtype f64 --split 1 --iters 6
"""
import argparse
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402
from ilqgames_amd import abi, examples, hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--config", default="modified_three_player_intersection")
ap.add_argument("--split", type=int, default=-1, help="-1 auto, 0 fused, 1 split passes")
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--compact", type=int, default=-1, help="-1 auto, 0 dense rows, 1 compact rows")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dtype = abi.F64 if a.dtype == "f64" else abi.F32
spec = examples.CONFIGS[a.config]()
spec.params.initial_alpha_scaling = 0.1
spec.params.expected_decrease_fraction = 0.001
spec.params.max_backtracking_steps = 100
prob = hip.Problem(spec, dtype)
x0 = torch.as_tensor(examples.jittered_x0(spec, a.batch, seed=0), dtype=hip.torch_dtype(dtype), device="cuda")
bufs = prob.alloc_solve_buffers(a.batch)
split = None if a.split < 0 else bool(a.split)
compact = None if a.compact < 0 else bool(a.compact)


def run():
    for k in ("xs", "us", "P", "alpha"):
        bufs[k].zero_()
    prob.solve(x0, bufs, fixed_iters=a.iters, split_trial=split, compact_rows=compact)


run()
torch.cuda.synchronize()
ts = []
for _ in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in ("xs", "us", "P", "alpha"):
        bufs[k].zero_()
    e0.record()
    prob.solve(x0, bufs, fixed_iters=a.iters, split_trial=split, compact_rows=compact)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
ms = ts[len(ts) // 2]
print("batch %d %s split=%s compact=%s iters=%d: %.3f ms per solve, %.3f ms per iteration of the batch, %.3f M it/s" %
      (a.batch, a.dtype, a.split, a.compact, a.iters, ms, ms / a.iters, a.batch * a.iters / ms / 1e3))
