"""The bench line's `latency` leg alone (one instance, free-running to its convergence test), for traces:
   bash scripts/trace_cmd.sh lat python scripts/latency_run.py [f32]
   python scripts/trace_gaps.py gpurun_out/prof_lat/trace/run_results.db 2000"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ilqgames_amd import abi, examples, hip  # noqa: E402

spec = examples.modified_three_player_intersection()
spec.params.initial_alpha_scaling = 0.5
spec.params.expected_decrease_fraction = 0.001
spec.params.max_backtracking_steps = 100
dtype = abi.F32 if "f32" in sys.argv[1:] else abi.F64
prob = hip.Problem(spec, dtype)
x0 = examples.jittered_x0(spec, 1, seed=0)
bufs = prob.alloc_solve_buffers(1)
for rep in range(3):
    for k in ("xs", "us", "P", "alpha"):
        bufs[k].zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prob.solve(x0, bufs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it = int(bufs["iters"][0].item())
    print("solve %d: %.1f ms, %d iterations, %.4f ms per iteration" % (rep, dt * 1e3, it, dt * 1e3 / max(1, it)))
