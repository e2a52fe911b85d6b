"""Single-instance free-running solve (bench.py's latency figure) — run under rocprofv3 to see kernel times and gaps."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from ilqgames_amd import abi, examples, hip
spec = examples.modified_three_player_intersection()
spec.params.initial_alpha_scaling = 0.5; spec.params.expected_decrease_fraction = 0.001
spec.params.convergence_tolerance = 1.0; spec.params.max_backtracking_steps = 100
prob = hip.Problem(spec, abi.F64)
x0 = torch.as_tensor(examples.jittered_x0(spec, 1, seed=0), dtype=torch.float64, device="cuda")
lb = prob.alloc_solve_buffers(1)
prob.solve(x0, lb)
for _ in range(2):
    for k in ("xs", "us", "P", "alpha"): lb[k].zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prob.solve(x0, lb); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it = int(lb["iters"][0].item())
    print("ms per solve %.2f, iterations %d, ms per iteration %.4f" % (dt * 1e3, it, dt * 1e3 / max(it, 1)))
