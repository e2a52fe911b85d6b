"""Diagnostic: single-instance free-running solve latency with / without round bursts."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from ilqgames_amd import abi, examples, hip
spec = examples.CONFIGS["modified_three_player_intersection"]()
spec.params.initial_alpha_scaling = 0.5; spec.params.expected_decrease_fraction = 0.001; spec.params.max_backtracking_steps = 100
for dtype in (abi.F64, abi.F32):
    prob = hip.Problem(spec, dtype)
    x0 = torch.as_tensor(examples.jittered_x0(spec, 1, seed=0), dtype=hip.torch_dtype(dtype), device="cuda")
    for bursts in (True, False):
        lb = prob.alloc_solve_buffers(1)
        prob.solve(x0, lb, round_bursts=bursts)
        lat = []
        for _ in range(3):
            for k in ("xs", "us", "P", "alpha"): lb[k].zero_()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            prob.solve(x0, lb, round_bursts=bursts)
            torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
        it = int(lb["iters"][0].item())
        print("dtype %d bursts=%s: %.1f ms per solve, %d iterations, %.3f ms per iteration, converged %d" % (dtype, bursts, sorted(lat)[1] * 1e3, it, sorted(lat)[1] * 1e3 / it, int(lb["converged"][0].item())))
