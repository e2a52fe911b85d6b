import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ilqgames_amd import abi, examples, hip
spec = examples.CONFIGS["modified_three_player_intersection"]()
B = 1024
prob = hip.Problem(spec, abi.F64)
x0 = examples.jittered_x0(spec, B, seed=0)
bufs = prob.alloc_solve_buffers(B)
prob.solve(torch.as_tensor(x0, dtype=torch.float64, device="cuda"), bufs, fixed_iters=2); torch.cuda.synchronize()
xs, us = bufs["xs"].clone(), bufs["us"].clone()
for _ in range(3):
    prob.quadraticize(xs, us)
    prob.total_costs(xs, us)
torch.cuda.synchronize()
