"""Is a configuration on compact rows?  Times a fixed-iteration solve with compact_rows left to the library / forced off."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from ilqgames_amd import abi, examples, hip
cfg = sys.argv[1] if len(sys.argv) > 1 else "three_player_intersection"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spec = examples.CONFIGS[cfg]()
spec.params.initial_alpha_scaling = 0.1; spec.params.expected_decrease_fraction = 0.001
x0 = torch.as_tensor(examples.jittered_x0(spec, B, seed=0), dtype=torch.float64, device="cuda")
for name, cr in (("library's choice", None), ("compact off", False)):
    prob = hip.Problem(spec, abi.F64)
    bufs = prob.alloc_solve_buffers(B)
    ts = []
    for rep in range(4):
        for k in ("xs", "us", "P", "alpha"): bufs[k].zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        prob.solve(x0, bufs, fixed_iters=6, compact_rows=cr)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%-18s %s B=%d: %.3f ms per 6 iterations" % (name, cfg, B, sorted(ts)[1] * 1e3))
