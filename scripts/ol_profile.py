"""Phase profile of the open-loop sweep inside the solve (needs the ILQG_PROFILE=1 build)."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from ilqgames_amd import abi, examples, hip
cfg = sys.argv[1] if len(sys.argv) > 1 else "roundabout_merging_T150"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 768
spec = examples.CONFIGS[cfg]()
spec.params.initial_alpha_scaling = 0.1; spec.params.expected_decrease_fraction = 0.001; spec.params.max_backtracking_steps = 100
prob = hip.Problem(spec, abi.F64)
x0 = torch.as_tensor(examples.jittered_x0(spec, B, seed=0), dtype=torch.float64, device="cuda")
bufs = prob.alloc_solve_buffers(B)
prob.solve(x0, bufs, fixed_iters=2); torch.cuda.synchronize()
prof = torch.zeros((B, 96), dtype=torch.int64, device="cuda")
hip.lib().ilqg_debug_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
for k in ("xs", "us", "P", "alpha"): bufs[k].zero_()
K = 3
prob.solve(x0, bufs, fixed_iters=K); torch.cuda.synchronize()
hip.lib().ilqg_debug_set_profile_buffer(None)
pm = prof.double().mean(0).cpu().numpy()
steps = K * (spec.T - 1)
q = pm[8:16] / steps
print("lq kernel cycles/launch %.0f ; per step (wave 0): V, g, [K | V A] to barrier 1 %.0f | elimination %.0f | barrier 2 %.0f | Xa, W, Ma' %.0f | transpose, DMA issue %.0f" %
      (pm[2] / K, q[0], q[1], q[2], q[3], q[4]))
print("trial kernel cycles/launch %.0f" % (pm[1] / (K + 1)))
