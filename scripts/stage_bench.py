"""Times each stage kernel and the fused solve (diagnostic)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from ilqgames_amd import abi, examples, hip
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--dtype", default="f64")
ap.add_argument("--config", default="modified_three_player_intersection")
a = ap.parse_args()
dtype = abi.F64 if a.dtype == "f64" else abi.F32
spec = examples.CONFIGS[a.config]()
spec.params.initial_alpha_scaling = 0.1; spec.params.expected_decrease_fraction = 0.001; spec.params.max_backtracking_steps = 100
B = a.batch
prob = hip.Problem(spec, dtype)
x0 = examples.jittered_x0(spec, B, seed=0)
td = hip.torch_dtype(dtype)
x0d = torch.as_tensor(x0, dtype=td, device="cuda")
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
bufs = prob.alloc_solve_buffers(B)
prob.solve(x0d, bufs, fixed_iters=3); torch.cuda.synchronize()
xs, us, P, al = bufs["xs"].clone(), bufs["us"].clone(), bufs["P"].clone(), bufs["alpha"].clone()
print("rollout    %.3f ms" % timeit(lambda: prob.rollout(x0d, xs, us, P, al)))
print("linearize  %.3f ms" % timeit(lambda: prob.linearize(xs, us)))
print("quadratize %.3f ms" % timeit(lambda: prob.quadraticize(xs, us)))
print("totalcosts %.3f ms" % timeit(lambda: prob.total_costs(xs, us)))
A, Bm = prob.linearize(xs, us); Q, l, Rr, r = prob.quadraticize(xs, us)
d = abi.make_dims(prob.n, spec.udims, prob.T, B, dtype, True)
print("lq_feedback (dx) %.3f ms" % timeit(lambda: hip.lq_feedback(d, A, Bm, Q, l, Rr, r, prob.pairs)))
print("lq_feedback (no dx) %.3f ms" % timeit(lambda: hip.lq_feedback(d, A, Bm, Q, l, Rr, r, prob.pairs, want_dx=False)))
for K in (1, 2, 4, 8):
    def f():
        for k in ("xs", "us", "P", "alpha"): bufs[k].zero_()
        prob.solve(x0d, bufs, fixed_iters=K)
    print("fused solve K=%d  %.3f ms" % (K, timeit(f, reps=2)))
prof = torch.zeros((B, 96), dtype=torch.int64, device="cuda")
import ctypes
hip.lib().ilqg_debug_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
for k in ("xs", "us", "P", "alpha"): bufs[k].zero_()
prob.solve(x0d, bufs, fixed_iters=4); torch.cuda.synchronize()
pm = prof.double().mean(0).cpu().numpy()
print("solve K=4 mean cycles/instance: trial kernel %.0f (5 launches)  lq kernel %.0f (5 launches, 4 sweeps)" % (pm[1], pm[2]))
print("  per launch: trial %.0f  lq sweep %.0f" % (pm[1] / 5, pm[2] / 4))
hip.lib().ilqg_debug_set_profile_buffer(None)

for wv in range(3):
    q = pm[8 + 16 * wv: 8 + 16 * wv + 16] / 396
    print("  wave %d (cycles/step): issue+ql %.0f | G,SY %.0f | bar1 %.0f | solve %.0f | bar2 %.0f | F,beta %.0f | players %.0f | zeta %.0f | dmawait %.0f | bar3 %.0f   sum %.0f" %
          (wv, q[0], q[1], q[7], q[2], q[8], q[3], q[4], q[5], q[9], q[6], q.sum()))
    print("     detail: G + LDS bounce %.0f | [S|Y] rows %.0f | y_zeta %.0f || solve: load + Gershgorin %.0f | elimination %.0f | stores %.0f || helpers: stage %.0f | stash Q l %.0f" % (q[12], q[13], q[1], q[10], q[11], q[2], q[14], q[15]))

for wv in range(2):
    q = pm[64 + 8 * wv: 72 + 8 * wv]
    chunks = max(q[7], 1.0)
    print("  trial wave %d: %.0f row chunks over 5 launches; cycles/chunk: staging+init %.0f | jacobians %.0f | cost terms %.0f | write-out %.0f | merit %.0f | claim/wait %.0f" %
          (wv, q[7], q[0] / chunks, q[1] / chunks, q[2] / chunks, q[3] / chunks, q[4] / chunks, q[6] / chunks))
q = pm[88:92] / 500.0
print("  rollout (cycles/step, 5 rollouts x 100 steps): publish+dx %.0f | u = u_ref - P dx - alpha %.0f | RK4 %.0f | commit prefetch %.0f" % tuple(q))
