import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ilqgames_amd import abi, examples, hip
from oracle import pyoracle as oracle
import test_gpu_forced as tf
from helpers import rel_err
_np = lambda t: t.detach().cpu().numpy()
scene = "three_player_intersection"
spec = examples.CONFIGS[scene](); B = 12
rng = np.random.default_rng(100 + tf.SCENES.index(scene))
x0 = examples.jittered_x0(spec, B, seed=11)
op = oracle.OracleProblem(spec)
free = op.solve(abi.F64, x0, merit_log_len=tf.K)
print("free iters", free["iters"], "status", free["status"])
print("free steps\n", free["log"][:, :, 2])
steps = tf._forced_steps(rng, free["log"], float(spec.params.initial_alpha_scaling))
print("forced\n", steps)
prob = hip.Problem(spec, abi.F64)
for k in (2, 3):
    ref = op.solve(abi.F64, x0, fixed_iters=k, forced_steps=steps[:, :k], merit_log_len=k)
    out = prob.solve(x0, fixed_iters=k, forced_steps=steps[:, :k])
    for b in range(B):
        print(k, b, "xs err %.1e  max|xs| ref %.2e dev %.2e  merit ref %.3e" % (rel_err(_np(out["xs"])[b], ref["xs"][b]), np.max(np.abs(ref["xs"][b])), np.max(np.abs(_np(out["xs"])[b])), ref["log"][b, k-1, 0]))
