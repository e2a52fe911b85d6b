import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ilqgames_amd import abi, examples, hip
from oracle import pyoracle as oracle
spec = examples.CONFIGS["roundabout_merging"]()
B, K = 12, 6
rng = np.random.default_rng(103)
x0 = examples.jittered_x0(spec, B, seed=11)
op = oracle.OracleProblem(spec)
free = op.solve(abi.F64, x0, merit_log_len=K)
acc = free["log"][:, :K, 2].astype(np.float64)
a0 = float(spec.params.initial_alpha_scaling)
acc = np.where(np.isfinite(acc) & (acc > 1e-6 * a0), acc, a0 / 256.0)
steps = acc * 0.5 ** rng.choice([0, 1, 2, 7, 9], p=[0.3, 0.3, 0.2, 0.1, 0.1], size=acc.shape)
def re(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
for dt in (abi.F32, abi.F64):
    prob = hip.Problem(spec, dt)
    for k in (1, 3, 6):
        r64 = op.solve(abi.F64, x0, fixed_iters=k, forced_steps=steps[:, :k])
        rdt = op.solve(dt, x0, fixed_iters=k, forced_steps=steps[:, :k])
        out = prob.solve(x0, fixed_iters=k, forced_steps=steps[:, :k])
        al = out["alpha"].cpu().numpy(); xs = out["xs"].cpu().numpy()
        e_dev = [re(al[b], r64["alpha"][b]) for b in range(B)]
        e_orc = [re(rdt["alpha"][b], r64["alpha"][b]) for b in range(B)]
        e_do = [re(al[b], rdt["alpha"][b]) for b in range(B)]
        print("dtype", dt, "k", k, "dev-vs-o64 max %.2e med %.2e | o(dt)-vs-o64 max %.2e med %.2e | dev-vs-o(dt) max %.2e" % (
            max(e_dev), np.median(e_dev), max(e_orc), np.median(e_orc), max(e_do)))
