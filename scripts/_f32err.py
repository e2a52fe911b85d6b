import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ilqgames_amd import abi, examples, hip
from oracle import pyoracle as oracle
import test_gpu_forced as tf
from helpers import rel_err
_np = lambda t: t.detach().cpu().numpy()
for scene in tf.SCENES:
    spec = examples.CONFIGS[scene](); B = 12
    rng = np.random.default_rng(100 + tf.SCENES.index(scene))
    x0 = examples.jittered_x0(spec, B, seed=11)
    op = oracle.OracleProblem(spec)
    free = op.solve(abi.F64, x0, merit_log_len=tf.K)
    steps = tf._forced_steps(rng, free["log"], float(spec.params.initial_alpha_scaling))
    for dtype in (abi.F64, abi.F32):
        prob = hip.Problem(spec, dtype)
        for k in range(1, tf.K + 1):
            ref = op.solve(dtype, x0, fixed_iters=k, forced_steps=steps[:, :k], merit_log_len=k)
            out = prob.solve(x0, fixed_iters=k, forced_steps=steps[:, :k])
            e = [max(rel_err(_np(out[a])[b], ref[r][b]) for b in range(B)) for a, r in (("xs","xs"),("us","us"),("P","rawP"),("alpha","alpha"))]
            print(scene[:28], "f64" if dtype == abi.F64 else "f32", "k=%d" % k, "xs %.1e us %.1e P %.1e alpha %.1e  max|xs| %.1e" % (*e, np.max(np.abs(ref["xs"]))))
