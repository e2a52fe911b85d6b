"""Prints per-stage HIP-vs-oracle errors for every config (diagnostic, not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ilqgames_amd import abi, examples, hip
from oracle import pyoracle
from helpers import rel_err
from test_gpu_parity import _random_op, _np

for cfg in examples.CONFIGS:
    for dtype in (abi.F64, abi.F32):
        spec = examples.CONFIGS[cfg]()
        rng = np.random.default_rng(7)
        B = 4
        x0, xs_ref, us_ref, P, alpha = _random_op(spec, rng, B)
        scale = np.array([1.0, 0.5, 0.25, 0.1])
        op = pyoracle.OracleProblem(spec)
        hp = hip.Problem(spec, dtype)
        xs_o, us_o = op.rollout(dtype, x0, xs_ref, us_ref, P, alpha, scale)
        xs_d, us_d = hp.rollout(x0, xs_ref, us_ref, P, alpha, scale)
        print(cfg, "f64" if dtype else "f32", "rollout xs %.2e us %.2e" % (rel_err(_np(xs_d), xs_o), rel_err(_np(us_d), us_o)))
        A_o, B_o = op.linearize(dtype, xs_o, us_o)
        A_d, B_d = hp.linearize(xs_o, us_o)
        print("   lin A %.2e B %.2e" % (rel_err(_np(A_d), A_o), rel_err(_np(B_d), B_o)))
        nc = spec.num_constraints
        lam = np.abs(rng.standard_normal((B, max(nc, 1), spec.T))) if nc else None
        mu = np.array([10.0, 11.0, 12.1, 5.0]) if nc else None
        te = rng.integers(0, spec.T, size=(B, len(spec.subsystems))).astype(np.int32)
        Q_o, l_o, R_o, r_o = op.quadraticize(dtype, xs_o, us_o, lam, mu, te)
        Q_d, l_d, R_d, r_d = hp.quadraticize(xs_o, us_o, lam, mu, te)
        for nm, a, b in (("Q", Q_d, Q_o), ("l", l_d, l_o), ("R", R_d, R_o), ("r", r_d, r_o)):
            a = _np(a)
            e = np.abs(a - b)
            idx = np.unravel_index(np.argmax(e), e.shape)
            print("   quad %s rel %.2e  max|d| %.3e at %s  (dev %.6g ref %.6g)" % (nm, rel_err(a, b), e.max(), idx, a[idx], b[idx]))
        c_o, te_o = op.total_costs(dtype, xs_o, us_o)
        c_d, te_d = hp.total_costs(xs_o, us_o)
        print("   costs %.2e te_equal %s" % (rel_err(_np(c_d), c_o), np.array_equal(_np(te_d), te_o)), _np(te_d).tolist(), te_o.tolist())
