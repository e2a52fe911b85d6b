"""Launch time of the stand-alone feedback sweep (ilqg_lq_feedback_batch) per shape (diagnostic): python scripts/diag/lq_time.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from ilqgames_amd import abi, hip
from helpers import random_lq_game, dims_of
rng = np.random.default_rng(0)
for (n, ms, T) in ((8, [2, 2], 100), (8, [1, 2], 100), (14, [2, 2, 2], 40), (12, [2, 2, 2], 40), (14, [2, 2, 2], 100)):
    g = random_lq_game(rng, n, ms, T, 64)
    B = 1024
    rep = lambda a: torch.as_tensor(np.ascontiguousarray(np.tile(a, (B // 64,) + (1,) * (a.ndim - 1))), device="cuda")
    arrs = [rep(g[k]) for k in ("A", "Bm", "Q", "l", "R", "r")]
    d = dims_of(g, abi.F64, batch=B, adaptive=True)
    for want_dx in (False, True):
        for _ in range(2):
            hip.lq_feedback(d, *arrs, g["pairs"], want_dx=want_dx)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            hip.lq_feedback(d, *arrs, g["pairs"], want_dx=want_dx)
        torch.cuda.synchronize()
        print(n, ms, T, "dx" if want_dx else "no-dx", "%.0f us" % ((time.perf_counter() - t0) / 5 * 1e6))
