"""Shared helpers for the parity tests: fixture loading and flat-layout packing
(layouts of include/ilqg.h: [B][T][...] trajectory-major, column-major blocks)."""
import os

import numpy as np

from ilqgames_amd import abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def colmajor(M):
    """Flatten a 2-D block (or a stack [..., r, c]) column-major."""
    M = np.asarray(M)
    return np.swapaxes(M, -1, -2).reshape(M.shape[:-2] + (-1,))


def load_golden_lq(name, dtype=np.float64):
    """-> dict(dims-args, A, Bm, Q, l, R, r, pairs, P_ref, alpha_ref) with batch=1."""
    g = np.load(os.path.join(GOLDEN, name))
    A = g["A"]
    T, n, _ = A.shape
    N = 0
    while "B%d" % N in g:
        N += 1
    ms = [g["B%d" % i].shape[2] for i in range(N)]
    m = sum(ms)
    Bm = np.concatenate([g["B%d" % i] for i in range(N)], axis=2)  # [T][n][m]
    Q = np.stack([g["Q%d" % i] for i in range(N)], axis=1)          # [T][N][n][n]
    l = np.stack([g["l%d" % i] for i in range(N)], axis=1)          # [T][N][n]
    pairs = [(i, j) for i in range(N) for j in range(N)]
    R = np.concatenate([colmajor(g["R%d%d" % (i, j)]) for (i, j) in pairs], axis=1)  # [T][sum mj^2]
    r = np.zeros((T, sum(ms[j] for _, j in pairs)))
    P_ref = np.zeros((T, m, n))
    alpha_ref = np.zeros((T, m))
    off = 0
    for i in range(N):
        P_ref[:T - 1, off:off + ms[i], :] = g["P%d" % i]
        alpha_ref[:T - 1, off:off + ms[i]] = g["alpha%d" % i]
        off += ms[i]
    out = dict(n=n, ms=ms, T=T, N=N, pairs=pairs,
               A=colmajor(A)[None].astype(dtype), Bm=colmajor(Bm)[None].astype(dtype),
               Q=colmajor(Q)[None].astype(dtype), l=l[None].astype(dtype), R=R[None].astype(dtype),
               r=r[None].astype(dtype), P_ref=colmajor(P_ref)[None], alpha_ref=alpha_ref[None])
    return out


def random_lq_game(rng, n, ms, T, B, pairs=None, with_r=True, block_structured=False):
    """Synthetic LQ game in the flat layout, float64. Q, R_ii SPD; optional r_ij, l."""
    N = len(ms)
    m = sum(ms)
    if pairs is None:
        pairs = [(i, j) for i in range(N) for j in range(N)]

    def spd(k, scale):
        M = rng.standard_normal((B, T, k, k))
        return scale * (M @ np.swapaxes(M, -1, -2) / k + np.eye(k))
    A = np.eye(n) + 0.1 * rng.standard_normal((B, T, n, n))
    Bm = 0.3 * rng.standard_normal((B, T, n, m))
    Q = np.stack([spd(n, 1.0) for _ in range(N)], axis=2)
    l = rng.standard_normal((B, T, N, n))
    Rb, rb = [], []
    for (i, j) in pairs:
        Rb.append(colmajor(spd(ms[j], 1.0 if i == j else 0.2)))
        rb.append(rng.standard_normal((B, T, ms[j])) if with_r else np.zeros((B, T, ms[j])))
    return dict(n=n, ms=ms, T=T, N=N, pairs=pairs, A=colmajor(A), Bm=colmajor(Bm), Q=colmajor(Q), l=l,
                R=np.concatenate(Rb, axis=2), r=np.concatenate(rb, axis=2))


def dims_of(g, dtype, batch=None, adaptive=False):
    return abi.make_dims(g["n"], g["ms"], g["T"], batch if batch is not None else g["A"].shape[0], dtype, adaptive)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(b))))


def oracle_with_stability(op, dtype, x0, keys=("iters", "status"), nudge=1e-12, tol=1e-7, draws=1, **kw):
    """The oracle's result and the mask of instances whose OUTCOME is decided by the problem, not by rounding.

    A free-running solve (line search, convergence test, augmented-Lagrangian schedule) makes discrete decisions; where
    one of them hangs on the last bits of two large numbers, two correct implementations end differently — and so does
    the oracle against itself.  That is measured: the oracle is run again from x0 + nudge * noise, and an instance is
    `stable` when both runs end with the same `keys` and trajectories within `tol`.  Parity tests then demand agreement
    on the stable instances instead of on a tuned fraction of all of them.  `draws` > 1 repeats the measurement with
    other noise (the third draw at a tenth of `nudge`): one draw can miss an instance that sits next to a decision
    boundary on one side only."""
    ref = op.solve(dtype, x0, **kw)
    stable = np.ones(x0.shape[0], dtype=bool)
    for seed, scale in ((12345, 1.0), (5, 1.0), (777, 0.1))[:draws]:
        rng = np.random.default_rng(seed)
        again = op.solve(dtype, x0 + scale * nudge * rng.standard_normal(x0.shape), **kw)
        for k in keys:
            stable &= (np.asarray(ref[k]) == np.asarray(again[k]))
        for b in np.where(stable)[0]:
            stable[b] = rel_err(again["xs"][b], ref["xs"][b]) < tol
    return ref, stable
