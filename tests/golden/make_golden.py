"""Generates tests/golden/*.npz by importing the reference's Python LQ solver.

Run ONLY in the build container (needs /root/reference):
    python tests/golden/make_golden.py
The reference Python never travels; only the vectors (inputs + expected outputs) are committed.

Fixtures:
  lq_feedback_random.npz   random 3-player game (n=5, m=(2,1,2), T=20), r_ij = 0, no
                           regularisation -> Ps/alphas of python/solve_lq_game.py:45-173
  lq_feedback_unicycle.npz BASELINE config 1: TwoPlayerUnicycle4D, T=50, (A,B) from the
                           reference's python/two_player_unicycle_4d.py linearize_discrete
                           along the zero-control nominal from x0=(0,-10,pi/4,5)
                           (src/two_player_reachability_example.cpp:63-66), Q_i = w_i I,
                           R_ii = I, R_ij = 0.1 I as test/test_lq_solver.cpp:227-248.
  lq_feedback_pointmass.npz test/test_lq_solver.cpp:143-186,227-264 fixture (nominal 0):
                           time-invariant TwoPlayerPointMass1D, T=100.
  product_dynamics_n14.npz BASELINE configs 2 / 3: the headline's own 14-state system.  The reference's
                           ProductMultiPlayerDynamicalSystem([Car5D(4.0, dt), Car5D(4.0, dt), Unicycle4D(dt)])
                           (python/product_multiplayer_dynamical_system.py, car_5d.py:50-88, unicycle_4d.py)
                           evaluated along two 100-step trajectories from the example's x0
                           (src/modified_three_player_intersection_example.cpp:108-126): zero controls, and
                           seeded random controls.  Stored per step: x, u, xdot = system(x, u) (the C++
                           ConcatenatedDynamicalSystem::Evaluate, src/concatenated_dynamical_system.cpp:69-84) and
                           (A, B_i) = linearize_discrete(x, u) = (I + dt J, dt dB_i) (the C++ ::Linearize, :86-107;
                           the MULTI-player class discretises by Euler like the C++ — the single-system
                           DynamicalSystem.linearize_discrete uses the zero-order hold and is not an oracle).
                           The states advance by x += dt * xdot of the reference's own __call__, so the fixture
                           holds nothing this repository computed.
Index mapping: solve_lq_game(As[0:T-1], Bs[i][0:T-1], Qs[i][k]=Q_i[k+1]... )
python's k-th Q is the state cost of time k+1 while the C++ sweep applies quad[k] at k and
quad[T-1] as terminal; with time-INVARIANT or explicitly shifted inputs both agree:
we pass Qs_py[i] = [Q_i[0], ..., Q_i[T-1]] and As_py = A[0:T-1], so python step k uses
(A[k], Q_i[k]) with terminal Z = Q_i[T-1] — exactly the C++ recursion.
"""
import os
import sys

import numpy as np

REF = "/root/reference/python"
sys.path.insert(0, REF)
from solve_lq_game import solve_lq_game  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run_reference(A, Bs, Q, l, R):
    """A [T][n,n]; Bs[i] [T][n,mi]; Q[i] [T][n,n]; l[i] [T][n]; R[i][j] [T][mj,mj] -> P[i] [T-1][mi,n], alpha[i] [T-1][mi]"""
    T = len(A)
    N = len(Bs)
    As = [A[k] for k in range(T - 1)]
    Bpy = [[Bs[i][k] for k in range(T - 1)] for i in range(N)]
    Qpy = [[Q[i][k] for k in range(T)] for i in range(N)]
    lpy = [[l[i][k].reshape(-1, 1) for k in range(T)] for i in range(N)]
    Rpy = [[[R[i][j][k] for k in range(T - 1)] for j in range(N)] for i in range(N)]
    # python loops k = len(As)-1 .. 0 indexing As[k], Bs[i][k], Qs[i][k], ls[i][k], Rs[i][j][k]
    # and seeds Z_i, zeta_i with Qs[i][-1], ls[i][-1]: with len(As) = T-1 and len(Qs[i]) = T
    # that is exactly the C++ recursion (terminal quad[T-1], stage k uses quad[k]).
    Ps, alphas = solve_lq_game(As, Bpy, Qpy, lpy, Rpy)
    return Ps, alphas


def save(name, A, Bs, Q, l, R, Ps, alphas, **extra):
    T = len(A)
    N = len(Bs)
    out = dict(A=np.stack(A), **extra)
    for i in range(N):
        out["B%d" % i] = np.stack(Bs[i])
        out["Q%d" % i] = np.stack(Q[i])
        out["l%d" % i] = np.stack(l[i])
        out["P%d" % i] = np.stack(Ps[i])
        out["alpha%d" % i] = np.stack([a.reshape(-1) for a in alphas[i]])
        for j in range(N):
            out["R%d%d" % (i, j)] = np.stack(R[i][j])
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, "T=%d N=%d" % (T, N))


def random_game():
    rng = np.random.default_rng(1234)
    n, ms, T = 5, (2, 1, 2), 20
    N = len(ms)
    A = [np.eye(n) + 0.1 * rng.standard_normal((n, n)) for _ in range(T)]
    Bs = [[0.3 * rng.standard_normal((n, ms[i])) for _ in range(T)] for i in range(N)]

    def spd(k, scale=1.0):
        M = rng.standard_normal((k, k))
        return scale * (M @ M.T / k + np.eye(k))
    Q = [[spd(n) for _ in range(T)] for i in range(N)]
    l = [[rng.standard_normal(n) for _ in range(T)] for i in range(N)]
    R = [[[spd(ms[j], 1.0 if i == j else 0.2) for _ in range(T)] for j in range(N)] for i in range(N)]
    Ps, alphas = run_reference(A, Bs, Q, l, R)
    save("lq_feedback_random.npz", A, Bs, Q, l, R, Ps, alphas)


def unicycle_game():
    from two_player_unicycle_4d import TwoPlayerUnicycle4D
    T, dt = 50, 0.1
    dyn = TwoPlayerUnicycle4D(T=dt)
    x = np.array([[0.0], [-10.0], [np.pi / 4], [5.0]])
    u0 = [np.zeros((2, 1)), np.zeros((2, 1))]
    A, B0, B1, xs = [], [], [], []
    for k in range(T):
        xs.append(x.copy())
        Ak, Bk = dyn.linearize_discrete(x, u0)
        A.append(np.array(Ak))
        B0.append(np.array(Bk[0]))
        B1.append(np.array(Bk[1]))
        x = dyn.integrate(x, u0)
    n = 4
    w = (1.0, 0.1)
    nominal = 0.5
    Q = [[w[i] * np.eye(n) for _ in range(T)] for i in range(2)]
    # QuadraticCost(w, -1, nominal) about the nominal trajectory: grad = w (x - nominal)
    l = [[w[i] * (xs[k].reshape(-1) - nominal) for k in range(T)] for i in range(2)]
    R = [[[(1.0 if i == j else 0.1) * np.eye(2) for _ in range(T)] for j in range(2)] for i in range(2)]
    Ps, alphas = run_reference(A, [B0, B1], Q, l, R)
    save("lq_feedback_unicycle.npz", A, [B0, B1], Q, l, R, Ps, alphas, xs=np.stack([v.reshape(-1) for v in xs]))


def pointmass_game():
    T, dt = 100, 0.1
    A1 = np.eye(2) + np.array([[0.0, 1.0], [0.0, 0.0]]) * dt
    B1 = np.array([[0.05], [1.0]]) * dt
    B2 = np.array([[0.032], [0.11]]) * dt
    A = [A1] * T
    Bs = [[B1] * T, [B2] * T]
    Q = [[1.0 * np.eye(2)] * T, [0.1 * np.eye(2)] * T]
    l = [[np.zeros(2)] * T, [np.zeros(2)] * T]
    R = [[[np.eye(1) * 1.0] * T, [np.eye(1) * 0.1] * T], [[np.eye(1) * 0.1] * T, [np.eye(1) * 1.0] * T]]
    Ps, alphas = run_reference(A, Bs, Q, l, R)
    save("lq_feedback_pointmass.npz", A, Bs, Q, l, R, Ps, alphas)


def product_dynamics():
    from car_5d import Car5D
    from unicycle_4d import Unicycle4D
    from product_multiplayer_dynamical_system import ProductMultiPlayerDynamicalSystem
    T, dt, L = 100, 0.1, 4.0
    system = ProductMultiPlayerDynamicalSystem([Car5D(L, dt), Car5D(L, dt), Unicycle4D(dt)], T=dt)
    # src/modified_three_player_intersection_example.cpp:108-126 (headings are float32 constants there)
    x0 = np.zeros((14, 1))
    x0[[0, 1, 2, 4], 0] = [-2.0, -30.0, np.float32(np.pi / 2), 4.0]
    x0[[5, 6, 7, 9], 0] = [-10.0, 45.0, np.float32(-np.pi / 2), 3.0]
    x0[[10, 11, 12, 13], 0] = [-11.0, 16.0, 0.0, 1.25]
    rng = np.random.default_rng(20261001)
    out = {}
    for name in ("zero", "random"):
        x = x0.copy()
        xs, us, xdots, As, Bs = [], [], [], [], [[], [], []]
        for k in range(T):
            if name == "zero":
                u = [np.zeros((2, 1)) for _ in range(3)]
            else:  # steering rate / acceleration (turn rate for the unicycle) of a size that bends the paths
                u = [rng.uniform(-1.0, 1.0, (2, 1)) * np.array([[0.4], [2.0]]) for _ in range(3)]
            xdot = system(x, u)
            A, B = system.linearize_discrete(x, u)
            xs.append(x.reshape(-1).copy())
            us.append(np.concatenate([v.reshape(-1) for v in u]))
            xdots.append(np.asarray(xdot).reshape(-1).copy())
            As.append(np.array(A))
            for i in range(3):
                Bs[i].append(np.array(B[i]))
            x = x + dt * xdot
        out["xs_" + name] = np.stack(xs)
        out["us_" + name] = np.stack(us)
        out["xdot_" + name] = np.stack(xdots)
        out["A_" + name] = np.stack(As)
        for i in range(3):
            out["B%d_%s" % (i, name)] = np.stack(Bs[i])
    out["dt"] = np.float64(dt)
    out["L"] = np.float64(L)
    np.savez_compressed(os.path.join(HERE, "product_dynamics_n14.npz"), **out)
    print("wrote product_dynamics_n14.npz", out["A_zero"].shape, out["B0_random"].shape)


if __name__ == "__main__":
    product_dynamics()
    random_game()
    unicycle_game()
    pointmass_game()
