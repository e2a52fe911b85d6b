"""Descriptor builders for the reference example Problems used by BASELINE.json's configs.

Each function restates the constants and the cost wiring of one reference
`Problem` subclass (file:line cited) as an `abi.ProblemSpec`; the reference's
quirks are reproduced, not fixed (e.g. roundabout players 2-4 penalising P1's
acceleration index).  x0 is the reference's ConstructInitialState.
"""
import math

import numpy as np

from . import abi
from .abi import (DYN_CAR_5D, DYN_CAR_6D, DYN_DUBINS_CAR, DYN_PLANAR_DISTURBANCE, DYN_UNICYCLE_4D,
                  DYN_UNICYCLE_4D_DISTURBED,
                  ProblemSpec, SolverParams)


def _lane_costs(spec, player, lane, xy, lane_w, boundary_w, half_width):
    # QuadraticPolyline2Cost + right/left SemiquadraticPolyline2Cost, e.g.
    # src/modified_three_player_intersection_example.cpp:196-209
    spec.quadratic_polyline2(player, lane_w, lane, xy)
    spec.semiquadratic_polyline2(player, boundary_w, lane, xy, half_width, True)
    spec.semiquadratic_polyline2(player, boundary_w, lane, xy, -half_width, False)


def modified_three_player_intersection(T=100, dt=0.1):
    """ModifiedThreePlayerIntersectionExample — n=14 (Car5D, Car5D, Unicycle4D), unconstrained.
    src/modified_three_player_intersection_example.cpp:76-329; params from
    exec/modified_three_player_intersection_example/main.cpp:74-76,110-116."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 1.0
    prm.convergence_tolerance = 1.0
    prm.expected_decrease_fraction = 0.9
    s = ProblemSpec(T, dt, prm)
    L = 4.0
    for kind in (DYN_CAR_5D, DYN_CAR_5D, DYN_UNICYCLE_4D):
        s.add_player(kind, L, state_reg=10.0, control_reg=10.0)
    P1X, P1Y, P1H, P1PHI, P1V = 0, 1, 2, 3, 4
    P2X, P2Y, P2H, P2PHI, P2V = 5, 6, 7, 8, 9
    P3X, P3Y, P3H, P3V = 10, 11, 12, 13
    p1x0, p2x0, p3x0 = -2.0, -10.0, -11.0
    p1y0, p2y0, p3y0 = -30.0, 45.0, 16.0
    lane1 = s.add_polyline([(p1x0, -1000.0), (p1x0, 1000.0)])
    lane2 = s.add_polyline([(p2x0, 1000.0), (p2x0, 28.0), (p2x0 + 0.5, 25.0), (p2x0 + 1.0, 24.0),
                            (p2x0 + 3.0, 22.5), (p2x0 + 6.0, 22.0), (1000.0, 22.0)])
    lane3 = s.add_polyline([(-1000.0, p3y0), (1000.0, p3y0)])
    _lane_costs(s, 0, lane1, (P1X, P1Y), 25.0, 100.0, 2.5)
    _lane_costs(s, 1, lane2, (P2X, P2Y), 25.0, 100.0, 2.5)
    _lane_costs(s, 2, lane3, (P3X, P3Y), 25.0, 100.0, 2.5)
    for pl, vidx, vmax, vnom in ((0, P1V, 12.0, 8.0), (1, P2V, 12.0, 6.0), (2, P3V, 2.0, 1.5)):
        s.semiquadratic(pl, 100.0, vidx, 1.0, False)   # MinV
        s.semiquadratic(pl, 100.0, vidx, vmax, True)   # MaxV
        s.quadratic(pl, 10.0, vidx, vnom)              # NominalV
    for pl in range(3):
        s.quadratic(pl, 0.1, 0, 0.0, control_of=pl)
        s.quadratic(pl, 0.1, 1, 0.0, control_of=pl)
    w = 0.0  # kProximityCostWeight = 0.0 (:89)
    s.proximity(0, w, (P1X, P1Y), (P2X, P2Y), 6.0)
    s.proximity(0, w, (P1X, P1Y), (P3X, P3Y), 6.0)
    s.proximity(1, w, (P2X, P2Y), (P1X, P1Y), 6.0)
    s.proximity(1, w, (P2X, P2Y), (P3X, P3Y), 6.0)
    s.proximity(2, w, (P3X, P3Y), (P1X, P1Y), 6.0)
    s.proximity(2, w, (P3X, P3Y), (P2X, P2Y), 6.0)
    x0 = np.zeros(14)
    x0[[P1X, P1Y, P1H, P1V]] = [p1x0, p1y0, np.float32(math.pi / 2), 4.0]
    x0[[P2X, P2Y, P2H, P2V]] = [p2x0, p2y0, np.float32(-math.pi / 2), 3.0]
    x0[[P3X, P3Y, P3H, P3V]] = [p3x0, p3y0, 0.0, 1.25]
    s.x0 = x0
    s.position_dims = [(P1X, P1Y), (P2X, P2Y), (P3X, P3Y)]
    s.heading_dims = [P1H, P2H, P3H]
    s.speed_dims = [P1V, P2V, P3V]
    return s


def three_player_intersection(T=100, dt=0.1):
    """ThreePlayerIntersectionExample — n=16 (Car6D, Car6D, Unicycle4D) with six
    ProximityConstraints.  src/three_player_intersection_example.cpp:78-394;
    params from exec/three_player_intersection/main.cpp:109-120."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.max_solver_iters = 100
    prm.unconstrained_solver_max_iters = 10
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 1.0
    prm.expected_decrease_fraction = 0.001
    s = ProblemSpec(T, dt, prm)
    L = 4.0
    for kind in (DYN_CAR_6D, DYN_CAR_6D, DYN_UNICYCLE_4D):
        s.add_player(kind, L, state_reg=1.0, control_reg=5.0)
    P1X, P1Y, P1H, P1V = 0, 1, 2, 4
    P2X, P2Y, P2H, P2V = 6, 7, 8, 10
    P3X, P3Y, P3H, P3V = 12, 13, 14, 15
    p1x0, p2x0, p3x0 = -2.0, -10.0, -11.0
    p1y0, p2y0, p3y0 = -30.0, 45.0, 16.0
    lane1 = s.add_polyline([(p1x0, -1000.0), (p1x0, 1000.0)])
    lane2 = s.add_polyline([(p2x0, 1000.0), (p2x0, 18.0), (p2x0 + 0.5, 15.0), (p2x0 + 1.0, 14.0),
                            (p2x0 + 3.0, 12.5), (p2x0 + 6.0, 12.0), (1000.0, 12.0)])
    lane3 = s.add_polyline([(-1000.0, p3y0), (1000.0, p3y0)])
    s.quadratic_polyline2(0, 25.0, lane1, (P1X, P1Y))
    s.quadratic_polyline2(1, 25.0, lane2, (P2X, P2Y))
    s.quadratic_polyline2(2, 25.0, lane3, (P3X, P3Y))
    s.quadratic(0, 100.0, P1V, 8.0)
    s.quadratic(1, 100.0, P2V, 5.0)
    s.quadratic(2, 100.0, P3V, 1.5)
    for pl in range(3):
        s.quadratic(pl, 0.1, 0, 0.0, control_of=pl)
        s.quadratic(pl, 0.1, 1, 0.0, control_of=pl)
    keep_close = True
    s.proximity_constraint(0, (P1X, P1Y), (P2X, P2Y), 6.0, not keep_close)
    s.proximity_constraint(0, (P1X, P1Y), (P3X, P3Y), 6.0, not keep_close)
    s.proximity_constraint(1, (P2X, P2Y), (P1X, P1Y), 6.0, not keep_close)
    s.proximity_constraint(1, (P2X, P2Y), (P3X, P3Y), 6.0, not keep_close)
    s.proximity_constraint(2, (P3X, P3Y), (P1X, P1Y), 6.0, not keep_close)
    s.proximity_constraint(2, (P3X, P3Y), (P2X, P2Y), 6.0, not keep_close)
    x0 = np.zeros(16)
    x0[[P1X, P1Y, P1H, P1V]] = [p1x0, p1y0, np.float32(math.pi / 2), 4.0]
    x0[[P2X, P2Y, P2H, P2V]] = [p2x0, p2y0, np.float32(-math.pi / 2), 3.0]
    x0[[P3X, P3Y, P3H, P3V]] = [p3x0, p3y0, 0.0, 1.25]
    s.x0 = x0
    s.position_dims = [(P1X, P1Y), (P2X, P2Y), (P3X, P3Y)]
    s.heading_dims = [P1H, P2H, P3H]
    s.speed_dims = [P1V, P2V, P3V]
    return s


def roundabout_lane_center(entrance_angle, exit_angle, distance_from_roundabout):
    """RoundaboutLaneCenter, src/roundabout_lane_center.cpp:50-106 (fp32 arithmetic)."""
    f = np.float32
    R, hw = f(12.0), f(2.5)
    ea = f(entrance_angle)
    xa = f(exit_angle)
    cx, cy = (R + hw) * f(math.cos(ea)), (R + hw) * f(math.sin(ea))
    a0 = f(float(ea) - math.pi / 2)  # double subtraction, rounded once
    fx, fy = cx + hw * f(math.cos(a0)), cy + hw * f(math.sin(a0))
    d = f(distance_from_roundabout)
    pts = [(fx + d * f(math.cos(ea)), fy + d * f(math.sin(ea))), (fx, fy)]
    for ii in range(1, 4):
        ang = f(float(a0) - math.pi / 2 * float(ii) / 3.0)
        pts.append((cx + hw * f(math.cos(ang)), cy + hw * f(math.sin(ang))))
    for ii in range(1, 11):
        na = f(ea + (xa - ea) * f(ii) / f(10))
        pts.append((R * f(math.cos(na)), R * f(math.sin(na))))
    far = f(1e4)
    pts.append((far * f(math.cos(xa)), far * f(math.sin(xa))))
    return [(float(x), float(y)) for x, y in pts]


def roundabout_merging(T=100, dt=0.1, open_loop=True):
    """RoundaboutMergingExample — n=24 (4 x Car6D).  src/roundabout_merging_example.cpp:76-436;
    params exec/roundabout_merging_example/main.cpp:73-75,108-113 (BASELINE config 4 asks
    for the open-loop LQ solver)."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.75
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    prm.open_loop = 1 if open_loop else 0
    s = ProblemSpec(T, dt, prm)
    for _ in range(4):
        s.add_player(DYN_CAR_6D, 4.0)
    f = np.float32
    off = f(math.pi / 2 * 0.5)
    wedge = f(math.pi)
    # float + double, rounded once when stored into the std::vector<float> (roundabout_merging_example.cpp:172-175)
    angles = [off, f(float(off) + 2.0 * math.pi / 4.0), f(float(off) + 2.0 * 2.0 * math.pi / 4.0),
              f(float(off) + 3.0 * 2.0 * math.pi / 4.0)]
    dists = [25.0, 10.0, 25.0, 10.0]
    speeds = [3.0, 2.0, 3.0, 2.0]
    x0 = np.zeros(24)
    s.position_dims, s.heading_dims, s.speed_dims = [], [], []
    for i in range(4):
        pts = roundabout_lane_center(angles[i], f(angles[i] + wedge), dists[i])
        lane = s.add_polyline(pts)
        X, Y, H, V, Aidx = 6 * i, 6 * i + 1, 6 * i + 2, 6 * i + 4, 6 * i + 5
        _lane_costs(s, i, lane, (X, Y), 25.0, 100.0, 2.5)
        (ax, ay), (bx, by) = pts[0], pts[1]
        x0[X], x0[Y] = ax, ay
        x0[H] = np.float32(math.atan2(f(by - ay), f(bx - ax)))  # LineSegment2::Heading
        x0[V] = speeds[i]
        s.position_dims.append((X, Y))
        s.heading_dims.append(H)
        s.speed_dims.append(V)
    for i in range(4):
        V = 6 * i + 4
        s.semiquadratic(i, 1000.0, V, 1.0, False)
        s.semiquadratic(i, 1000.0, V, 12.0, True)
        s.quadratic(i, 10.0, V, 10.0)
    for i in range(4):
        s.quadratic(i, 50.0, 5, 0.0)  # every player uses kP1AIdx (:356-365), reproduced
    for i in range(4):
        s.quadratic(i, 500.0, 0, 0.0, control_of=i)
        s.quadratic(i, 5.0, 1, 0.0, control_of=i)

    def xy(i):
        return (6 * i, 6 * i + 1)
    for i, (a, b) in enumerate(((1, 3), (0, 2), (1, 3), (0, 2))):  # :395-436 (which pairs are ADDED)
        s.proximity(i, 100.0, xy(i), xy(a), 6.0)
        s.proximity(i, 100.0, xy(i), xy(b), 6.0)
    s.x0 = x0
    return s


def three_player_collision_avoidance_reachability(T=100, dt=0.1, d0=5.0, v0=5.0, buffer=3.0):
    """ThreePlayerCollisionAvoidanceReachabilityExample — n=15 (3 x Car5D), max-over-time
    costs, control box constraints.  src/three_player_collision_avoidance_reachability_example.cpp:62-219;
    params exec/receding_horizon_three_player_collision_avoidance_reachability_example/main.cpp:74-81,116-124."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    for _ in range(3):
        s.add_player(DYN_CAR_5D, 4.0, structure=abi.MAX)
    X = [0, 5, 10]
    Y = [1, 6, 11]
    H = [2, 7, 12]
    V = [4, 9, 14]
    for i in range(3):
        s.quadratic(i, 0.1, -1, 0.0, control_of=i)
    for i in range(3):
        s.single_dimension_constraint(i, 0, 1.0, True, control_of=i)
        s.single_dimension_constraint(i, 0, -1.0, False, control_of=i)
        s.single_dimension_constraint(i, 1, 0.1, True, control_of=i)
        s.single_dimension_constraint(i, 1, -0.1, False, control_of=i)

    def sd(a, b):
        return lambda role: s.signed_distance(-1, (X[a], Y[a]), (X[b], Y[b]), buffer, True, role=role)
    children = [[sd(0, 1), sd(0, 2)], [sd(0, 1), sd(1, 2)], [sd(1, 2), sd(0, 2)]]
    for i in range(3):
        begin = len(s.terms)
        s.extreme_value(i, children[i], is_min=False)
        for t in s.terms[begin:]:
            t["player"] = i
    x0 = np.zeros(15)
    pert = 0.1
    f = np.float32
    x0[[X[0], Y[0], H[0], V[0]]] = [d0, 0.0, f(-math.pi + pert), v0]
    x0[[X[1], Y[1], H[1], V[1]]] = [-0.5 * d0, 0.5 * math.sqrt(3.0) * d0, f(-math.pi / 3.0 + pert), v0]
    x0[[X[2], Y[2], H[2], V[2]]] = [-0.5 * d0, -0.5 * math.sqrt(3.0) * d0, f(math.pi / 3.0 + pert), v0]
    s.x0 = x0
    s.position_dims = list(zip(X, Y))
    s.heading_dims = H
    s.speed_dims = V
    return s


def three_player_intersection_reachability(T=100, dt=0.1):
    """ThreePlayerIntersectionReachabilityExample — the n=14 intersection (Car5D, Car5D, Unicycle4D) with player 1
    as a max-over-time reachability player: its only state cost is the max of its signed distances to the two
    other players; players 2 and 3 keep the lane / speed / (zero-weight) proximity costs of the modified
    intersection.  src/three_player_intersection_reachability_example.cpp:72-318 (the reference ships no main
    for it; solver parameters are those of its other reachability mains)."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    s.add_player(DYN_CAR_5D, 4.0, state_reg=10.0, control_reg=10.0, structure=abi.MAX)
    s.add_player(DYN_CAR_5D, 4.0, state_reg=10.0, control_reg=10.0)
    s.add_player(DYN_UNICYCLE_4D, 4.0, state_reg=10.0, control_reg=10.0)
    P1X, P1Y, P1H, P1V = 0, 1, 2, 4
    P2X, P2Y, P2H, P2V = 5, 6, 7, 9
    P3X, P3Y, P3H, P3V = 10, 11, 12, 13
    p2x0, p3y0 = -10.0, 16.0
    lane2 = s.add_polyline([(p2x0, 1000.0), (p2x0, 28.0), (p2x0 + 0.5, 25.0), (p2x0 + 1.0, 24.0),
                            (p2x0 + 3.0, 22.5), (p2x0 + 6.0, 22.0), (1000.0, 22.0)])
    lane3 = s.add_polyline([(-1000.0, p3y0), (1000.0, p3y0)])
    _lane_costs(s, 1, lane2, (P2X, P2Y), 25.0, 100.0, 2.5)
    _lane_costs(s, 2, lane3, (P3X, P3Y), 25.0, 100.0, 2.5)
    for pl, vidx, vmax, vnom in ((1, P2V, 12.0, 6.0), (2, P3V, 2.0, 1.5)):
        s.semiquadratic(pl, 100.0, vidx, 1.0, False)   # MinV
        s.semiquadratic(pl, 100.0, vidx, vmax, True)   # MaxV
        s.quadratic(pl, 10.0, vidx, vnom)              # NominalV
    for pl in range(3):
        s.quadratic(pl, 0.1, 0, 0.0, control_of=pl)
        s.quadratic(pl, 0.1, 1, 0.0, control_of=pl)
    s.proximity(1, 0.0, (P2X, P2Y), (P1X, P1Y), 6.0)
    s.proximity(1, 0.0, (P2X, P2Y), (P3X, P3Y), 6.0)
    s.proximity(2, 0.0, (P3X, P3Y), (P1X, P1Y), 6.0)
    s.proximity(2, 0.0, (P3X, P3Y), (P2X, P2Y), 6.0)
    begin = len(s.terms)
    s.extreme_value(0, [lambda role: s.signed_distance(0, (P1X, P1Y), (P2X, P2Y), 6.0, True, role=role),
                        lambda role: s.signed_distance(0, (P1X, P1Y), (P3X, P3Y), 6.0, True, role=role)], is_min=False)
    f = np.float32
    x0 = np.zeros(14)
    x0[[P1X, P1Y, P1H, P1V]] = [-2.0, -30.0, float(f(np.pi / 2)), 4.0]
    x0[[P2X, P2Y, P2H, P2V]] = [-10.0, 45.0, float(f(-np.pi / 2)), 3.0]
    x0[[P3X, P3Y, P3H, P3V]] = [-11.0, 16.0, 0.0, 1.25]
    s.x0 = x0
    s.position_dims = [(P1X, P1Y), (P2X, P2Y), (P3X, P3Y)]
    s.heading_dims = [P1H, P2H, P3H]
    s.speed_dims = [P1V, P2V, P3V]
    return s


def three_player_overtaking(T=100, dt=0.1):
    """ThreePlayerOvertakingExample — n=18 (3 x Car6D): a fast car overtakes a slow one with a third ahead in the
    next lane.  src/three_player_overtaking_example.cpp:68-330 (player 3's proximity costs are constructed there
    but never added); params exec/three_player_overtaking/main.cpp:72-74,108-112."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.75
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    for _ in range(3):
        s.add_player(DYN_CAR_6D, 4.0)
    X, Y, H, V = [0, 6, 12], [1, 7, 13], [2, 8, 14], [4, 10, 16]
    lane1 = s.add_polyline([(-1.0, -1000.0), (-1.0, 1000.0)])
    lane2 = s.add_polyline([(2.5, -1000.0), (2.5, 1000.0)])
    for pl, lane in ((0, lane1), (1, lane1), (2, lane2)):
        _lane_costs(s, pl, lane, (X[pl], Y[pl]), 25.0, 100.0, 2.5)
    for pl, w, vnom in ((0, 10.0, 15.0), (1, 1.0, 10.0), (2, 1.0, 10.0)):
        s.quadratic(pl, w, V[pl], vnom)
    for pl in range(3):
        s.quadratic(pl, 500000.0, 0, 0.0, control_of=pl)  # steering rate
        s.quadratic(pl, 500.0, 1, 0.0, control_of=pl)     # jerk
    s.proximity(0, 100.0, (X[0], Y[0]), (X[1], Y[1]), 5.0)
    s.proximity(0, 100.0, (X[0], Y[0]), (X[2], Y[2]), 5.0)
    s.proximity(1, 100.0, (X[1], Y[1]), (X[0], Y[0]), 5.0)
    s.proximity(1, 100.0, (X[1], Y[1]), (X[2], Y[2]), 5.0)
    f = np.float32
    x0 = np.zeros(18)
    for pl, (px, py, v) in enumerate(((2.5, -10.0, 10.0), (-1.0, -10.0, 2.0), (2.5, 10.0, 2.0))):
        x0[[X[pl], Y[pl], H[pl], V[pl]]] = [px, py, float(f(np.pi / 2)), v]
    s.x0 = x0
    s.position_dims, s.heading_dims, s.speed_dims = list(zip(X, Y)), H, V
    return s


def two_player_collision(T=100, dt=0.1):
    """TwoPlayerCollisionExample — n=12 (2 x Car6D) driving at each other in one lane, player 1 with a side
    road to turn into; goal costs are FinalTimeCosts switched on for the last 0.5 s of the horizon.
    src/two_player_collision_example.cpp:66-318; params exec/two_player_collision/main.cpp:72-74,108-112."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.75
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    for _ in range(2):
        s.add_player(DYN_CAR_6D, 4.0, state_reg=1.0, control_reg=0.0)
    X, Y, H, V = [0, 6], [1, 7], [2, 8], [4, 10]
    xy = [(X[0], Y[0]), (X[1], Y[1])]
    lane_w, bound_w, half = 250.0, 50000.0, 2.5
    lane1 = s.add_polyline([(2.5, -50.0), (2.5, 50.0)])
    s.quadratic_polyline2(0, lane_w, lane1, xy[0])
    s.quadratic_polyline2(1, lane_w * 10, lane1, xy[1])
    s.semiquadratic_polyline2(0, bound_w * 1000, lane1, xy[0], -half, False)
    s.semiquadratic_polyline2(1, bound_w * 10, lane1, xy[1], -half, False)
    s.semiquadratic_polyline2(1, bound_w, lane1, xy[1], half, True)
    side = [([(2.5 + half, -50.0), (2.5 + half, -5.0)], True), ([(2.5 + half, 5.0), (2.5 + half, 50.0)], True),
            ([(10.0, -5.0), (10.0, 5.0)], True), ([(2.5 + half, 5.0), (25.0, 5.0)], False),
            ([(2.5 + half, -5.0), (25.0, -5.0)], True)]
    for pts, right in side:
        s.semiquadratic_polyline2(0, bound_w, s.add_polyline(pts), xy[0], 0.0, right)
    s.quadratic(0, 10.0, V[0], 5.0)
    s.quadratic(1, 1.0, V[1], 5.0)
    for pl in range(2):
        s.quadratic(pl, 5000.0, 0, 0.0, control_of=pl)
        s.quadratic(pl, 3250.0, 1, 0.0, control_of=pl)
    window = T * dt - 0.5  # time::kTimeHorizon - kFinalTimeWindow, double minus float
    s.final_time(window, s.quadratic(0, 1000.0, X[0], 2.5))
    s.final_time(window, s.quadratic(0, 1000.0, Y[0], 50.0))
    s.final_time(window, s.quadratic(1, 1000.0, X[1], 2.5))
    s.final_time(window, s.quadratic(1, 1000.0, Y[1], -50.0))
    s.proximity(0, 5000.0, xy[0], xy[1], 7.5)
    s.proximity(1, 5000.0, xy[1], xy[0], 7.5)
    f = np.float32
    x0 = np.zeros(12)
    x0[[X[0], Y[0], H[0], V[0]]] = [2.5, -50.0, float(f(np.pi / 2)), 10.0]
    x0[[X[1], Y[1], H[1], V[1]]] = [2.5, 50.0, float(f(-np.pi / 2)), 2.0]
    s.x0 = x0
    s.position_dims, s.heading_dims, s.speed_dims = xy, H, V
    return s


def one_player_reachability(T=100, dt=0.1, px0=1.75, py0=1.75, theta0=0.0):
    """OnePlayerReachabilityExample — a single Dubins car (n=3, one control) avoiding a disc of radius 2 around the
    origin: max-over-time signed distance to the disc's rim, a control cost and box constraints on the turn rate
    (augmented Lagrangian).  src/one_player_reachability_example.cpp:63-141 (its Polyline2SignedDistanceCost call
    passes (kAvoid, "Target") into (nominal, oriented_same_as_polyline): nominal 1, oriented true); params
    exec/one_player_reachability_example/main.cpp:80-82,117-121."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    s.add_player(DYN_DUBINS_CAR, 1.0, structure=abi.MAX)
    s.quadratic(0, 0.1, -1, 0.0, control_of=0)
    s.single_dimension_constraint(0, 0, 1.0, True, control_of=0)
    s.single_dimension_constraint(0, 0, -1.0, False, control_of=0)
    circle = s.add_polyline(draw_circle((0.0, 0.0), 2.0, 10))
    s.polyline2_signed_distance(0, circle, (0, 1), 1.0, True)
    s.x0 = [px0, py0, theta0]
    s.position_dims, s.heading_dims, s.speed_dims = [(0, 1)], [2], []
    return s


def dubins_origin(T=100, dt=0.1):
    """DubinsOriginExample — two Dubins cars (n=6, one control each): player 2 is attracted to player 1
    (QuadraticDifferenceCost), player 1 wants player 2 at the origin.  src/dubins_origin_example.cpp:54-128;
    params exec/dubins_origin_example/main.cpp:73-75,110-114."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.1
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    for _ in range(2):
        s.add_player(DYN_DUBINS_CAR, 1.0)
    s.quadratic_difference(1, 10.0, (0, 1), (3, 4))
    s.quadratic(0, 100.0, 0, 0.0, control_of=0)
    s.quadratic(1, 100.0, 0, 0.0, control_of=1)
    s.quadratic(0, 10.0, 3, 0.0)
    s.quadratic(0, 10.0, 4, 0.0)
    f = np.float32
    s.x0 = [0.0, -10.0, float(f(np.pi - 0.01)), 0.0, 10.0, float(f(1.5 * np.pi))]
    s.position_dims, s.heading_dims, s.speed_dims = [(0, 1), (3, 4)], [2, 5], []
    return s


def air_3d(T=100, dt=0.1, rx0=4.0, ry0=3.0, rtheta0=math.pi / 4.0, ve=1.0, vp=1.0):
    """Air3DExample — the two-aircraft pursuit-evasion game in relative coordinates (n=3, one turn rate per player):
    the evader (player 1) maximises over time, the pursuer minimises over time, the signed distance to a disc of
    radius 5; turn-rate box constraints on both.  src/air_3d_example.cpp:62-137 (its Polyline2SignedDistanceCost
    calls pass (!kReach, "Target") / (kReach, "Target") into (nominal, oriented_same_as_polyline)); params
    exec/air_3d_example/main.cpp:76-78,112-116."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    s.add_player(abi.DYN_AIR_3D_EVADER, ve, structure=abi.MAX)
    s.add_player(abi.DYN_AIR_3D_PURSUER, vp, structure=abi.MIN)
    for pl in range(2):
        s.quadratic(pl, 0.1, -1, 0.0, control_of=pl)
        s.single_dimension_constraint(pl, 0, 1.0, True, control_of=pl)
        s.single_dimension_constraint(pl, 0, -1.0, False, control_of=pl)
    circle = s.add_polyline(draw_circle((0.0, 0.0), 5.0, 10))
    s.polyline2_signed_distance(0, circle, (0, 1), 0.0, True)
    s.polyline2_signed_distance(1, circle, (0, 1), 1.0, True)
    s.x0 = [rx0, ry0, float(np.float32(rtheta0))]
    s.position_dims, s.heading_dims, s.speed_dims = [(0, 1)], [2], []
    return s


def modified_air_3d(T=100, dt=0.1, rx0=4.0, ry0=3.0, rtheta0=math.pi / 4.0, ve=1.0, vp=1.0):
    """ModifiedAir3DExample — pursuit-evasion between two planar point masses (n=8): the evader is paid, the
    pursuer charged, 1e6 times half the squared distance between them; quadratic acceleration costs.
    src/modified_air_3d_example.cpp:82-160 (state regularisation 1, control regularisation 0, :117-118); params
    exec/modified_air_3d_example/main.cpp:76-78,112-116."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    for pl in range(2):
        s.add_player(abi.DYN_POINT_MASS_2D, 0.0, state_reg=1.0, control_reg=0.0)
        s.quadratic(pl, 0.1, -1, 0.0, control_of=pl)
    s.quadratic_difference(0, -1e6, (0, 1), (4, 5))
    s.quadratic_difference(1, 1e6, (0, 1), (4, 5))
    f = np.float32
    s.x0 = [0.0, 0.0, float(f(ve)), 0.0, float(f(rx0)), float(f(ry0)), float(f(vp * math.cos(rtheta0))),
            float(f(vp * math.sin(rtheta0)))]
    s.position_dims, s.heading_dims, s.speed_dims = [(0, 1), (4, 5)], [], []
    return s


def skeleton(T=100, dt=0.1):
    """SkeletonExample — the reference's template problem: two Car5D (n=10) crossing paths, lane-centre, speed,
    control and proximity costs.  src/skeleton_example.cpp:60-185; params exec/skeleton_example/main.cpp:73-80,113-121."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.25
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    for _ in range(2):
        s.add_player(DYN_CAR_5D, 4.0)
    X, Y, H, V = [0, 5], [1, 6], [2, 7], [4, 9]
    for i in range(2):
        s.quadratic(i, 25.0, 0, 0.0, control_of=i)  # omega
        s.quadratic(i, 15.0, 1, 0.0, control_of=i)  # acceleration
    s.quadratic(0, 10.0, V[0], 8.0)
    s.quadratic(1, 10.0, V[1], 8.0)
    lane1 = s.add_polyline([(0.0, -1000.0), (0.0, 1000.0)])
    lane2 = s.add_polyline([(-5.0, 1000.0), (-5.0, 5.0), (0.0, 0.0), (995.0, 0.0)])
    s.quadratic_polyline2(0, 25.0, lane1, (X[0], Y[0]))
    s.quadratic_polyline2(1, 25.0, lane2, (X[1], Y[1]))
    for i in range(2):
        s.proximity(i, 100.0, (X[0], Y[0]), (X[1], Y[1]), 6.0)
    f = np.float32
    x0 = np.zeros(10)
    x0[[X[0], Y[0], H[0], V[0]]] = [0.0, -30.0, float(f(np.pi / 2)), 4.0]
    x0[[X[1], Y[1], H[1], V[1]]] = [-5.0, 30.0, float(f(-np.pi / 2)), 3.0]
    s.x0 = x0
    s.position_dims, s.heading_dims, s.speed_dims = list(zip(X, Y)), H, V
    return s


def two_player_collision_avoidance_reachability(T=100, dt=0.1, px0=0.0, py0=-5.0):
    """TwoPlayerCollisionAvoidanceReachabilityExample — n=10 (2 x Car5D), both players max-over-time of one shared
    SignedDistanceCost whose nominal is the players' distance half-way through the horizon when both drive straight.
    src/two_player_collision_avoidance_reachability_example.cpp:60-140; params
    exec/two_player_collision_avoidance_reachability_example/main.cpp:72-79,113-121."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    for _ in range(2):
        s.add_player(DYN_CAR_5D, 4.0, structure=abi.MAX)
    f = np.float32
    X, Y, H, V = [0, 5], [1, 6], [2, 7], [4, 9]
    h1, speed = f(0.1), f(5.0)
    half = 0.5 * (T * dt)  # 0.5 * time::kTimeHorizon, double
    reach = f(half * float(speed))  # t * speed narrows to the points' float when it scales them
    p1 = (f(px0) + reach * f(np.cos(np.float64(h1))), f(py0) + reach * f(np.sin(np.float64(h1))))
    p2 = (f(0.0) + reach * f(1.0), f(0.0) + reach * f(0.0))
    dx, dy = f(p1[0] - p2[0]), f(p1[1] - p2[1])
    nominal = float(np.sqrt(f(dx * dx + dy * dy), dtype=np.float32))
    for i in range(2):
        s.quadratic(i, 0.1, -1, 0.0, control_of=i)
        s.signed_distance(i, (X[0], Y[0]), (X[1], Y[1]), nominal, True)
    x0 = np.zeros(10)
    x0[[X[0], Y[0], H[0], V[0]]] = [px0, py0, float(h1), float(speed)]
    x0[[X[1], Y[1], H[1], V[1]]] = [0.0, 0.0, 0.0, 5.0]
    s.x0 = x0
    s.position_dims, s.heading_dims, s.speed_dims = list(zip(X, Y)), H, V
    return s


def draw_circle(center, radius, num_segments):
    """DrawCircle (src/draw_shapes.cpp:61-73), same float / double arithmetic."""
    f = np.float32
    pts = [(f(center[0]) + f(radius), f(center[1]) + f(0.0))]
    for ii in range(num_segments):
        angle = 2.0 * np.pi * float(f(ii + 1) / f(num_segments))
        pts.append((f(center[0]) + f(radius) * f(np.cos(angle)), f(center[1]) + f(radius) * f(np.sin(angle))))
    return [(float(x), float(y)) for x, y in pts]


def two_player_reachability(T=100, dt=0.1):
    """TwoPlayerReachabilityExample — TwoPlayerUnicycle4D (n=4), player 1 max-over-time, player 2 min-over-time of
    the signed distance to a unit circle around the origin.  src/two_player_reachability_example.cpp:62-126;
    params exec/two_player_reachability_example/main.cpp:71-78,110-119.  As written there the two
    Polyline2SignedDistanceCost calls pass (!kReach, "Target") / (kReach, "Target") into (nominal,
    oriented_same_as_polyline): nominal 0 resp. 1, oriented true for both."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.1
    s = ProblemSpec(T, dt, prm)
    s.add_player(DYN_UNICYCLE_4D_DISTURBED, 0.0, structure=abi.MAX)
    s.add_player(DYN_PLANAR_DISTURBANCE, 0.0, structure=abi.MIN)
    PX, PY, TH, V = 0, 1, 2, 3
    circle = s.add_polyline(draw_circle((0.0, 0.0), 1.0, 10))
    s.polyline2_signed_distance(0, circle, (PX, PY), 0.0, True)
    s.polyline2_signed_distance(1, circle, (PX, PY), 1.0, True)
    s.quadratic(0, 0.1, -1, 0.0, control_of=0)
    s.quadratic(1, 0.1, -1, 0.0, control_of=1)
    s.x0 = [0.0, -10.0, float(np.float32(np.pi / 4.0)), 5.0]
    s.position_dims, s.heading_dims, s.speed_dims = [(PX, PY)], [TH], [V]
    return s


def two_player_unicycle_4d_scene(T=100, dt=0.1):
    """A test scene on TwoPlayerUnicycle4D (include/ilqgames/dynamics/two_player_unicycle_4d.h:57-139): player 1
    steers the unicycle (omega, a) towards the origin at a nominal speed, player 2 pushes it with a bounded planar
    disturbance (dx, dy) towards a point of its own.  Initial state and control weight are those of
    src/two_player_reachability_example.cpp:62-66,74-75; that example's target cost (Polyline2SignedDistanceCost)
    is outside the built cost kinds, so quadratic / semiquadratic costs stand in — this is NOT a reference
    example, it exists to exercise the shared-state dynamics through the whole solve."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.5
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.001
    s = ProblemSpec(T, dt, prm)
    s.add_player(DYN_UNICYCLE_4D_DISTURBED, 0.0, state_reg=1.0, control_reg=1.0)
    s.add_player(DYN_PLANAR_DISTURBANCE, 0.0, state_reg=1.0, control_reg=1.0)
    PX, PY, TH, V = 0, 1, 2, 3
    s.quadratic(0, 1.0, PX, 0.0)
    s.quadratic(0, 1.0, PY, 0.0)
    s.quadratic(0, 2.0, V, 3.0)
    s.semiquadratic(0, 50.0, V, 8.0, True)
    s.quadratic(1, 0.5, PX, 6.0)
    s.quadratic(1, 0.5, PY, -4.0)
    s.quadratic(0, 0.1, -1, 0.0, control_of=0)
    s.quadratic(1, 0.1, -1, 0.0, control_of=1)
    s.quadratic(1, 5.0, -1, 0.0, control_of=1)  # keeps the disturbance small (kDMax = 0.5 m/s in the reference)
    s.quadratic(0, 0.05, -1, 0.0, control_of=1)
    s.x0 = [0.0, -10.0, float(np.float32(np.pi / 4.0)), 5.0]
    s.position_dims, s.heading_dims, s.speed_dims = [(PX, PY)], [TH], [V]
    return s


def cost_zoo_scene(T=100, dt=0.1):
    """A test scene, NOT a reference example: the two crossing Car5D of the skeleton example (n=10) carrying the cost
    and constraint kinds no reference example in CONFIGS uses — OrientationCost, QuadraticNormCost,
    SemiquadraticNormCost (on a state pair and on a control pair), RelativeDistanceCost, LocallyConvexProximityCost,
    CurvatureCost, Polyline2SignedDistanceConstraint and FinalTimeConstraint — so that every stage kernel and the whole solve are compared
    with the restatement on them.  CurvatureCost reads (phi, v) here: the reference pairs it with a yaw-rate state
    (Unicycle5D / Car7D); only its (omega_idx, v_idx) pattern matters to the kernels."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.001
    s = ProblemSpec(T, dt, prm)
    for _ in range(2):
        s.add_player(DYN_CAR_5D, 4.0)
    X, Y, H, PHI, V = [0, 5], [1, 6], [2, 7], [3, 8], [4, 9]
    for i in range(2):
        s.quadratic(i, 25.0, 0, 0.0, control_of=i)  # omega
        s.quadratic(i, 15.0, 1, 0.0, control_of=i)  # acceleration
        s.semiquadratic_norm(i, 40.0, (0, 1), 1.5, True, control_of=i)  # (QuadraticNormCost is singular at u = 0)
    s.quadratic(0, 10.0, V[0], 8.0)
    s.quadratic(1, 10.0, V[1], 8.0)
    lane1 = s.add_polyline([(0.0, -1000.0), (0.0, 1000.0)])
    lane2 = s.add_polyline([(-5.0, 1000.0), (-5.0, 5.0), (0.0, 0.0), (995.0, 0.0)])
    wall = s.add_polyline([(3.0, -1000.0), (3.0, -10.0), (0.2, 0.0), (3.0, 10.0), (3.0, 1000.0)])  # bulges into lane 1
    s.quadratic_polyline2(0, 25.0, lane1, (X[0], Y[0]))
    s.quadratic_polyline2(1, 25.0, lane2, (X[1], Y[1]))
    s.orientation(0, 5.0, H[0], float(np.float32(np.pi / 2)))
    s.orientation(1, 2.0, H[1], -1.0)
    s.quadratic_norm(0, 0.5, (X[0], Y[0]), 40.0)          # stay on a circle of radius 40 about the origin
    s.semiquadratic_norm(1, 0.5, (X[1], Y[1]), 45.0, True)  # and the other car inside radius 45
    s.semiquadratic_norm(1, 0.5, (X[1], Y[1]), 2.0, False)  # but not at the origin
    s.relative_distance(1, 1.0, (X[1], Y[1]), (X[0], Y[0]))
    for i in range(2):
        s.locally_convex_proximity(i, 50.0, (X[i], Y[i]), (X[1 - i], Y[1 - i]), 6.0)
        s.curvature(i, 20.0, PHI[i], V[i])
    s.polyline2_signed_distance_constraint(0, wall, (X[0], Y[0]), -0.5, True)
    # FinalTimeConstraint: from 4 s on player 2 must be within 4 m of lane 1's left side (it starts 5 m away)
    s.final_time(4.0, s.polyline2_signed_distance_constraint(1, lane1, (X[1], Y[1]), -4.0, False))
    f = np.float32
    x0 = np.zeros(10)
    x0[[X[0], Y[0], H[0], V[0]]] = [0.0, -30.0, float(f(np.pi / 2)), 4.0]
    x0[[X[1], Y[1], H[1], V[1]]] = [-5.0, 30.0, float(f(-np.pi / 2)), 3.0]
    s.x0 = x0
    s.position_dims, s.heading_dims, s.speed_dims = list(zip(X, Y)), H, V
    return s


# coefficients of affine_constraint_scene (its C++ twin: tests/host/zoo_scene.h)
AFFINE_A_U = [[1.0, 0.3], [-0.2, 1.0]]
AFFINE_B_U = [2.0, 3.0]


def affine_constraint_scene(T=100, dt=0.1):
    """A test scene, NOT a reference example: two Car5D (n = 10) following crossing lanes with the reference's two
    DENSE constraints (constraint/affine_scalar_constraint.h, affine_vector_constraint.h; its tests only run their
    quadraticisation, test/test_quadraticization.cpp:305-316): player 1 stays behind a line in the (px1, px2) plane
    (an inequality a^T x <= b on the whole state), player 2 keeps its heading tied to its speed (an EQUALITY
    a^T x = b), and player 1's controls are drawn to A u = b by an AffineVectorConstraint on its control vector (b
    far enough from zero that |A u - b| stays away from its singularity at 0).  Convex costs and the regularisation
    of the n = 14 example keep free-running solves of jittered instances well conditioned."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.1
    prm.expected_decrease_fraction = 0.001
    prm.max_solver_iters = 30
    s = ProblemSpec(T, dt, prm)
    for _ in range(2):
        s.add_player(DYN_CAR_5D, 4.0, state_reg=10.0, control_reg=10.0)
    X, Y, H, PHI, V = [0, 5], [1, 6], [2, 7], [3, 8], [4, 9]
    lane1 = s.add_polyline([(0.0, -1000.0), (0.0, 1000.0)])
    lane2 = s.add_polyline([(-1000.0, 2.0), (1000.0, 2.0)])
    for i in range(2):
        s.quadratic(i, 10.0, 0, 0.0, control_of=i)
        s.quadratic(i, 5.0, 1, 0.0, control_of=i)
        s.quadratic(i, 10.0, V[i], 6.0)
    s.quadratic_polyline2(0, 25.0, lane1, (X[0], Y[0]))
    s.quadratic_polyline2(1, 25.0, lane2, (X[1], Y[1]))
    a1 = [0.0] * 10
    a1[X[0]], a1[X[1]] = 1.0, -0.25
    s.affine_scalar_constraint(0, a1, 6.0)                      # px1 - 0.25 px2 <= 6
    a2 = [0.0] * 10
    a2[H[1]], a2[V[1]] = 1.0, -0.01
    s.affine_scalar_constraint(1, a2, -0.06, is_equality=True)  # theta2 = 0.01 v2 - 0.06
    s.affine_vector_constraint(0, AFFINE_A_U, AFFINE_B_U, control_of=0)
    f = np.float32
    x0 = np.zeros(10)
    x0[[X[0], Y[0], H[0], V[0]]] = [0.0, -25.0, float(f(np.pi / 2)), 5.0]
    x0[[X[1], Y[1], H[1], V[1]]] = [-30.0, 2.0, 0.0, 5.0]
    s.x0 = x0
    s.position_dims, s.heading_dims, s.speed_dims = list(zip(X, Y)), H, V
    return s


def weighted_proximity_scene(T=100, dt=0.1):
    """A test scene, NOT a reference example: the skeleton example's two Car5D with WeightedConvexProximityCost in
    place of its ProximityCost.  That cost's Quadraticize is not the derivative of its Evaluate in the reference
    (src/weighted_convex_proximity_cost.cpp:91-98: the speed gradient has the opposite sign), so an iLQ solve that
    leans on it goes nowhere in particular; the scene exists for the stage kernels and the cost evaluation, which are
    compared with the restatement wherever the cost is active."""
    s = skeleton(T, dt)
    s.terms = [t for t in s.terms if t["kind"] != abi.COST_PROXIMITY]
    X, Y, V = [0, 5], [1, 6], [4, 9]
    for i in range(2):
        s.weighted_convex_proximity(i, 0.02, (X[i], Y[i]), (X[1 - i], Y[1 - i]), V[i], V[1 - i], 40.0)
    return s


def dynamics_zoo_scene(T=100, dt=0.1):
    """A test scene, NOT a reference example: the single-player models no reference example uses — one Car7D and two
    Unicycle5D (n = 17) — on the crossing lanes of the skeleton example, with costs on the states only these models
    have (curvature kappa, path length s) and the two time-dependent costs (NominalPathLengthCost, RouteProgressCost)."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.001
    s = ProblemSpec(T, dt, prm)
    s.add_player(abi.DYN_CAR_7D, 4.0)
    s.add_player(abi.DYN_UNICYCLE_5D)
    s.add_player(abi.DYN_UNICYCLE_5D)
    X, Y, H, V = [0, 7, 12], [1, 8, 13], [2, 9, 14], [4, 10, 15]
    KAPPA, S = 5, [6, 11, 16]
    for i in range(3):
        s.quadratic(i, 25.0, 0, 0.0, control_of=i)  # omega
        s.quadratic(i, 15.0, 1, 0.0, control_of=i)  # acceleration
        s.quadratic(i, 10.0, V[i], 6.0)
        s.quadratic(i, 0.02, S[i], 50.0)           # path length driven towards 50 m
    s.quadratic(0, 30.0, KAPPA, 0.0)
    s.nominal_path_length(1, 0.5, S[1], 5.0)        # the time-dependent costs: path length 5 m/s * t ...
    lane1 = s.add_polyline([(0.0, -1000.0), (0.0, 1000.0)])
    lane2 = s.add_polyline([(-5.0, 1000.0), (-5.0, 5.0), (0.0, 0.0), (995.0, 0.0)])
    lane3 = s.add_polyline([(-1000.0, 8.0), (1000.0, 8.0)])
    for i, lane in enumerate((lane1, lane2, lane3)):
        s.quadratic_polyline2(i, 25.0, lane, (X[i], Y[i]))
    s.route_progress(0, 2.0, 6.0, lane1, (X[0], Y[0]), 970.0)   # ... and a point moving up lane 1 at 6 m/s from y = -30
    s.route_progress(1, 1.0, 4.0, lane2, (X[1], Y[1]), 968.0)   # (lane 2's corner at route position 995 is passed)
    for i in range(3):
        for j in range(3):
            if i != j:
                s.proximity(i, 100.0, (X[i], Y[i]), (X[j], Y[j]), 6.0)
    f = np.float32
    x0 = np.zeros(17)
    x0[[X[0], Y[0], H[0], V[0]]] = [0.0, -30.0, float(f(np.pi / 2)), 4.0]
    x0[[X[1], Y[1], H[1], V[1]]] = [-5.0, 30.0, float(f(-np.pi / 2)), 3.0]
    x0[[X[2], Y[2], H[2], V[2]]] = [-25.0, 8.0, 0.0, 5.0]
    s.x0 = x0
    s.position_dims, s.heading_dims, s.speed_dims = list(zip(X, Y)), H, V
    return s


def delayed_dubins_scene(T=100, dt=0.1):
    """A test scene, NOT a reference example: two SinglePlayerDelayedDubinsCar (n = 8, one control each) — the game
    of DubinsOriginExample (src/dubins_origin_example.cpp) with the turn rate as a state: player 1 is drawn to the
    origin, player 2 to player 1."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.01
    prm.expected_decrease_fraction = 0.001
    s = ProblemSpec(T, dt, prm)
    for _ in range(2):
        s.add_player(abi.DYN_DELAYED_DUBINS_CAR, 1.0)
    X, Y, H, W = [0, 4], [1, 5], [2, 6], [3, 7]
    for i in range(2):
        s.quadratic(i, 1.0, 0, 0.0, control_of=i)
        s.quadratic(i, 2.0, W[i], 0.0)
    s.quadratic(0, 1.0, X[0], 0.0)
    s.quadratic(0, 1.0, Y[0], 0.0)
    s.quadratic_difference(1, 1.0, (X[1], Y[1]), (X[0], Y[0]))
    s.x0 = [2.0, 1.0, float(np.float32(np.pi / 2)), 0.0, -1.0, -2.0, 0.3, 0.1]
    s.position_dims, s.heading_dims, s.speed_dims = list(zip(X, Y)), H, []
    return s


def mixed_dubins_car_scene(T=100, dt=0.1, open_loop=False, constrained=False):
    """A test scene, NOT a reference example: ConcatenatedDynamicalSystem({SinglePlayerDubinsCar, SinglePlayerCar5D})
    — n = 3 + 5, control dimensions (1, 2): players with DIFFERENT control dimensions, which only the
    run-time-dimensioned kernels run (src/concatenated_dynamical_system.cpp:52-66 and src/lq_feedback_solver.cpp:118-160
    walk cumulative dimensions; the reference's test/test_linearization.cpp:200-260 concatenates mixed models too).
    The Dubins car is drawn to a point while keeping clear of the car; the car follows a lane at a nominal speed."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.1
    prm.expected_decrease_fraction = 0.001
    prm.max_solver_iters = 40
    prm.open_loop = 1 if open_loop else 0
    s = ProblemSpec(T, dt, prm)
    s.add_player(abi.DYN_DUBINS_CAR, 2.0, state_reg=10.0, control_reg=10.0)
    s.add_player(abi.DYN_CAR_5D, 4.0, state_reg=10.0, control_reg=10.0)
    DX, DY, DH = 0, 1, 2
    CX, CY, CH, CPHI, CV = 3, 4, 5, 6, 7
    lane = s.add_polyline([(-50.0, -2.0), (0.0, -2.0), (20.0, 2.0), (80.0, 2.0)])
    s.quadratic(0, 2.0, DX, 12.0)
    s.quadratic(0, 2.0, DY, 3.0)
    s.proximity(0, 20.0, (DX, DY), (CX, CY), 4.0)
    s.quadratic(0, 5.0, 0, 0.0, control_of=0)
    s.quadratic_polyline2(1, 10.0, lane, (CX, CY))
    s.quadratic(1, 4.0, CV, 6.0)
    s.semiquadratic(1, 50.0, CPHI, 0.4, True)
    s.proximity(1, 20.0, (CX, CY), (DX, DY), 4.0)
    s.quadratic(1, 5.0, 0, 0.0, control_of=1)
    s.quadratic(1, 2.0, 1, 0.0, control_of=1)
    s.quadratic(1, 0.5, 0, 0.0, control_of=0)  # the car's cost sees the Dubins car's turn rate: an (1, 0) block
    if constrained:
        s.single_dimension_constraint(0, 0, 1.2, True, control_of=0)
        s.single_dimension_constraint(1, CV, 8.0, True)
        s.proximity_constraint(1, (CX, CY), (DX, DY), 2.0, False)
    s.x0 = [0.0, 4.0, float(np.float32(-0.4)), -10.0, -2.0, 0.0, 0.0, 5.0]
    s.position_dims, s.heading_dims, s.speed_dims = [(DX, DY), (CX, CY)], [DH, CH], [None, CV]
    return s


def three_unicycle_scene(T=40, dt=0.1, open_loop=False):
    """A test scene, NOT a reference example: three SinglePlayerUnicycle4D (n = 12, N = 3, m_i = 2) — equal control
    dimensions, but a shape the library holds no specialised instantiation of, and a horizon that is not the reference's
    100 steps: the run-time-dimensioned kernels again."""
    prm = SolverParams.default()
    prm.max_backtracking_steps = 100
    prm.initial_alpha_scaling = 0.1
    prm.convergence_tolerance = 0.1
    prm.expected_decrease_fraction = 0.001
    prm.max_solver_iters = 30
    prm.open_loop = 1 if open_loop else 0
    s = ProblemSpec(T, dt, prm)
    for _ in range(3):
        s.add_player(abi.DYN_UNICYCLE_4D, 0.0, state_reg=10.0, control_reg=10.0)
    X, Y, H, V = [0, 4, 8], [1, 5, 9], [2, 6, 10], [3, 7, 11]
    goals = [(8.0, 0.0), (0.0, 8.0), (-6.0, -6.0)]
    for i in range(3):
        s.quadratic(i, 2.0, X[i], goals[i][0])
        s.quadratic(i, 2.0, Y[i], goals[i][1])
        s.quadratic(i, 1.0, V[i], 2.0)
        for j in range(3):
            if j != i:
                s.proximity(i, 15.0, (X[i], Y[i]), (X[j], Y[j]), 3.0)
        s.quadratic(i, 2.0, 0, 0.0, control_of=i)
        s.quadratic(i, 2.0, 1, 0.0, control_of=i)
    s.x0 = [-8.0, 0.5, 0.0, 2.0, 0.5, -8.0, float(np.float32(np.pi / 2)), 2.0, 6.0, 6.5, float(np.float32(-2.3)), 2.0]
    s.position_dims, s.heading_dims, s.speed_dims = list(zip(X, Y)), H, V
    return s


def jittered_x0(spec, batch, seed=0):
    """Per-instance initial states of SURVEY.md §8(d): U(-1,1) m on px,py, U(-0.1,0.1) rad
    heading, U(-0.5,0.5) m/s speed; instance b uses numpy default_rng(seed + b)."""
    x0 = np.tile(np.asarray(spec.x0, dtype=np.float64), (batch, 1))
    for b in range(batch):
        rng = np.random.default_rng(seed + b)
        speeds = list(spec.speed_dims) + [None] * (len(spec.position_dims) - len(spec.speed_dims))
        headings = list(spec.heading_dims) + [None] * (len(spec.position_dims) - len(spec.heading_dims))
        for (xi, yi), hi, vi in zip(spec.position_dims, headings, speeds):
            x0[b, xi] += rng.uniform(-1, 1)
            x0[b, yi] += rng.uniform(-1, 1)
            dh = rng.uniform(-0.1, 0.1)
            if hi is not None:
                x0[b, hi] += dh
            dv = rng.uniform(-0.5, 0.5)  # drawn even for a model without a speed state: same stream per instance
            if vi is not None:
                x0[b, vi] += dv
    return x0


CONFIGS = {
    "modified_three_player_intersection": modified_three_player_intersection,
    "three_player_intersection": three_player_intersection,
    "roundabout_merging": roundabout_merging,
    "roundabout_merging_T150": lambda: roundabout_merging(T=150),  # BASELINE.json config 4 (n=24, T=150, open loop)
    # the same game on the feedback solver, as the reference's own main runs it (exec/roundabout_merging_example/main.cpp
    # leaves SolverParams::open_loop at its default): the 2 x 2-tile feedback sweep
    "roundabout_merging_feedback": lambda: roundabout_merging(open_loop=False),
    "three_player_collision_avoidance_reachability": three_player_collision_avoidance_reachability,
    "two_player_unicycle_4d_scene": two_player_unicycle_4d_scene,
    "two_player_reachability": two_player_reachability,
    "two_player_collision_avoidance_reachability": two_player_collision_avoidance_reachability,
    "skeleton": skeleton,
    "cost_zoo_scene": cost_zoo_scene,
    "weighted_proximity_scene": weighted_proximity_scene,
    "affine_constraint_scene": affine_constraint_scene,
    "dynamics_zoo_scene": dynamics_zoo_scene,
    "delayed_dubins_scene": delayed_dubins_scene,
    # shapes without a specialised instantiation: the run-time-dimensioned kernels
    "mixed_dubins_car_scene": mixed_dubins_car_scene,
    "mixed_dubins_car_scene_open_loop": lambda: mixed_dubins_car_scene(open_loop=True),
    "mixed_dubins_car_scene_constrained": lambda: mixed_dubins_car_scene(constrained=True),
    "three_unicycle_scene": three_unicycle_scene,
    "three_unicycle_scene_open_loop": lambda: three_unicycle_scene(open_loop=True),
    "air_3d": air_3d,
    "modified_air_3d": modified_air_3d,
    "dubins_origin": dubins_origin,
    "one_player_reachability": one_player_reachability,
    "two_player_collision": two_player_collision,
    "three_player_overtaking": three_player_overtaking,
    "three_player_intersection_reachability": three_player_intersection_reachability,
}
