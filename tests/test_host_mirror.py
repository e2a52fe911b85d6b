"""Host-side C++ mirror of the reference API (include/ilqgames/, ilqgames_amd/host/).

CPU part: the mirror compiles, and — where the reference checkout is present (this container, not
the GPU box) — the reference's own example problem definitions compile UNCHANGED against it and
flatten to the same descriptor as the hand-written builders in ilqgames_amd/examples.py.
GPU part: tests/host/host_solve_demo.cpp drives Problem / ILQSolver / AugmentedLagrangianSolver /
LQFeedbackSolver through the C ABI and is checked against the oracle.
"""
import os
import subprocess

import numpy as np
import pytest

from ilqgames_amd import abi, examples
import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BIN = os.path.join(ROOT, "tests", "host", "_bin")

EXAMPLES = [
    ("ModifiedThreePlayerIntersectionExample", "modified_three_player_intersection_example",
     examples.modified_three_player_intersection),
    ("ThreePlayerIntersectionExample", "three_player_intersection_example", examples.three_player_intersection),
    ("RoundaboutMergingExample", "roundabout_merging_example", examples.roundabout_merging),
    ("ThreePlayerCollisionAvoidanceReachabilityExample", "three_player_collision_avoidance_reachability_example",
     examples.three_player_collision_avoidance_reachability),
    ("TwoPlayerReachabilityExample", "two_player_reachability_example", examples.two_player_reachability),
    ("TwoPlayerCollisionAvoidanceReachabilityExample", "two_player_collision_avoidance_reachability_example",
     examples.two_player_collision_avoidance_reachability),
    ("SkeletonExample", "skeleton_example", examples.skeleton),
    ("ThreePlayerIntersectionReachabilityExample", "three_player_intersection_reachability_example",
     examples.three_player_intersection_reachability),
    ("ThreePlayerOvertakingExample", "three_player_overtaking_example", examples.three_player_overtaking),
    ("TwoPlayerCollisionExample", "two_player_collision_example", examples.two_player_collision),
    ("OnePlayerReachabilityExample", "one_player_reachability_example", examples.one_player_reachability),
    ("DubinsOriginExample", "dubins_origin_example", examples.dubins_origin),
    ("Air3DExample", "air_3d_example", examples.air_3d),
    ("ModifiedAir3DExample", "modified_air_3d_example", examples.modified_air_3d),
]


def _compile(out, sources, defines=()):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1"] + entry.host_compile_flags() + list(defines) + ["-o", out] + sources + \
        entry.host_link_flags()
    subprocess.check_call(cmd)


def test_host_library_builds_and_links_the_c_abi():
    so = entry.build_host()
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", so], text=True)
    # the mirror is a binding: it must resolve its numerics from the C ABI, not carry its own
    for s in ("ilqg_problem_create", "ilqg_ilq_solve_batch", "ilqg_al_solve_batch", "ilqg_lq_feedback_batch",
              "ilqg_lq_openloop_batch", "ilqg_workspace_bytes"):
        assert s in syms, s


def test_solver_log_on_disk_layout_and_row_format(tmp_path):
    """SolverLog::Save / SaveLogs write the reference's layout (src/solver_log.cpp:113-171, :208-240):
    <dir>/<experiment>/<iterate>/{t0,xs,u<i>,costs,cumulative_runtimes}.txt, one row per time step, Eigen's
    default formatting (6 significant digits, columns padded to a common width, single space)."""
    entry.build_host()
    exe = str(tmp_path / "log_format_check")
    _compile(exe, [os.path.join(ROOT, "tests", "host", "log_format_check.cpp")])
    env = dict(os.environ, ILQGAMES_LOG_DIR=str(tmp_path))
    assert subprocess.check_output([exe, "exp"], env=env, text=True).strip().splitlines()[-1] == "ok"
    base = tmp_path / "exp"
    assert sorted(os.listdir(base)) == ["0", "1"]
    assert sorted(os.listdir(base / "0")) == ["costs.txt", "cumulative_runtimes.txt", "t0.txt", "u0.txt", "u1.txt",
                                              "xs.txt"]
    assert (base / "0" / "t0.txt").read_text() == "1.5\n" and (base / "1" / "t0.txt").read_text() == "2\n"
    assert (base / "0" / "costs.txt").read_text() == "1.5\n2.25\n"
    assert (base / "1" / "costs.txt").read_text() == "0.75\n0.001\n"
    assert (base / "1" / "cumulative_runtimes.txt").read_text() == "0.25\n"
    xs = (base / "0" / "xs.txt").read_text().splitlines()
    assert xs[0] == "      0     -10 1214.57     -30"      # widest cell sets the width, 6 significant digits
    assert xs[1] == "   0.25   -9.75 1214.82  -29.75"
    assert np.allclose(np.loadtxt(base / "0" / "xs.txt")[2], [0.5, -9.5, 1215.07, -29.5], rtol=1e-5)
    assert (base / "0" / "u0.txt").read_text().splitlines()[1] == "      0.5 -0.333333"
    assert (base / "0" / "u1.txt").read_text().splitlines() == ["100", "101", "102"]
    # SaveLogs: one sub-experiment per log, last trajectory only
    lst = tmp_path / "exp_list"
    assert sorted(os.listdir(lst)) == ["0", "1"] and os.listdir(lst / "0") == ["1"]


def test_cpp_solution_splicer_keeps_five_rows_and_moves_the_start_time():
    """SolutionSplicer of the C++ mirror (tests/host/splicer_check.cpp, no device involved) against the layout
    src/solution_splicer.cpp:60-129 produces: up to five rows of the stored plan in front of the new solution,
    start time moved by the rows dropped, everything after the splice point replaced."""
    exe = os.path.join(BIN, "splicer_check")
    if not os.path.exists(exe):
        entry.build_host()
    lines = subprocess.check_output([exe], timeout=60).decode().splitlines()
    T, dt, n, m = 100, 0.1, 3, 2
    f = np.float32

    def rows_of(tag):  # the driver's MakeLog, same float32 arithmetic
        out = np.zeros((T, n + m + m + m * n), np.float32)
        for k in range(T):
            for e in range(n):
                out[k, e] = f(tag) + f(0.01) * f(k) + f(0.001) * f(e)
            for e in range(m):
                out[k, n + e] = -f(tag) - f(0.02) * f(k) + f(0.003) * f(e)
                out[k, n + m + e] = f(0.5) * f(tag) + f(0.001) * f(k * (e + 1))
                for c in range(n):  # column-major (m x n) gain
                    out[k, n + 2 * m + c * m + e] = f(tag) + f(0.1) * f(e) + f(0.01) * f(c) + f(0.0001) * f(k)
        return out

    def splice(plan, plan_t0, fresh, fresh_t0):
        at = int(1e-4 + (fresh_t0 - plan_t0) / dt)
        keep = min(at, 5)
        return np.concatenate([plan[at - keep:at], fresh]), plan_t0 + (at - keep) * dt

    it = iter(lines)
    cases = 0
    for line in it:
        tok = line.split()
        assert tok[0] == "case"
        start = float(tok[1])
        assert tok[3] == "1" and tok[4] == "0"  # ContainsTime inside / before the plan
        expect, t0 = splice(rows_of(1.0), 1.5, rows_of(2.0), 1.5 + start)
        for rnd in range(2):
            head = next(it).split()
            assert head[0] == "plan" and int(head[1]) == len(expect) and abs(float(head[2]) - t0) < 1e-6
            got = np.array([[float(v) for v in next(it).split()[1:]] for _ in range(len(expect))])
            assert np.max(np.abs(got - expect)) < 1e-5
            if rnd == 0:
                expect, t0 = splice(expect, t0, rows_of(3.0), t0 + 0.7)
        cases += 1
    assert cases == 5


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("cls,stem,builder", EXAMPLES, ids=[e[1] for e in EXAMPLES])
def test_reference_example_source_compiles_unchanged_and_flattens_like_examples_py(cls, stem, builder, tmp_path):
    entry.build_host()
    exe = str(tmp_path / ("dump_" + stem))
    _compile(exe, [os.path.join(ROOT, "tests", "host", "dump_example.cpp"), os.path.join(REF, "src", stem + ".cpp")],
             ['-DEXAMPLE_HEADER=<ilqgames/examples/%s.h>' % stem, "-DEXAMPLE_CLASS=" + cls])
    text = subprocess.check_output([exe], text=True)
    got = abi.ProblemSpec.from_dump(text)
    want = builder()
    g, w = got.canonical(), want.canonical()
    assert g["subsystems"] == w["subsystems"]
    assert g["player_costs"] == w["player_costs"]
    assert g["pairs"] == w["pairs"]
    assert (g["T"], g["dt"]) == (w["T"], w["dt"])
    assert set(g["groups"]) == set(w["groups"])
    for key in w["groups"]:
        assert g["groups"][key] == w["groups"][key], key
    np.testing.assert_allclose(np.array(got.x0, np.float32), np.array(want.x0, np.float32), rtol=1e-6, atol=1e-6)
    assert got.num_constraints == want.num_constraints


def test_every_reference_header_path_resolves_in_the_mirror(tmp_path):
    """A translation unit written for the reference keeps its #include lines: every header of the reference's
    include/ilqgames tree has a counterpart at the same path, except the GUI, the internal utilities no problem
    definition includes, and the classes DESIGN.md section 6 lists as not built (the flat systems)."""
    not_mirrored = {"dynamics/concatenated_flat_system.h", "dynamics/multi_player_flat_system.h",
                    "dynamics/single_player_flat_car_6d.h", "dynamics/single_player_flat_system.h",
                    "dynamics/single_player_flat_unicycle_4d.h", "examples/flat_roundabout_merging_example.h",
                    "examples/three_player_flat_intersection_example.h", "examples/three_player_flat_overtaking_example.h",
                    "gui/control_sliders.h", "gui/cost_inspector.h", "gui/top_down_renderer.h",
                    "solver/solve_feedback_lq_game.h", "utils/loop_timer.h", "utils/make_directory.h",
                    "utils/player_cost_cache.h", "utils/relative_time_tracker.h", "utils/uncopyable.h"}
    inc = os.path.join(ROOT, "include", "ilqgames")
    mine = set()
    for dirpath, _, files in os.walk(inc):
        for f in files:
            rel = os.path.relpath(os.path.join(dirpath, f), inc)
            if f.endswith(".h") and not rel.startswith("host"):
                mine.add(rel)
    if os.path.isdir(REF):  # the list above is the whole difference (checked where the reference is present)
        theirs = set()
        ref_inc = os.path.join(REF, "include", "ilqgames")
        for dirpath, _, files in os.walk(ref_inc):
            for f in files:
                if f.endswith(".h"):
                    theirs.add(os.path.relpath(os.path.join(dirpath, f), ref_inc))
        assert theirs - mine == not_mirrored
    src = tmp_path / "all_headers.cpp"
    src.write_text("".join("#include <ilqgames/%s>\n" % h for h in sorted(mine)) + "int main() { return 0; }\n")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only"] + entry.host_compile_flags() + [str(src)])


@pytest.mark.parametrize("cls,builder,nc", [("CostZooScene", examples.cost_zoo_scene, 2),
                                            ("AffineConstraintScene", examples.affine_constraint_scene, 3),
                                            ("WeightedProximityScene", examples.weighted_proximity_scene, 0),
                                            ("DynamicsZooScene", examples.dynamics_zoo_scene, 0),
                                            ("DelayedDubinsScene", examples.delayed_dubins_scene, 0)])
def test_cpp_zoo_scenes_flatten_like_examples_py(tmp_path, cls, builder, nc):
    """The mirrored classes no reference example uses — the costs OrientationCost, QuadraticNormCost,
    SemiquadraticNormCost, RelativeDistanceCost, LocallyConvexProximityCost, WeightedConvexProximityCost, CurvatureCost,
    NominalPathLengthCost, RouteProgressCost, the constraints FinalTimeConstraint and
    Polyline2SignedDistanceConstraint, the models SinglePlayerCar7D, SinglePlayerUnicycle5D,
    SinglePlayerDelayedDubinsCar: tests/host/zoo_scene.h builds the scenes of examples.py through them."""
    entry.build_host()
    exe = str(tmp_path / "dump_zoo")
    _compile(exe, [os.path.join(ROOT, "tests", "host", "dump_example.cpp")],
             ['-DEXAMPLE_HEADER="zoo_scene.h"', "-DEXAMPLE_CLASS=" + cls, "-I" + os.path.join(ROOT, "tests", "host")])
    got = abi.ProblemSpec.from_dump(subprocess.check_output([exe], text=True))
    want = builder()
    g, w = got.canonical(), want.canonical()
    for key in ("subsystems", "player_costs", "pairs", "T", "dt"):
        assert g[key] == w[key], key
    assert set(g["groups"]) == set(w["groups"])
    for key in w["groups"]:
        assert g["groups"][key] == w["groups"][key], key
    np.testing.assert_allclose(np.array(got.x0, np.float32), np.array(want.x0, np.float32), rtol=1e-6, atol=1e-6)
    assert got.num_constraints == want.num_constraints == nc


# ------------------------------------------------------------------------------------------------
# GPU: the C++ mirror end to end (Problem -> descriptor -> C ABI -> kernels -> SolverLog)
# ------------------------------------------------------------------------------------------------
def _as_host_floats(tokens):
    """An initial state the demo printed from its float containers (nine significant digits round-trip a float, not a
    double): the values the DEVICE was given are those floats widened to double, so the oracle gets exactly those — a
    decimal string parsed straight to double is up to 5e-10 relative away, and these scenes' line searches flip on 1e-12
    (measured: scripts ran the oracle from x0 nudged by 1e-12 and every instance of the receding-horizon scene changed its
    outcome)."""
    return np.array([float(v) for v in tokens], dtype=np.float32).astype(np.float64)


def _parse_log(path):
    out = dict(xs=[], us=[], alpha=[])
    for line in open(path):
        tok = line.split()
        if tok[0] == "x0":
            out["x0"] = _as_host_floats(tok[1:])
        elif tok[0] == "success":
            out["success"], out["converged"], out["iters"] = int(tok[1]), int(tok[3]), int(tok[5])
        elif tok[0] == "costs":
            out["costs"] = np.array([float(v) for v in tok[1:]])
        elif tok[0] == "x":
            out["xs"].append([float(v) for v in tok[1:]])
        elif tok[0] == "u":
            out["us"].append([float(v) for v in tok[1:]])
        elif tok[0] == "alpha":
            out["alpha"].append([float(v) for v in tok[1:]])
    for k in ("xs", "us", "alpha"):
        out[k] = np.array(out[k])
    return out


@pytest.fixture(scope="module")
def demo_out(tmp_path_factory):
    exe = os.path.join(BIN, "host_solve_demo")
    if not os.path.exists(exe):
        entry.build_host()
    out = str(tmp_path_factory.mktemp("host_demo"))
    subprocess.check_call([exe, out], timeout=600)
    return out


def _oracle_solve(oracle, scene_file, x0, augmented_lagrangian=False):
    spec = abi.ProblemSpec.from_dump(open(scene_file).read())
    return spec, oracle.OracleProblem(spec).solve(abi.F64, x0[None, :], augmented_lagrangian=augmented_lagrangian)


def _check_against_oracle(got, ref, tol=2e-4):
    # host containers are float (as the reference's), the device solve ran in fp64
    assert got["iters"] == int(ref["iters"][0])
    assert got["success"] == int(ref["status"][0])
    assert got["converged"] == int(ref["converged"][0])
    scale = max(1.0, np.max(np.abs(ref["xs"])))
    assert np.max(np.abs(got["xs"] - ref["xs"][0])) < tol * scale
    assert np.max(np.abs(got["us"] - ref["us"][0])) < tol * max(1.0, np.max(np.abs(ref["us"])))
    assert np.max(np.abs(got["alpha"] - ref["alpha"][0])) < tol * max(1.0, np.max(np.abs(ref["alpha"])))
    np.testing.assert_allclose(got["costs"], ref["costs"][0], rtol=1e-4)


@pytest.mark.gpu
def test_cpp_ilq_solver_single_solve_matches_oracle(demo_out, oracle):
    got = _parse_log(os.path.join(demo_out, "ilq_single.txt"))
    spec, ref = _oracle_solve(oracle, os.path.join(demo_out, "scene.txt"), got["x0"])
    assert got["iters"] >= 2
    _check_against_oracle(got, ref)


@pytest.mark.gpu
def test_cpp_device_produced_solver_log_on_disk_matches_oracle_iterates(demo_out, oracle):
    """SURVEY 8(f) item 3 end to end: the log of a solve that ran on the device — every iterate, copied by the device
    into an ilqg_iterate_log — is written by SolverLog::Save in the reference's layout (src/solver_log.cpp:113-171:
    <dir>/<experiment>/<iterate>/{t0,xs,u<i>,costs,cumulative_runtimes}.txt), read back from disk and compared with
    the oracle's iterates: iterate 0 is the initial rollout (src/ilq_solver.cpp:100-112), iterate q the operating
    point after q outer iterations.  The files hold six significant digits (Eigen's default stream precision)."""
    # a solve whose second line search gives up (src/ilq_solver.cpp:146-155 returns the log as it stands): iterates 0, 1
    failed = _parse_log(os.path.join(demo_out, "ilq_single.txt"))
    meta = open(os.path.join(demo_out, "ilq_single_log_meta.txt")).read().split()
    assert failed["success"] == 0 and int(meta[1]) == int(meta[3]) == failed["iters"] == 2
    assert sorted(os.listdir(os.path.join(demo_out, "ilq_single_log"))) == ["0", "1"]
    # a solve that runs its six iterations: iterates 0 .. 6
    got = _parse_log(os.path.join(demo_out, "ilq_logged.txt"))
    spec = abi.ProblemSpec.from_dump(open(os.path.join(demo_out, "scene_logged.txt")).read())
    O = oracle.OracleProblem(spec)
    meta = open(os.path.join(demo_out, "ilq_logged_log_meta.txt")).read().split()
    iterates, device_iterations = int(meta[1]), int(meta[3])
    assert got["success"] == 1 and iterates == device_iterations + 1 == got["iters"] + 1 == 7
    base = os.path.join(demo_out, "ilq_logged_log")
    assert sorted(os.listdir(base), key=int) == [str(q) for q in range(iterates)]
    x0 = got["x0"][None, :]
    udims = [sub[2] for sub in spec.subsystems]
    for q in range(iterates):
        d = os.path.join(base, str(q))
        assert sorted(os.listdir(d)) == sorted(["t0.txt", "xs.txt", "costs.txt", "cumulative_runtimes.txt"] +
                                               ["u%d.txt" % i for i in range(len(udims))])
        if q == 0:  # the warm start (zero strategies about a zero operating point) played from x0
            z = lambda *shape: np.zeros(shape)  # noqa: E731
            xs, us = O.rollout(abi.F64, x0, z(1, spec.T, spec.n), z(1, spec.T, spec.m), z(1, spec.T, spec.m * spec.n),
                               z(1, spec.T, spec.m))
            costs, _ = O.total_costs(abi.F64, xs, us)
        else:
            ref = O.solve(abi.F64, x0, fixed_iters=q)
            xs, us, costs = ref["xs"], ref["us"], ref["costs"]
        scale = max(1.0, np.max(np.abs(xs)))
        assert np.max(np.abs(np.loadtxt(os.path.join(d, "xs.txt")) - xs[0])) < 2e-5 * scale, q
        off = 0
        for i, mi in enumerate(udims):
            ui = np.loadtxt(os.path.join(d, "u%d.txt" % i)).reshape(spec.T, mi)
            assert np.max(np.abs(ui - us[0][:, off:off + mi])) < 2e-5 * max(1.0, np.max(np.abs(us))), (q, i)
            off += mi
        np.testing.assert_allclose(np.loadtxt(os.path.join(d, "costs.txt")), costs[0], rtol=2e-5)
        assert float(open(os.path.join(d, "t0.txt")).read()) == 0.0
    # the last iterate on disk is the solve's result; SaveLogs keeps only that one
    last = os.path.join(demo_out, "ilq_logged_last", "0")
    assert os.listdir(last) == [str(iterates - 1)]
    assert open(os.path.join(last, str(iterates - 1), "xs.txt")).read() == \
        open(os.path.join(base, str(iterates - 1), "xs.txt")).read()
    assert np.max(np.abs(np.loadtxt(os.path.join(base, str(iterates - 1), "xs.txt")) - got["xs"])) < \
        2e-5 * max(1.0, np.max(np.abs(got["xs"])))


@pytest.mark.gpu
def test_cpp_solve_honours_max_runtime_like_the_reference_loop(demo_out, oracle):
    """ILQSolver::Solve(success, max_runtime) (src/ilq_solver.cpp:123-124): with a budget no iteration fits in
    (elapsed = 0 is not below 1e-6 - RuntimeUpperBound()) the loop body never runs — the log holds iterate 0, the
    call reports success; with a generous budget the solve is the unbudgeted one."""
    lines = open(os.path.join(demo_out, "ilq_deadline.txt")).read().splitlines()
    assert lines[1].split() == ["iterates", "1"]
    got = _parse_log(os.path.join(demo_out, "ilq_deadline.txt"))
    assert got["success"] == 1 and got["converged"] == 0 and got["iters"] == 0
    spec = abi.ProblemSpec.from_dump(open(os.path.join(demo_out, "scene.txt")).read())
    O = oracle.OracleProblem(spec)
    z = lambda *shape: np.zeros(shape)  # noqa: E731
    xs, us = O.rollout(abi.F64, got["x0"][None, :], z(1, spec.T, spec.n), z(1, spec.T, spec.m),
                       z(1, spec.T, spec.m * spec.n), z(1, spec.T, spec.m))
    assert np.max(np.abs(got["xs"] - xs[0])) < 2e-4 * max(1.0, np.max(np.abs(xs)))
    assert np.max(np.abs(got["alpha"])) == 0.0
    relaxed = _parse_log(os.path.join(demo_out, "ilq_relaxed.txt"))
    plain = _parse_log(os.path.join(demo_out, "ilq_single.txt"))
    assert relaxed["iters"] == plain["iters"] and relaxed["success"] == plain["success"]
    assert np.array_equal(relaxed["xs"], plain["xs"]) and np.array_equal(relaxed["alpha"], plain["alpha"])


@pytest.mark.gpu
def test_cpp_receding_horizon_resync_matches_oracle(demo_out, oracle):
    """Problem::SetUpNextRecedingHorizon through the C++ mirror (float containers, fp64 device) against the
    oracle's restatement applied to the same solved plan."""
    got = _parse_log(os.path.join(demo_out, "ilq_single.txt"))
    spec, ref = _oracle_solve(oracle, os.path.join(demo_out, "scene.txt"), got["x0"])
    rows = {}
    for line in open(os.path.join(demo_out, "receding.txt")):
        tok = line.split()
        rows.setdefault(tok[0], []).append([float(v) for v in tok[1:]])
    x_meas = np.array(rows["x_meas"][0])
    # the C++ side adopted the float-rounded device solution; do the same with the oracle's plan
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    o = oracle.OracleProblem(spec).receding_horizon_shift(abi.F64, x_meas[None, :], 0.33, 0.25, 0.0, f32(ref["xs"]),
                                                          f32(ref["us"]), f32(ref["P"]), f32(ref["alpha"]))
    assert abs(rows["t0"][0][0] - o["new_plan_t0"]) < 1e-6
    assert np.max(np.abs(np.array(rows["x0"][0]) - o["x0_next"][0])) < 2e-4
    assert np.max(np.abs(np.array(rows["x"]) - o["xs"][0])) < 2e-4 * max(1.0, np.max(np.abs(o["xs"])))
    assert np.max(np.abs(np.array(rows["u"]) - o["us"][0])) < 2e-4 * max(1.0, np.max(np.abs(o["us"])))
    assert int(o["first_step"][0]) > 0


def _parse_rh(path):
    logs, calls = [], 0
    for line in open(path):
        tok = line.split()
        if tok[0] == "calls":
            calls = int(tok[1])
        elif tok[0] == "t0":
            logs.append(dict(t0=float(tok[1]), xs=[], us=[]))
        elif tok[0] == "success":
            logs[-1]["converged"], logs[-1]["iters"] = int(tok[3]), int(tok[5])
        elif tok[0] == "x":
            logs[-1]["xs"].append([float(v) for v in tok[1:]])
        elif tok[0] == "u":
            logs[-1]["us"].append([float(v) for v in tok[1:]])
    assert calls == len(logs)
    return logs


def _compare_rh_run(oracle, spec, logs, tag, compare_us=False):
    """A device receding-horizon run (parsed demo log) against the oracle's from the same initial state, call by call.
    This scene's line searches are decided by rounding — the oracle run from x0 nudged by 1e-12 changes the outcome of
    the FIRST call of every instance tried (iterations, success, convergence flag; measured with scripts on the GPU
    box) — so two correct implementations may part ways at some call.  Where the device's flags first differ from the
    oracle's, the oracle must itself produce more than one outcome for the calls up to there under such nudges;
    otherwise that is a real difference and the test fails.  Every call before the divergence is held to the usual
    tolerances.  Returns the number of calls compared."""
    x0 = _as_host_floats(logs[0]["xs"][0])
    O = oracle.OracleProblem(spec)
    ref = O.receding_horizon_simulate(abi.F64, x0[None, :], 3.0, 0.25, max_records=32)
    R = int(ref["num_records"][0])
    div = None
    for r in range(max(R, len(logs))):
        if r >= R or r >= len(logs) or logs[r]["iters"] != ref["iters"][0, r] or logs[r]["converged"] != ref["converged"][0, r]:
            div = r
            break
    upto = len(logs) if div is None else div
    for r in range(upto):
        log = logs[r]
        assert abs(log["t0"] - ref["plan_t0"][0, r]) < 1e-6, (tag, r)
        xs = np.array(log["xs"])
        assert np.max(np.abs(xs - ref["xs"][0, r])) < 5e-4 * max(1.0, np.max(np.abs(ref["xs"][0, r]))), (tag, r)
        if compare_us:
            us = np.array(log["us"])
            assert np.max(np.abs(us - ref["us"][0, r])) < 5e-4 * max(1.0, np.max(np.abs(ref["us"][0, r]))), (tag, r)
    if div is not None:
        # the flags the device saw at the diverging call (or "gone" when it left the loop earlier than the oracle)
        # must be an outcome the oracle itself reaches from a nudged x0 — counting only nudged runs that still agree
        # with the reference run on every call before the divergence (a run that parted ways earlier says nothing
        # about call `div`)
        def outcome_at(num_records, iters, converged):
            if num_records <= div:
                return (False, -1, -1)  # left the loop before call `div`
            return (True, int(iters[div]), int(converged[div]))

        def agrees_before(run, num_records):
            if min(num_records, R) < div:
                return False
            return all(int(run["iters"][0, r]) == int(ref["iters"][0, r]) and
                       int(run["converged"][0, r]) == int(ref["converged"][0, r]) for r in range(div))

        dev_outcome = outcome_at(len(logs), [g["iters"] for g in logs], [g["converged"] for g in logs])
        ref_outcome = outcome_at(R, ref["iters"][0], ref["converged"][0])
        assert dev_outcome != ref_outcome
        rng = np.random.default_rng(99)
        outcomes = {ref_outcome}
        for scale in [1e-13, 1e-12, 1e-12, 1e-11, 1e-11, 1e-10, 1e-10, 1e-9] * 8:  # 64 nudged runs: the outcomes of this scene's calls are few
            if dev_outcome in outcomes:
                break
            again = O.receding_horizon_simulate(abi.F64, (x0 + scale * rng.standard_normal(x0.shape))[None, :], 3.0, 0.25,
                                                max_records=32)
            Ra = int(again["num_records"][0])
            if agrees_before(again, Ra):
                outcomes.add(outcome_at(Ra, again["iters"][0], again["converged"][0]))
        assert dev_outcome in outcomes, (tag, "the device parts ways with the oracle at call", div, "with", dev_outcome,
                                         "which the oracle does not reach from nudged initial states:", sorted(outcomes))
    return upto, ref


@pytest.mark.gpu
def test_cpp_receding_horizon_batch_matches_oracle(demo_out, oracle):
    """host::RecedingHorizonSimulatorBatch (plans, states and solver workspace resident on the device between the
    solver calls) against the oracle's loop, instance by instance."""
    spec = abi.ProblemSpec.from_dump(open(os.path.join(demo_out, "scene_rh.txt")).read())
    compared = 0
    for b in range(3):
        logs = _parse_rh(os.path.join(demo_out, "rh_batch_%d.txt" % b))
        compared += _compare_rh_run(oracle, spec, logs, b)[0]
    assert compared >= 6, "most calls of the three runs must have been compared before any rounding-decided divergence"


@pytest.mark.gpu
def test_cpp_receding_horizon_simulator_matches_oracle(demo_out, oracle):
    """RecedingHorizonSimulator of the C++ mirror (Integrate, OverwriteSolution, SetUpNextRecedingHorizon, repeated
    Solve() on one ILQSolver, SolutionSplicer) with a fixed 0.25 s per call, against the oracle's restatement of
    the same loop: same number of solver calls, same window start times and flags, trajectories within the float
    round-off of the host containers."""
    logs = []
    for line in open(os.path.join(demo_out, "rh_sim.txt")):
        tok = line.split()
        if tok[0] == "calls":
            calls = int(tok[1])
        elif tok[0] == "t0":
            logs.append(dict(t0=float(tok[1]), xs=[], us=[]))
        elif tok[0] == "success":
            logs[-1]["converged"], logs[-1]["iters"] = int(tok[3]), int(tok[5])
        elif tok[0] == "x":
            logs[-1]["xs"].append([float(v) for v in tok[1:]])
        elif tok[0] == "u":
            logs[-1]["us"].append([float(v) for v in tok[1:]])
    assert calls == len(logs) and calls >= 4
    spec = abi.ProblemSpec.from_dump(open(os.path.join(demo_out, "scene_rh.txt")).read())
    upto, ref = _compare_rh_run(oracle, spec, logs, "sim", compare_us=True)
    assert upto >= 2
    # the second call already starts from the carried merit value: its line search fails on the first iteration
    assert ref["iters"][0, 1] == 1 and ref["ok"][0, 1] == 0


def _parse_mi(path):
    active, logs = None, []
    for line in open(path):
        tok = line.split()
        if tok[0] == "active":
            active = [int(v) for v in tok[1:]]
        elif tok[0] == "calls":
            calls = (int(tok[1]), int(tok[2]))
        elif tok[0] == "t0":
            logs.append(dict(t0=float(tok[1]), xs=[], us=[]))
        elif tok[0] == "success":
            logs[-1]["converged"], logs[-1]["iters"] = int(tok[3]), int(tok[5])
        elif tok[0] == "costs":
            logs[-1]["costs"] = [float(v) for v in tok[1:]]
        elif tok[0] == "x":
            logs[-1]["xs"].append([float(v) for v in tok[1:]])
        elif tok[0] == "u":
            logs[-1]["us"].append([float(v) for v in tok[1:]])
    assert calls[0] == calls[1] and 2 * calls[0] == len(logs)
    return active, logs[0::2], logs[1::2]


@pytest.mark.gpu
def test_cpp_minimally_invasive_simulator(demo_out):
    """MinimallyInvasiveRecedingHorizonSimulator of the C++ mirror (src/minimally_invasive_receding_horizon_
    simulator.cpp:68-218).  Same scene on both sides: the two planners see identical inputs at every call, so
    their logs are identical, and the positive P1 value hands every decision to the safety planner.  Safety scene
    with a P1 value below the threshold: every decision follows the rule of :201-214 applied to the logs, and
    both branches' start times advance by 0.25 s of motion plus the 0.25 s charged per pair of calls."""
    rh_first = _parse_rh(os.path.join(demo_out, "rh_sim.txt"))[0]
    for variant in (0, 1):
        active, logs_a, logs_b = _parse_mi(os.path.join(demo_out, "mi_sim_%d.txt" % variant))
        assert len(logs_a) >= 4 and active[0] == 0
        # one decision per completed pair of re-solves; the loop may end right after the last pair's solves
        assert len(active) in (len(logs_a), len(logs_a) - 1), (len(active), len(logs_a))
        assert np.array_equal(np.array(logs_a[0]["xs"]), np.array(rh_first["xs"]))  # the original planner's first solve
        for r in range(1, len(logs_a)):
            assert abs(logs_a[r]["t0"] - 0.5 * r) < 1e-5 and abs(logs_b[r]["t0"] - 0.5 * r) < 1e-5, (variant, r)
            assert np.array_equal(np.array(logs_a[r]["xs"])[0], np.array(logs_b[r]["xs"])[0]), (variant, r)
        for r in range(1, len(active)):
            a, b = logs_a[r], logs_b[r]
            want = 1 if (b["costs"][0] > -1.0 or (b["converged"] and not a["converged"])) else 0
            assert active[r] == want, (variant, r, active, b["costs"][0], a["converged"], b["converged"])
        if variant == 0:
            assert all(v == 1 for v in active[1:])
            for a, b in zip(logs_a, logs_b):
                assert a["iters"] == b["iters"] and a["converged"] == b["converged"]
                assert np.array_equal(np.array(a["xs"]), np.array(b["xs"]))
                assert np.array_equal(np.array(a["us"]), np.array(b["us"]))
        else:
            assert all(b["costs"][0] < -1.0 for b in logs_b)
            assert any(v == 0 for v in active[1:])


@pytest.mark.gpu
def test_cpp_two_player_unicycle_solve_matches_oracle(demo_out, oracle):
    """TwoPlayerUnicycle4D through the C++ mirror: its descriptor equals the python builder's, and five iLQ
    iterations on the device match the oracle."""
    got = _parse_log(os.path.join(demo_out, "unicycle_single.txt"))
    spec, ref = _oracle_solve(oracle, os.path.join(demo_out, "scene_unicycle.txt"), got["x0"])
    want = examples.two_player_unicycle_4d_scene()
    a, b = spec.canonical(), want.canonical()
    assert a["subsystems"] == b["subsystems"] and a["groups"] == b["groups"] and a["pairs"] == b["pairs"]
    assert [s[0] for s in spec.subsystems] == [abi.DYN_UNICYCLE_4D_DISTURBED, abi.DYN_PLANAR_DISTURBANCE]
    assert got["iters"] == 5
    _check_against_oracle(got, ref)
    # ComputeStrategyCosts / NumericalCheckLocalNashEquilibrium of the mirror on the adopted (float) solution
    rows = {ln.split()[0]: [float(v) for v in ln.split()[1:]] for ln in open(os.path.join(demo_out, "unicycle_checks.txt"))}
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    op = oracle.OracleProblem(spec)
    sol = [f32(ref[k]) for k in ("xs", "us", "P", "alpha")]
    want_costs = op.strategy_costs(abi.F64, got["x0"][None, :], *sol, euler=False)
    np.testing.assert_allclose(rows["costs"], want_costs[0], rtol=1e-5)
    ok_large, margin = op.check_local_nash(abi.F64, got["x0"][None, :], *sol, 0.5)
    assert rows["nash_small"] == [1.0]
    assert abs(margin[0]) > 1e-3 and rows["nash_large"] == [float(ok_large[0])]
    psd, worst = op.check_sufficient_nash(abi.F64, sol[0], sol[1])
    assert abs(worst[0] + 1e-4) > 1e-6 and rows["sufficient"] == [float(psd[0])]


@pytest.mark.gpu
def test_cpp_solve_batch_matches_oracle_per_instance(demo_out, oracle):
    for b in range(6):
        got = _parse_log(os.path.join(demo_out, "ilq_batch_%d.txt" % b))
        spec, ref = _oracle_solve(oracle, os.path.join(demo_out, "scene.txt"), got["x0"])
        _check_against_oracle(got, ref)


@pytest.mark.gpu
def test_cpp_augmented_lagrangian_solver_matches_oracle(demo_out, oracle):
    got = _parse_log(os.path.join(demo_out, "al_single.txt"))
    spec, ref = _oracle_solve(oracle, os.path.join(demo_out, "scene_constrained.txt"), got["x0"],
                              augmented_lagrangian=True)
    assert spec.num_constraints == 2
    _check_against_oracle(got, ref)


def _parse_lq(path):
    rows = {}
    order = []
    for line in open(path):
        tok = line.split()
        if tok[0] == "dims":
            n, N, mi, T = (int(v) for v in tok[1:])
        else:
            rows.setdefault(tok[0], []).append([float(v) for v in tok[1:]])
    return (n, N, mi, T), {k: np.array(v) for k, v in rows.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("open_loop", [False, True])
def test_cpp_lq_solvers_match_oracle(demo_out, oracle, open_loop):
    (n, N, mi, T), d = _parse_lq(os.path.join(demo_out, "lq_openloop.txt" if open_loop else "lq_feedback.txt"))
    m = N * mi
    dims = abi.make_dims(n, [mi] * N, T, 1, abi.F64)
    pairs = [(i, j) for i in range(N) for j in range(N)]
    A = d["A"].reshape(1, T, n * n)
    Bm = d["B"].reshape(1, T, n * m)
    Q = d["Q"].reshape(1, T, N, n * n)
    l = d["l"].reshape(1, T, N, n)
    R = d["R"].reshape(1, T, len(pairs) * mi * mi)
    r = d["r"].reshape(1, T, len(pairs) * mi)
    P, alpha, dx, co = oracle.lq_solve(dims, A, Bm, Q, l, R, r, pairs, x0=d["x0"].reshape(1, n), open_loop=open_loop,
                                       want_costates=True)
    # C++ side printed per-player (m_i x n) gains; restack to the (m x n) column-major layout
    gotP = d["P"].reshape(T, N, n, mi).transpose(0, 2, 1, 3).reshape(T, n * m)
    gotA = d["alpha"].reshape(T, m)
    tol = 2e-5
    assert np.max(np.abs(gotP - P[0])) < tol * max(1.0, np.max(np.abs(P)))
    assert np.max(np.abs(gotA - alpha[0])) < tol * max(1.0, np.max(np.abs(alpha)))
    assert np.max(np.abs(d["dx"] - dx[0])) < tol * max(1.0, np.max(np.abs(dx)))
    assert np.max(np.abs(d["costate"].reshape(T, N, n) - co[0])) < tol * max(1.0, np.max(np.abs(co)))
    assert np.max(np.abs(gotA)) > 1e-3 and np.max(np.abs(co)) > 1e-3


@pytest.mark.gpu
def test_cpp_lq_solver_properties_like_the_reference_suite():
    """tests/host/lq_solver_checks.cpp restates the properties of the reference's test/test_lq_solver.cpp
    (:292-434) against LQFeedbackSolver / LQOpenLoopSolver of the mirror, i.e. against the device kernels:
    Lyapunov iterations, feedback Nash by perturbation (with and without linear terms), single-player
    open loop = feedback."""
    exe = os.path.join(BIN, "lq_solver_checks")
    if not os.path.exists(exe):
        entry.build_host()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("PASS") == 4


def _shard_exe():
    exe = os.path.join(BIN, "shard_check")
    if not os.path.exists(exe):
        entry.build_host()
    return exe


def test_cpp_instance_range_is_the_sharding_rule():
    """host::InstanceRange (the cut GameSolver::SolveBatchSharded uses) against ilqgames_amd/sharding.py::instance_range
    (the cut bench.py uses): contiguous blocks, the first total % world ranks one instance longer."""
    from ilqgames_amd import sharding
    lines = subprocess.check_output([_shard_exe(), "range"], text=True, timeout=60).strip().splitlines()
    assert len(lines) == 6 * (1 + 2 + 3 + 8)
    for line in lines:
        total, world, rank, lo, hi = (int(v) for v in line.split())
        assert (lo, hi) == sharding.instance_range(total, rank, world), line


def test_cpp_two_process_rendezvous_and_blocks():
    """The bootstrap of the native multi-GPU entry on the CPU: two processes, RANK / WORLD_SIZE / ILQG_RENDEZVOUS_PORT
    from the environment (host::ShardFromEnvironment), rank 0 hands a 128-byte token — where ShardContext hands the
    ncclUniqueId — to rank 1 over TCP (host::RendezvousBroadcast), each prints its block of a 10-instance batch.  (The RCCL
    communicator and the all-gather themselves need GPUs: `shard_check solve`, tests below, runs them on a world of one.)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   ILQG_RENDEZVOUS_PORT=str(port))
        procs.append(subprocess.Popen([_shard_exe(), "rendezvous"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert outs[0].strip() == "rank 0 of 2 token ncclUniqueId-stand-in:0123456789abcdef block 0 5"
    assert outs[1].strip() == "rank 1 of 2 token ncclUniqueId-stand-in:0123456789abcdef block 5 10"


@pytest.mark.gpu
def test_cpp_solve_batch_sharded_on_a_world_of_one_equals_solve_batch():
    """GameSolver::SolveBatchSharded through ShardContext on one GPU: the block is the whole batch, the gather a
    device copy — every instance's operating point and gains must equal SolveBatch's bit for bit."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([_shard_exe(), "solve"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "largest difference between SolveBatch and SolveBatchSharded 0" in res.stdout, res.stdout


@pytest.mark.gpu
def test_cpp_solve_batch_sharded_on_two_gpus_equals_solve_batch():
    """The native multi-GPU entry on a world of TWO: two processes, one GPU each (host::ShardContext: hipSetDevice by
    LOCAL_RANK, ncclUniqueId over the TCP rendezvous, ncclCommInitRank), every rank solves its block of the batch and one
    ncclAllGather per result array gives each rank the whole BatchResult — which must equal SolveBatch's bit for bit on
    both ranks.  Needs two visible GPUs: skipped on the one-GPU boxes this build has had (no scaling curve has been
    measured), there so that the first multi-GPU lease runs the collective under a test, not under the bench."""
    import socket
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   ILQG_RENDEZVOUS_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([_shard_exe(), "solve"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d of 2" % r in o and "largest difference between SolveBatch and SolveBatchSharded 0" in o, o
