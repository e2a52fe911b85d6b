"""ctypes mirrors of the PODs in include/ilqg.h (the C-ABI drop-in boundary).

Python is only the test/bench harness here; the product is libilqg_hip.so and
the C++ mirror of the reference API under include/ilqgames/.
"""
import ctypes as C

MAX_PLAYERS = 8
MAX_XDIM = 32

F32, F64 = 0, 1

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE = 0, 1, 2, 3, 4

# ilqg_dyn_kind
ABI_VERSION = 8  # ILQG_ABI_VERSION of include/ilqg.h these mirrors were written against
DYN_UNICYCLE_4D, DYN_CAR_5D, DYN_CAR_6D = 1, 2, 3
DYN_UNICYCLE_4D_DISTURBED, DYN_PLANAR_DISTURBANCE = 4, 5  # the two rows of TwoPlayerUnicycle4D
DYN_DUBINS_CAR = 6  # (px, py, theta), u = (omega), param0 = speed
DYN_AIR_3D_EVADER, DYN_AIR_3D_PURSUER = 7, 8  # the two rows of Air3D (param0 = that aircraft's speed)
DYN_POINT_MASS_2D = 9  # (px, py, vx, vy), u = (ax, ay)
DYN_UNICYCLE_5D = 10  # (px, py, theta, v, s), u = (omega, a)
DYN_CAR_7D = 11  # (px, py, theta, phi, v, kappa, s), u = (omega, a), param0 = inter-axle distance
DYN_DELAYED_DUBINS_CAR = 12  # (px, py, theta, omega), u = (alpha), param0 = speed
# ilqg_cost_kind
(COST_QUADRATIC, COST_QUADRATIC_POLYLINE2, COST_SEMIQUADRATIC, COST_SEMIQUADRATIC_POLYLINE2,
 COST_PROXIMITY, COST_SIGNED_DISTANCE, COST_EXTREME_VALUE, CONSTRAINT_PROXIMITY,
 CONSTRAINT_SINGLE_DIMENSION) = range(1, 10)
COST_POLYLINE2_SIGNED_DISTANCE = 10
COST_QUADRATIC_DIFFERENCE = 11
COST_ORIENTATION = 12
COST_QUADRATIC_NORM = 13
COST_SEMIQUADRATIC_NORM = 14
COST_RELATIVE_DISTANCE = 15
COST_LOCALLY_CONVEX_PROXIMITY = 16
COST_CURVATURE = 17
CONSTRAINT_POLYLINE2_SIGNED_DISTANCE = 18
COST_NOMINAL_PATH_LENGTH = 19  # time-dependent: quadratic about k * dt * speed
COST_ROUTE_PROGRESS = 20       # time-dependent: quadratic about the polyline point at pos0 + k * dt * speed
COST_WEIGHTED_CONVEX_PROXIMITY = 21  # idx = (x1, y1, x2, y2), idx_extra = (v1, v2)
CONSTRAINT_AFFINE_SCALAR = 22  # g = a^T v - b over the whole argument vector; `polyline` = offset of [a | b] in dense_params
CONSTRAINT_AFFINE_VECTOR = 23  # g = |A v - b|; `polyline` = offset of [A (column-major) | b]
# ilqg_cost_role
ROLE_STATE_COST, ROLE_CONTROL_COST, ROLE_STATE_CONSTRAINT, ROLE_CONTROL_CONSTRAINT, ROLE_CHILD = range(5)
FLAG_ORIENTED, FLAG_IS_MIN, FLAG_EQUALITY = 1, 2, 4
# ILQG_SCHEDULE_* (ilqg_problem_last_schedule)
SCHEDULE_SINGLE_WAVE_SWEEP, SCHEDULE_ADJOINT_DECREASE, SCHEDULE_SPLIT_TRIAL, SCHEDULE_COMPACT_ROWS = 1, 2, 4, 8
SCHEDULE_COUNTED, SCHEDULE_GENERIC, SCHEDULE_OPEN_LOOP, SCHEDULE_STATIC_ROWS = 16, 32, 64, 128
SCHEDULE_PADDED_SWEEP = 256
SUM, MAX, MIN = 0, 1, 2


class Pair(C.Structure):
    _fields_ = [("i", C.c_int32), ("j", C.c_int32)]


class Dims(C.Structure):
    _fields_ = [("n", C.c_int32), ("num_players", C.c_int32), ("udim", C.c_int32 * MAX_PLAYERS),
                ("T", C.c_int32), ("batch", C.c_int32), ("dtype", C.c_int32),
                ("adaptive_regularization", C.c_int32), ("sweep_formulation", C.c_int32)]


CHOICE_AUTO, CHOICE_OFF, CHOICE_ON = 0, 1, 2
SWEEP_GENERIC = 3  # ilqg_dims::sweep_formulation: the run-time-dimensioned sweeps


class IterateLog(C.Structure):
    """ilqg_iterate_log (include/ilqg.h): device arrays [B][capacity][...] + count [B]."""
    _fields_ = [("xs", C.c_void_p), ("us", C.c_void_p), ("costs", C.c_void_p), ("P", C.c_void_p),
                ("alpha", C.c_void_p), ("count", C.c_void_p), ("capacity", C.c_int32)]


class SolveOptions(C.Structure):
    """ilqg_solve_options (include/ilqg.h)."""
    _fields_ = [("fixed_iters", C.c_int32), ("augmented_lagrangian", C.c_int32), ("resume", C.c_int32),
                ("deterministic", C.c_int32), ("active", C.c_void_p), ("forced_steps", C.c_void_p),
                ("split_trial", C.c_int32), ("handoff", C.c_int32), ("probe", C.c_int32), ("counted", C.c_int32),
                ("compact_rows", C.c_int32), ("round_bursts", C.c_int32), ("generic_kernels", C.c_int32),
                ("probe_first", C.c_int32), ("single_wave_sweep", C.c_int32), ("adjoint_expected_decrease", C.c_int32),
                ("static_rows", C.c_int32), ("padded_sweep", C.c_int32), ("probe_lanes", C.c_int32), ("reserved2", C.c_int32),
                ("iterate_log", C.POINTER(IterateLog)), ("max_runtime", C.c_double)]


class Subsystem(C.Structure):
    _fields_ = [("kind", C.c_int32), ("xdim", C.c_int32), ("udim", C.c_int32), ("param0", C.c_float)]


class CostTerm(C.Structure):
    _fields_ = [("kind", C.c_int32), ("role", C.c_int32), ("player", C.c_int32), ("arg", C.c_int32),
                ("idx", C.c_int32 * 4), ("idx_extra", C.c_int32 * 2), ("weight", C.c_float), ("value", C.c_float),
                ("value2", C.c_float),
                ("flags", C.c_int32),
                ("polyline", C.c_int32), ("child_begin", C.c_int32), ("child_count", C.c_int32),
                ("constraint_slot", C.c_int32), ("first_step", C.c_int32)]


class PlayerCost(C.Structure):
    _fields_ = [("state_regularization", C.c_float), ("control_regularization", C.c_float),
                ("structure", C.c_int32)]


class SolverParams(C.Structure):
    """SolverParams, include/ilqgames/solver/solver_params.h:50-84 (defaults as there)."""
    _fields_ = [("convergence_tolerance", C.c_float), ("max_solver_iters", C.c_int32),
                ("linesearch", C.c_int32), ("initial_alpha_scaling", C.c_float),
                ("geometric_alpha_scaling", C.c_float), ("max_backtracking_steps", C.c_int32),
                ("expected_decrease_fraction", C.c_float), ("open_loop", C.c_int32),
                ("unconstrained_solver_max_iters", C.c_int32), ("geometric_mu_scaling", C.c_float),
                ("geometric_mu_downscaling", C.c_float), ("geometric_lambda_downscaling", C.c_float),
                ("constraint_error_tolerance", C.c_float)]

    @staticmethod
    def default():
        return SolverParams(1e-1, 1000, 1, 0.5, 0.5, 10, 0.1, 0, 10, 1.1, 0.5, 0.5, 1e-1)


class ProblemDesc(C.Structure):
    _fields_ = [("num_players", C.c_int32), ("subsystems", Subsystem * MAX_PLAYERS),
                ("player_costs", PlayerCost * MAX_PLAYERS), ("num_terms", C.c_int32),
                ("terms", C.POINTER(CostTerm)), ("num_polylines", C.c_int32),
                ("polyline_offsets", C.POINTER(C.c_int32)), ("polyline_points", C.POINTER(C.c_float)),
                ("T", C.c_int32), ("dt", C.c_double), ("dtype", C.c_int32), ("params", SolverParams),
                ("num_dense_params", C.c_int32), ("dense_params", C.POINTER(C.c_float))]


def make_dims(n, udims, T, batch, dtype, adaptive_regularization=True):
    d = Dims()
    d.n, d.num_players, d.T, d.batch, d.dtype = n, len(udims), T, batch, dtype
    for i, u in enumerate(udims):
        d.udim[i] = u
    d.adaptive_regularization = 1 if adaptive_regularization else 0
    return d


def make_pairs(pairs):
    arr = (Pair * max(1, len(pairs)))()
    for q, (i, j) in enumerate(pairs):
        arr[q].i, arr[q].j = i, j
    return arr


class ProblemSpec:
    """Host-side builder of an ilqg_problem_desc; mirrors how a reference `Problem`
    subclass wires dynamics and PlayerCosts (include/ilqgames/solver/problem.h:66-73,136-148)."""

    def __init__(self, T=100, dt=0.1, params=None):
        self.subsystems = []   # (kind, xdim, udim, param0)
        self.player_costs = []  # (state_reg, control_reg, structure)
        self.terms = []
        self.polylines = []    # list of [(x, y), ...]
        self.T, self.dt = T, dt
        self.params = params or SolverParams.default()
        self.x0 = None
        self._num_constraints = 0
        self.dense_params = []  # coefficients of the affine constraints (ilqg_problem_desc::dense_params)

    # --- dynamics (ConcatenatedDynamicalSystem subsystem list) ---
    def add_player(self, kind, param0=0.0, state_reg=0.0, control_reg=0.0, structure=SUM):
        xdim = {DYN_UNICYCLE_4D: 4, DYN_CAR_5D: 5, DYN_CAR_6D: 6, DYN_UNICYCLE_4D_DISTURBED: 4,
                DYN_PLANAR_DISTURBANCE: 0, DYN_DUBINS_CAR: 3, DYN_AIR_3D_EVADER: 3, DYN_AIR_3D_PURSUER: 0,
                DYN_POINT_MASS_2D: 4, DYN_UNICYCLE_5D: 5, DYN_CAR_7D: 7, DYN_DELAYED_DUBINS_CAR: 4}[kind]
        one_control = kind in (DYN_DUBINS_CAR, DYN_AIR_3D_EVADER, DYN_AIR_3D_PURSUER, DYN_DELAYED_DUBINS_CAR)
        self.subsystems.append((kind, xdim, 1 if one_control else 2, param0))
        self.player_costs.append((state_reg, control_reg, structure))
        return len(self.subsystems) - 1

    @property
    def n(self):
        return sum(s[1] for s in self.subsystems)

    @property
    def udims(self):
        return [s[2] for s in self.subsystems]

    @property
    def m(self):
        return sum(self.udims)

    def xoff(self, i):
        return sum(s[1] for s in self.subsystems[:i])

    def add_polyline(self, pts):
        self.polylines.append([(float(x), float(y)) for x, y in pts])
        return len(self.polylines) - 1

    def _term(self, kind, role, player, arg=-1, idx=(0, 0, 0, 0), weight=1.0, value=0.0, flags=0,
              polyline=-1, child_begin=0, child_count=0, constraint=False, first_step=0, value2=0.0,
              idx_extra=(0, 0)):
        idx = tuple(idx) + (0,) * (4 - len(idx))
        slot = -1
        if constraint:
            slot = self._num_constraints
            self._num_constraints += 1
        self.terms.append(dict(kind=kind, role=role, player=player, arg=arg, idx=idx, weight=weight,
                               value=value, flags=flags, polyline=polyline, child_begin=child_begin,
                               child_count=child_count, constraint_slot=slot, first_step=first_step, value2=value2,
                               idx_extra=tuple(idx_extra)))
        return len(self.terms) - 1

    def final_time(self, threshold_time, term):
        """FinalTimeCost(cost, threshold_time) (cost/final_time_cost.h:55-88) around the term with index `term`
        (as returned by the cost methods): inactive while k * dt < threshold_time."""
        k = 0
        while float(k) * self.dt < threshold_time:
            k += 1
        self.terms[term]["first_step"] = k
        return term

    # --- PlayerCost::AddStateCost / AddControlCost / Add*Constraint with the reference cost ctors ---
    def quadratic(self, player, weight, dim, nominal=0.0, control_of=None):
        role, arg = (ROLE_STATE_COST, -1) if control_of is None else (ROLE_CONTROL_COST, control_of)
        return self._term(COST_QUADRATIC, role, player, arg, (dim,), weight, nominal)

    def semiquadratic(self, player, weight, dim, threshold, oriented_right, control_of=None):
        role, arg = (ROLE_STATE_COST, -1) if control_of is None else (ROLE_CONTROL_COST, control_of)
        return self._term(COST_SEMIQUADRATIC, role, player, arg, (dim,), weight, threshold,
                          FLAG_ORIENTED if oriented_right else 0)

    def quadratic_polyline2(self, player, weight, polyline, xy):
        return self._term(COST_QUADRATIC_POLYLINE2, ROLE_STATE_COST, player, -1, xy, weight, 0.0, 0, polyline)

    def semiquadratic_polyline2(self, player, weight, polyline, xy, threshold, oriented_right):
        return self._term(COST_SEMIQUADRATIC_POLYLINE2, ROLE_STATE_COST, player, -1, xy, weight, threshold,
                          FLAG_ORIENTED if oriented_right else 0, polyline)

    def proximity(self, player, weight, xy1, xy2, threshold):
        return self._term(COST_PROXIMITY, ROLE_STATE_COST, player, -1, tuple(xy1) + tuple(xy2), weight, threshold)

    def signed_distance(self, player, xy1, xy2, nominal=0.0, less_is_positive=True, role=ROLE_STATE_COST):
        return self._term(COST_SIGNED_DISTANCE, role, player, -1, tuple(xy1) + tuple(xy2), 1.0, nominal,
                          FLAG_ORIENTED if less_is_positive else 0)

    def polyline2_signed_distance(self, player, polyline, xy, nominal=0.0, oriented_same_as_polyline=True):
        return self._term(COST_POLYLINE2_SIGNED_DISTANCE, ROLE_STATE_COST, player, -1, xy, 1.0, nominal,
                          FLAG_ORIENTED if oriented_same_as_polyline else 0, polyline)

    def quadratic_difference(self, player, weight, dims1, dims2):
        return self._term(COST_QUADRATIC_DIFFERENCE, ROLE_STATE_COST, player, -1, tuple(dims1) + tuple(dims2), weight)

    # --- the rest of the reference's cost / constraint zoo that fits the device's scatter patterns ---
    def orientation(self, player, weight, dim, nominal=0.0):
        """OrientationCost (src/orientation_cost.cpp:50-80)."""
        return self._term(COST_ORIENTATION, ROLE_STATE_COST, player, -1, (dim,), weight, nominal)

    def quadratic_norm(self, player, weight, dims, nominal=0.0, control_of=None):
        """QuadraticNormCost (src/quadratic_norm_cost.cpp:50-94)."""
        role, arg = (ROLE_STATE_COST, -1) if control_of is None else (ROLE_CONTROL_COST, control_of)
        return self._term(COST_QUADRATIC_NORM, role, player, arg, tuple(dims), weight, nominal)

    def semiquadratic_norm(self, player, weight, dims, threshold, oriented_right, control_of=None):
        """SemiquadraticNormCost (src/semiquadratic_norm_cost.cpp:50-99)."""
        role, arg = (ROLE_STATE_COST, -1) if control_of is None else (ROLE_CONTROL_COST, control_of)
        return self._term(COST_SEMIQUADRATIC_NORM, role, player, arg, tuple(dims), weight, threshold,
                          FLAG_ORIENTED if oriented_right else 0)

    def relative_distance(self, player, weight, xy1, xy2):
        """RelativeDistanceCost (src/relative_distance_cost.cpp:50-104)."""
        return self._term(COST_RELATIVE_DISTANCE, ROLE_STATE_COST, player, -1, tuple(xy1) + tuple(xy2), weight)

    def locally_convex_proximity(self, player, weight, xy1, xy2, threshold):
        """LocallyConvexProximityCost (src/locally_convex_proximity_cost.cpp:50-108)."""
        return self._term(COST_LOCALLY_CONVEX_PROXIMITY, ROLE_STATE_COST, player, -1, tuple(xy1) + tuple(xy2), weight,
                          threshold)

    def curvature(self, player, weight, omega_idx, v_idx):
        """CurvatureCost (src/curvature_cost.cpp:50-86)."""
        return self._term(COST_CURVATURE, ROLE_STATE_COST, player, -1, (omega_idx, v_idx), weight)

    def polyline2_signed_distance_constraint(self, player, polyline, xy, threshold, keep_left):
        """Polyline2SignedDistanceConstraint (src/polyline2_signed_distance_constraint.cpp:52-144)."""
        return self._term(CONSTRAINT_POLYLINE2_SIGNED_DISTANCE, ROLE_STATE_CONSTRAINT, player, -1, xy, 1.0, threshold,
                          FLAG_ORIENTED if keep_left else 0, polyline, constraint=True)

    def nominal_path_length(self, player, weight, dim, nominal_speed):
        """NominalPathLengthCost (src/nominal_path_length_cost.cpp:50-78)."""
        return self._term(COST_NOMINAL_PATH_LENGTH, ROLE_STATE_COST, player, -1, (dim,), weight, nominal_speed)

    def route_progress(self, player, weight, nominal_speed, polyline, xy, initial_route_pos=0.0):
        """RouteProgressCost (src/route_progress_cost.cpp:52-110)."""
        return self._term(COST_ROUTE_PROGRESS, ROLE_STATE_COST, player, -1, xy, weight, nominal_speed, 0, polyline,
                          value2=initial_route_pos)

    def weighted_convex_proximity(self, player, weight, xy1, xy2, vidx1, vidx2, threshold):
        """WeightedConvexProximityCost (src/weighted_convex_proximity_cost.cpp:50-158)."""
        return self._term(COST_WEIGHTED_CONVEX_PROXIMITY, ROLE_STATE_COST, player, -1, tuple(xy1) + tuple(xy2), weight,
                          threshold, idx_extra=(vidx1, vidx2))

    def extreme_value(self, player, children, is_min):
        """children: list of callables(role) -> term index, created contiguously as CHILD terms."""
        begin = len(self.terms)
        for mk in children:
            mk(ROLE_CHILD)
        cnt = len(self.terms) - begin
        return self._term(COST_EXTREME_VALUE, ROLE_STATE_COST, player, -1, (0,), 1.0, 0.0,
                          FLAG_IS_MIN if is_min else 0, -1, begin, cnt)

    def proximity_constraint(self, player, xy1, xy2, threshold, keep_within):
        return self._term(CONSTRAINT_PROXIMITY, ROLE_STATE_CONSTRAINT, player, -1, tuple(xy1) + tuple(xy2), 1.0,
                          threshold, FLAG_ORIENTED if keep_within else 0, constraint=True)

    def single_dimension_constraint(self, player, dim, threshold, keep_below, control_of=None):
        role, arg = (ROLE_STATE_CONSTRAINT, -1) if control_of is None else (ROLE_CONTROL_CONSTRAINT, control_of)
        return self._term(CONSTRAINT_SINGLE_DIMENSION, role, player, arg, (dim,), 1.0, threshold,
                          FLAG_ORIENTED if keep_below else 0, constraint=True)

    def affine_scalar_constraint(self, player, a, b, is_equality=False, control_of=None):
        """AffineScalarConstraint(a, b, is_equality) (constraint/affine_scalar_constraint.h:54-100): a^T v - b on the whole
        state, or on the control vector of player `control_of`."""
        role, arg = (ROLE_STATE_CONSTRAINT, -1) if control_of is None else (ROLE_CONTROL_CONSTRAINT, control_of)
        dim = self.n if control_of is None else self.udims[control_of]
        a = [float(v) for v in a]
        assert len(a) == dim
        off = len(self.dense_params)
        self.dense_params += a + [float(b)]
        return self._term(CONSTRAINT_AFFINE_SCALAR, role, player, arg, (0,), 1.0, 0.0,
                          FLAG_EQUALITY if is_equality else 0, off, constraint=True)

    def affine_vector_constraint(self, player, A, b, is_equality=False, control_of=None):
        """AffineVectorConstraint(A, b, is_equality) (constraint/affine_vector_constraint.h:52-112), A square (its
        Quadraticize CHECKs input.size() == b.size()): |A v - b|."""
        import numpy as np
        role, arg = (ROLE_STATE_CONSTRAINT, -1) if control_of is None else (ROLE_CONTROL_CONSTRAINT, control_of)
        dim = self.n if control_of is None else self.udims[control_of]
        A = np.asarray(A, dtype=np.float32)
        assert A.shape == (dim, dim) and len(b) == dim
        off = len(self.dense_params)
        self.dense_params += [float(v) for v in A.T.reshape(-1)] + [float(v) for v in b]  # column-major
        return self._term(CONSTRAINT_AFFINE_VECTOR, role, player, arg, (0,), 1.0, 0.0,
                          FLAG_EQUALITY if is_equality else 0, off, constraint=True)

    @property
    def num_constraints(self):
        return self._num_constraints

    @staticmethod
    def from_dump(text):
        """Parse host::DumpDescription output (include/ilqgames/host/api.hpp) — the flattened form of a
        C++ `Problem` built through the mirrored reference API — back into a ProblemSpec."""
        spec = None
        subs, pcs = [], []
        for line in text.splitlines():
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "problem":
                spec = ProblemSpec(int(tok[2]), float(tok[3]))
            elif tok[0] == "subsystem":
                subs.append((int(tok[1]), int(tok[2]), int(tok[3]), float(tok[4])))
            elif tok[0] == "player_cost":
                pcs.append((float(tok[1]), float(tok[2]), int(tok[3])))
            elif tok[0] == "term":
                v = tok[1:]
                spec.terms.append(dict(kind=int(v[0]), role=int(v[1]), player=int(v[2]), arg=int(v[3]),
                                       idx=tuple(int(a) for a in v[4:8]), weight=float(v[8]), value=float(v[9]),
                                       flags=int(v[10]), polyline=int(v[11]), child_begin=int(v[12]),
                                       child_count=int(v[13]), constraint_slot=int(v[14]),
                                       first_step=int(v[15]) if len(v) > 15 else 0,
                                       value2=float(v[16]) if len(v) > 16 else 0.0,
                                       idx_extra=(int(v[17]), int(v[18])) if len(v) > 18 else (0, 0)))
                if int(v[14]) >= 0:
                    spec._num_constraints = max(spec._num_constraints, int(v[14]) + 1)
            elif tok[0] == "polyline":
                f = [float(a) for a in tok[1:]]
                spec.polylines.append(list(zip(f[0::2], f[1::2])))
            elif tok[0] == "params":
                v = tok[1:]
                spec.params = SolverParams(float(v[0]), int(v[1]), int(v[2]), float(v[3]), float(v[4]), int(v[5]),
                                           float(v[6]), int(v[7]), int(v[8]), float(v[9]), float(v[10]),
                                           float(v[11]), float(v[12]))
            elif tok[0] == "dense":
                spec.dense_params = [float(a) for a in tok[1:]]
            elif tok[0] == "x0":
                spec.x0 = [float(a) for a in tok[1:]]
        spec.subsystems, spec.player_costs = subs, pcs
        return spec

    def canonical(self):
        """Order-insensitive view for comparing two specs of the same problem: per (player, role) the
        terms in order (that order fixes the floating-point summation), polylines by content, constraint
        slots dropped (slot numbering only indexes the multiplier table)."""
        import numpy as np
        f32 = lambda v: float(np.float32(v))

        def term_key(t):
            # polyline by content, to 6 significant digits: libm's cosf/sinf and a correctly rounded
            # double->float cos/sin may differ in the last bit of a vertex
            if t["kind"] in (CONSTRAINT_AFFINE_SCALAR, CONSTRAINT_AFFINE_VECTOR):  # its coefficient block, by content
                dim = self.n if t["arg"] < 0 else self.udims[t["arg"]]
                cnt = dim + 1 if t["kind"] == CONSTRAINT_AFFINE_SCALAR else dim * dim + dim
                poly = tuple(f32(v) for v in self.dense_params[t["polyline"]:t["polyline"] + cnt])
            else:
                poly = tuple((float('%.6g' % x), float('%.6g' % y)) for x, y in self.polylines[t["polyline"]]) \
                    if t["polyline"] >= 0 else None
            kids = tuple(term_key(self.terms[c]) for c in range(t["child_begin"], t["child_begin"] + t["child_count"]))
            return (t["kind"], t["arg"], tuple(t["idx"]), f32(t["weight"]), f32(t["value"]), t["flags"], poly, kids,
                    t.get("first_step", 0), f32(t.get("value2", 0.0)), tuple(t.get("idx_extra", (0, 0))))

        groups = {}
        for t in self.terms:
            if t["role"] == ROLE_CHILD:
                continue
            groups.setdefault((t["player"], t["role"]), []).append(term_key(t))
        subs = [(k, xd, ud, f32(p0) if k in (DYN_CAR_5D, DYN_CAR_6D, DYN_DUBINS_CAR, DYN_AIR_3D_EVADER, DYN_AIR_3D_PURSUER, DYN_CAR_7D,
                                         DYN_DELAYED_DUBINS_CAR) else 0.0)
                for k, xd, ud, p0 in self.subsystems]
        pcs = [(f32(a), f32(b), c) for a, b, c in self.player_costs]
        return dict(T=self.T, dt=self.dt, subsystems=subs, player_costs=pcs, groups=groups, pairs=self.pairs())

    def pairs(self):
        """(i, j) control blocks in PlayerCost first-touch order (src/player_cost.cpp:59-86)."""
        out = []
        for i in range(len(self.subsystems)):
            for role in (ROLE_CONTROL_COST, ROLE_CONTROL_CONSTRAINT):
                for t in self.terms:
                    if t["player"] == i and t["role"] == role and (i, t["arg"]) not in out:
                        out.append((i, t["arg"]))
        return out

    def build(self, dtype):
        """Returns (ProblemDesc, keepalive) — keepalive owns the arrays the desc points into."""
        d = ProblemDesc()
        d.num_players = len(self.subsystems)
        for i, (kind, xdim, udim, p0) in enumerate(self.subsystems):
            d.subsystems[i] = Subsystem(kind, xdim, udim, p0)
            sr, cr, st = self.player_costs[i]
            d.player_costs[i] = PlayerCost(sr, cr, st)
        terms = (CostTerm * max(1, len(self.terms)))()
        for q, t in enumerate(self.terms):
            ct = terms[q]
            for k in ("kind", "role", "player", "arg", "weight", "value", "flags", "polyline", "child_begin",
                      "child_count", "constraint_slot"):
                setattr(ct, k, t[k])
            ct.first_step = t.get("first_step", 0)
            ct.value2 = t.get("value2", 0.0)
            for a in range(2):
                ct.idx_extra[a] = t.get("idx_extra", (0, 0))[a]
            for a in range(4):
                ct.idx[a] = t["idx"][a]
        d.num_terms = len(self.terms)
        d.terms = terms
        offs = [0]
        pts = []
        for pl in self.polylines:
            for x, y in pl:
                pts += [x, y]
            offs.append(offs[-1] + len(pl))
        offs_arr = (C.c_int32 * len(offs))(*offs)
        pts_arr = (C.c_float * max(1, len(pts)))(*pts)
        d.num_polylines = len(self.polylines)
        d.polyline_offsets = offs_arr
        d.polyline_points = pts_arr
        d.T, d.dt, d.dtype = self.T, self.dt, dtype
        d.params = self.params
        dense_arr = (C.c_float * max(1, len(self.dense_params)))(*self.dense_params)
        d.num_dense_params = len(self.dense_params)
        d.dense_params = dense_arr
        return d, (terms, offs_arr, pts_arr, dense_arr)
