/*
 * ilqg.h — C ABI of the MI355X-native iLQGames inner solver (libilqg_hip.so).
 *
 * This is the drop-in boundary for the hot path of HJReachability/ilqgames:
 * forward rollout + linearisation, per-player cost quadraticisation, the
 * coupled backward LQ Nash sweep, and the iterative-LQ loop around them,
 * batched over independent game instances.
 *
 * The reference has no FFI of its own; each entry point below replaces the
 * body of one reference C++ method (cited per function, paths relative to the
 * reference repo root).  The host-side C++ mirror of the reference classes
 * (include/ilqgames/...) packs its objects into the POD descriptors declared
 * here and calls these functions.  INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch / STL / Eigen types.
 *  - All array arguments are DEVICE pointers (HBM resident) unless the
 *    parameter name ends in `_host`.
 *  - Every small matrix is stored column-major (Eigen's default), so buffers
 *    can be handed back through `Strategy::Ps` etc. without reordering.
 *  - Arrays are trajectory-major per instance: [batch][T][...].
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - Return value: ilqg_status; no exceptions cross the boundary; programmer
 *    errors that glog CHECKs abort on in the reference (dimension mismatch,
 *    missing R_ii — src/lq_feedback_solver.cpp:77-78,139-140) are reported as
 *    ILQG_ERR_INVALID.  Algorithmic failure (line-search exhausted,
 *    src/ilq_solver.cpp:146-155) is reported per instance in `status[]`.
 */
#ifndef ILQG_H_
#define ILQG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILQG_MAX_PLAYERS 8
#define ILQG_MAX_XDIM 32
#define ILQG_MAX_UDIM_TOTAL 16
#define ILQG_MAX_PAIRS 64

typedef enum {
  ILQG_OK = 0,
  ILQG_ERR_INVALID = 1,     /* bad argument / dimension mismatch / missing R_ii */
  ILQG_ERR_UNSUPPORTED = 2, /* dims or model kind have no device kernel */
  ILQG_ERR_HIP = 3,         /* HIP runtime error (see ilqg_last_error) */
  ILQG_ERR_NO_DEVICE = 4    /* no gfx950 device visible */
} ilqg_status;

typedef enum { ILQG_F32 = 0, ILQG_F64 = 1 } ilqg_dtype;

/* (i, j): player i's cost carries a control block (R_ij, r_ij) on player j's
 * input — the keys of QuadraticCostApproximation::control
 * (include/ilqgames/utils/quadratic_cost_approximation.h:61-86). */
typedef struct {
  int32_t i;
  int32_t j;
} ilqg_pair;

/* Dimensions of one batched LQ game. */
typedef struct {
  int32_t n;                      /* state dimension                      */
  int32_t num_players;            /* N                                    */
  int32_t udim[ILQG_MAX_PLAYERS]; /* m_i                                  */
  int32_t T;                      /* number of time steps                 */
  int32_t batch;                  /* B independent instances              */
  int32_t dtype;                  /* ilqg_dtype                           */
  int32_t adaptive_regularization;/* Gershgorin step of lq_feedback_solver.cpp:163-176 */
  int32_t sweep_formulation;      /* ilqg_choice: ILQG_CHOICE_OFF selects the VALU / LDS formulation of the feedback sweep
                                     where the matrix-core one is the default (n <= 16); ILQG_SWEEP_GENERIC the
                                     run-time-dimensioned kernels every shape without a specialised instantiation runs
                                     on anyway; same results (A/B runs) */
} ilqg_dims;
#define ILQG_SWEEP_GENERIC 3

/* A scheduling choice that does not change results: let the library decide, or force it off / on. */
typedef enum { ILQG_CHOICE_AUTO = 0, ILQG_CHOICE_OFF = 1, ILQG_CHOICE_ON = 2 } ilqg_choice;

/* ------------------------------------------------------------------------ *
 *  LQ Nash sweeps                                                          *
 * ------------------------------------------------------------------------ */

/* Replaces LQFeedbackSolver::Solve (src/lq_feedback_solver.cpp:71-244).
 *
 *  A      [B][T][n*n]            lin.A
 *  Bm     [B][T][n*m]            [lin.Bs[0] | lin.Bs[1] | ...], m = sum m_i
 *  Q      [B][T][N][n*n]         quad[k][i].state.hess
 *  l      [B][T][N][n]           quad[k][i].state.grad
 *  R      [B][T][sum_p m_j^2]    quad[k][i].control[j].hess, pair order
 *  r      [B][T][sum_p m_j]      quad[k][i].control[j].grad, pair order
 *  pairs_host / npairs           which (i,j) blocks exist; (i,i) is mandatory
 *  x0     [B][n] or NULL (=0)    initial delta-x for the forward pass
 *  P      [B][T][m*n]            stacked gains, rows of player i at sum_{p<i} m_p
 *  alpha  [B][T][m]
 *  dx     [B][T][n] or NULL      delta_xs (forward pass, lq_feedback_solver.cpp:217-241)
 *  costates [B][T][N][n] or NULL  -Z_i[k+1] dx_k - zeta_i[k+1], zero at T-1 (lq_feedback_solver.cpp:223-227);
 *                                needs dx (the reference CHECKs the pair, :77-78): ILQG_ERR_INVALID otherwise
 *  Entry T-1 of P/alpha is written as zero (strategy.h:64-70; loop starts at T-2).
 *  Any dimensions within ILQG_MAX_XDIM / ILQG_MAX_PLAYERS / ILQG_MAX_UDIM_TOTAL run, players with different control
 *  dimensions included (src/lq_feedback_solver.cpp:118-160 walks cumulative dimensions): the shapes of the reference's
 *  examples have specialised kernels (matrix cores, LDS-DMA staging), everything else the run-time-dimensioned ones.
 */
ilqg_status ilqg_lq_feedback_batch(const ilqg_dims* d, const void* A,
                                   const void* Bm, const void* Q, const void* l,
                                   const void* R, const void* r,
                                   const ilqg_pair* pairs_host, int32_t npairs,
                                   const void* x0, void* P, void* alpha,
                                   void* dx, void* costates, void* stream);

/* Replaces LQOpenLoopSolver::Solve (src/lq_open_loop_solver.cpp:73-195).
 * Same inputs; P is written as zero, alpha/dx as the reference; costates [B][T][N][n] or NULL:
 * A_k^T (M_i[k+1] x_{k+1} + m_i[k+1]), zero at T-1 (:171-176, :191), dx required with it (:83-84). */
ilqg_status ilqg_lq_openloop_batch(const ilqg_dims* d, const void* A,
                                   const void* Bm, const void* Q, const void* l,
                                   const void* R, const void* r,
                                   const ilqg_pair* pairs_host, int32_t npairs,
                                   const void* x0, void* P, void* alpha,
                                   void* dx, void* costates, void* stream);

/* ------------------------------------------------------------------------ *
 *  Problem descriptor (what Problem::Initialize builds, flattened to PODs)  *
 * ------------------------------------------------------------------------ */

typedef enum {
  ILQG_DYN_UNICYCLE_4D = 1, /* include/ilqgames/dynamics/single_player_unicycle_4d.h:90-116 */
  ILQG_DYN_CAR_5D = 2,      /* include/ilqgames/dynamics/single_player_car_5d.h:100-133     */
  ILQG_DYN_CAR_6D = 3,      /* include/ilqgames/dynamics/single_player_car_6d.h:102-138     */
  /* TwoPlayerUnicycle4D (include/ilqgames/dynamics/two_player_unicycle_4d.h:57-139): ONE 4-state
   * unicycle driven by two players.  Player 0's row is the unicycle with u = (omega, a)
   * (xdim 4, udim 2); player 1's row is a planar disturbance u = (dx, dy) added to the position
   * rates of the row before it (xdim 0, udim 2).  The two kinds only occur as this pair. */
  ILQG_DYN_UNICYCLE_4D_DISTURBED = 4,
  ILQG_DYN_PLANAR_DISTURBANCE = 5,
  ILQG_DYN_DUBINS_CAR = 6, /* include/ilqgames/dynamics/single_player_dubins_car.h:57-120: x = (px, py, theta),
                             u = (omega), constant speed param0; xdim 3, udim 1 */
  /* Air3D (include/ilqgames/dynamics/air_3d.h:64-157): the pursuer's pose relative to the evader, x = (rx, ry,
   * rtheta), driven by the evader's turn rate (player 0) and the pursuer's (player 1).  Row pair like kinds 4/5:
   * the evader row carries the state (xdim 3, udim 1, param0 = evader speed), the pursuer row none (xdim 0,
   * udim 1, param0 = pursuer speed). */
  ILQG_DYN_AIR_3D_EVADER = 7,
  ILQG_DYN_AIR_3D_PURSUER = 8,
  /* include/ilqgames/dynamics/single_player_point_mass_2d.h:55-110: x = (px, py, vx, vy), u = (ax, ay);
   * xdim 4, udim 2.  Point masses only occur in games made of point masses. */
  ILQG_DYN_POINT_MASS_2D = 9,
  /* The rest of the reference's single-player models.  They run on the plain RK4 (one lane walks the four stages of
   * a block) instead of the stage-per-lane integrator of kinds 1-3 and 6, in the instantiations that list them
   * (dims_use_plain_rk4 in csrc/ilqg_stages.hpp). */
  ILQG_DYN_UNICYCLE_5D = 10, /* include/ilqgames/dynamics/single_player_unicycle_5d.h:55-135: x = (px, py, theta, v, s),
                                u = (omega, a); s = path length; xdim 5, udim 2 */
  ILQG_DYN_CAR_7D = 11,      /* include/ilqgames/dynamics/single_player_car_7d.h:61-170: x = (px, py, theta, phi, v,
                                kappa, s), u = (omega, a), param0 = inter-axle distance; xdim 7, udim 2 */
  ILQG_DYN_DELAYED_DUBINS_CAR = 12 /* include/ilqgames/dynamics/single_player_delayed_dubins_car.h:57-129: x = (px, py,
                                theta, omega), u = (alpha), constant speed param0; xdim 4, udim 1 */
} ilqg_dyn_kind;

/* One block of a ConcatenatedDynamicalSystem
 * (src/concatenated_dynamical_system.cpp:52-107); player i owns subsystem i.
 * (TwoPlayerUnicycle4D is written as two rows, see ilqg_dyn_kind.) */
typedef struct {
  int32_t kind; /* ilqg_dyn_kind */
  int32_t xdim;
  int32_t udim;
  float param0; /* inter-axle distance for the car models; the speed of the Dubins car */
} ilqg_subsystem;

typedef enum {
  ILQG_COST_QUADRATIC = 1,           /* src/quadratic_cost.cpp:51-94             */
  ILQG_COST_QUADRATIC_POLYLINE2 = 2, /* src/quadratic_polyline2_cost.cpp:52-126  */
  ILQG_COST_SEMIQUADRATIC = 3,       /* src/semiquadratic_cost.cpp:51-85         */
  ILQG_COST_SEMIQUADRATIC_POLYLINE2 = 4, /* src/semiquadratic_polyline2_cost.cpp:52-142 */
  ILQG_COST_PROXIMITY = 5,           /* src/proximity_cost.cpp:52-122            */
  ILQG_COST_SIGNED_DISTANCE = 6,     /* src/signed_distance_cost.cpp:51-113      */
  ILQG_COST_EXTREME_VALUE = 7,       /* src/extreme_value_cost.cpp:51-85         */
  ILQG_CONSTRAINT_PROXIMITY = 8,     /* src/proximity_constraint.cpp:56-116      */
  ILQG_CONSTRAINT_SINGLE_DIMENSION = 9, /* include/ilqgames/constraint/single_dimension_constraint.h:57-103 */
  ILQG_COST_POLYLINE2_SIGNED_DISTANCE = 10, /* src/polyline2_signed_distance_cost.cpp:52-126: idx = (xidx, yidx),
                                              value = nominal, ORIENTED = oriented_same_as_polyline, weight unused */
  ILQG_COST_QUADRATIC_DIFFERENCE = 11, /* src/quadratic_difference_cost.cpp:51-91 with two dimension pairs:
                                         idx = (dims1[0], dims1[1], dims2[0], dims2[1]) */
  ILQG_COST_ORIENTATION = 12,        /* src/orientation_cost.cpp:50-80: idx[0] = heading dimension, value = nominal;
                                        0.5 w wrap(x - nominal)^2 with the difference wrapped into [-pi, pi) */
  ILQG_COST_QUADRATIC_NORM = 13,     /* src/quadratic_norm_cost.cpp:50-94: idx = (dim1, dim2), value = nominal;
                                        0.5 w (|(x[dim1], x[dim2])| - nominal)^2 */
  ILQG_COST_SEMIQUADRATIC_NORM = 14, /* src/semiquadratic_norm_cost.cpp:50-99: the one-sided form; value = threshold,
                                        ORIENTED = oriented_right */
  ILQG_COST_RELATIVE_DISTANCE = 15,  /* src/relative_distance_cost.cpp:50-104: idx = (x1, y1, x2, y2); w |p1 - p2| */
  ILQG_COST_LOCALLY_CONVEX_PROXIMITY = 16, /* src/locally_convex_proximity_cost.cpp:50-108: idx = (x1, y1, x2, y2),
                                        value = threshold; 0.5 w min((thr - |dx|)^2, (thr - |dy|)^2) inside the box */
  ILQG_COST_CURVATURE = 17,          /* src/curvature_cost.cpp:50-86: idx = (omega index, v index); 0.5 w (omega / v)^2 */
  ILQG_CONSTRAINT_POLYLINE2_SIGNED_DISTANCE = 18, /* src/polyline2_signed_distance_constraint.cpp:52-144: idx = (x, y),
                                        polyline, value = threshold, ORIENTED = keep_left */
  /* The two costs that depend on time.  ILQSolver hands them the time RELATIVE to the start of the window
   * (src/ilq_solver.cpp:186,236: RelativeTime(kk) = kk * dt), so each is a quadratic about a per-step nominal that
   * ilqg_problem_create tabulates.  RouteProgressCost subtracts RelativeTimeTracker's initial time from that
   * relative time (src/route_progress_cost.cpp:58), which is 0 until a receding-horizon step resets it: the tables
   * are those of a first solve. */
  ILQG_COST_NOMINAL_PATH_LENGTH = 19, /* src/nominal_path_length_cost.cpp:50-78: idx[0] = dimension, value =
                                        nominal speed; 0.5 w (x[dim] - t * speed)^2 */
  ILQG_COST_ROUTE_PROGRESS = 20,     /* src/route_progress_cost.cpp:52-110: idx = (x, y), polyline, value = nominal
                                        speed, value2 = initial route position; 0.5 w |p - PointAt(pos0 + t speed)|^2 */
  ILQG_COST_WEIGHTED_CONVEX_PROXIMITY = 21, /* src/weighted_convex_proximity_cost.cpp:50-158: idx = (x1, y1, x2, y2),
                                        idx_extra = (v1, v2), value = threshold; LOCALLY_CONVEX_PROXIMITY scaled by
                                        v1^2 + v2^2, derivatives as written there.  Top-level state cost only. */
  /* The two constraints on the WHOLE argument vector (the state, or the control vector of player `arg`; d = its
   * dimension): dense d x d Hessian blocks.  Their coefficients sit in ilqg_problem_desc::dense_params, the term's
   * `polyline` field is the offset of its block there. */
  ILQG_CONSTRAINT_AFFINE_SCALAR = 22, /* include/ilqgames/constraint/affine_scalar_constraint.h:54-100: g = a^T v - b;
                                        block [a (d) | b] */
  ILQG_CONSTRAINT_AFFINE_VECTOR = 23  /* include/ilqgames/constraint/affine_vector_constraint.h:52-112: g = |A v - b| with
                                        a square A; block [A (d x d, column-major) | b (d)]; derivatives as written
                                        there (its Hessian mixes A A^T and A^T A) */
} ilqg_cost_kind;

/* Where a term sits inside PlayerCost::Quadraticize (src/player_cost.cpp:194-215):
 * state costs, control costs, state constraints, control constraints — in that
 * order; CHILD terms are only reached through an EXTREME_VALUE parent. */
typedef enum {
  ILQG_ROLE_STATE_COST = 0,
  ILQG_ROLE_CONTROL_COST = 1,
  ILQG_ROLE_STATE_CONSTRAINT = 2,
  ILQG_ROLE_CONTROL_CONSTRAINT = 3,
  ILQG_ROLE_CHILD = 4
} ilqg_cost_role;

#define ILQG_FLAG_ORIENTED 1 /* oriented_right / keep_within / keep_below / less_is_positive */
#define ILQG_FLAG_IS_MIN 2   /* ExtremeValueCost::is_min_ */
#define ILQG_FLAG_EQUALITY 4 /* Constraint::is_equality_ (constraint.h:74-76,98-117): no inactive-constraint gate on mu, the
                                multiplier is not clipped at zero.  The affine constraints take it; the reference's other
                                constraints are inequalities by construction. */

typedef struct {
  int32_t kind;        /* ilqg_cost_kind                                       */
  int32_t role;        /* ilqg_cost_role                                       */
  int32_t player;      /* owning PlayerCost                                    */
  int32_t arg;         /* control costs/constraints: which player's u; else -1 */
  int32_t idx[4];      /* QUADRATIC/SEMIQUADRATIC/SINGLE_DIM: idx[0]=dimension (-1 = all dims);
                          *_POLYLINE2: (xidx, yidx); PROXIMITY/SIGNED_DISTANCE/
                          CONSTRAINT_PROXIMITY: (x1, y1, x2, y2)                */
  int32_t idx_extra[2]; /* WEIGHTED_CONVEX_PROXIMITY: (v1, v2); else 0          */
  float weight;        /* Cost::weight_                                        */
  float value;         /* nominal_ or threshold_                               */
  float value2;        /* ROUTE_PROGRESS: initial_route_pos_; else 0            */
  int32_t flags;       /* ILQG_FLAG_*                                          */
  int32_t polyline;    /* index into the polyline table or -1                  */
  int32_t child_begin; /* EXTREME_VALUE: first child term index                */
  int32_t child_count; /* EXTREME_VALUE: number of children                    */
  int32_t constraint_slot; /* constraints: index into the per-instance lambda table, else -1 */
  int32_t first_step;  /* FinalTimeCost (include/ilqgames/cost/final_time_cost.h:55-88): the term is zero (value
                          and derivatives) before this time step — the smallest k with k * dt >= threshold_time,
                          times taken relative to the start of the window.  0 = always active. */
} ilqg_cost_term;

typedef enum { ILQG_SUM = 0, ILQG_MAX = 1, ILQG_MIN = 2 } ilqg_cost_structure;

/* PlayerCost ctor arguments + cost structure (include/ilqgames/cost/player_cost.h:63-109). */
typedef struct {
  float state_regularization;
  float control_regularization;
  int32_t structure; /* ilqg_cost_structure */
} ilqg_player_cost;

/* SolverParams (include/ilqgames/solver/solver_params.h:50-84). */
typedef struct {
  float convergence_tolerance;
  int32_t max_solver_iters;
  int32_t linesearch;
  float initial_alpha_scaling;
  float geometric_alpha_scaling;
  int32_t max_backtracking_steps;
  float expected_decrease_fraction;
  int32_t open_loop;
  int32_t unconstrained_solver_max_iters;
  float geometric_mu_scaling;
  float geometric_mu_downscaling;
  float geometric_lambda_downscaling;
  float constraint_error_tolerance;
} ilqg_solver_params;

typedef struct {
  int32_t num_players;
  ilqg_subsystem subsystems[ILQG_MAX_PLAYERS];
  ilqg_player_cost player_costs[ILQG_MAX_PLAYERS];
  int32_t num_terms;
  const ilqg_cost_term* terms;     /* host array                             */
  int32_t num_polylines;
  const int32_t* polyline_offsets; /* host, num_polylines+1, in points       */
  const float* polyline_points;    /* host, 2 floats per point               */
  int32_t T;                       /* time::kNumTimeSteps (types.h:141-142)  */
  double dt;                       /* time::kTimeStep     (types.h:135)      */
  int32_t dtype;                   /* ilqg_dtype of state/gain arithmetic    */
  ilqg_solver_params params;
  int32_t num_dense_params;        /* coefficients of the affine constraints  */
  const float* dense_params;       /* host array (may be NULL when there are none) */
} ilqg_problem_desc;

void ilqg_default_solver_params(ilqg_solver_params* p);

typedef struct ilqg_problem ilqg_problem;

/* Builds the device-side tables of one Problem (what Problem::Initialize +
 * ILQSolver::ILQSolver set up, include/ilqgames/solver/problem.h:66-73,
 * include/ilqgames/solver/ilq_solver.h:69-95).
 * A handle also owns the four round counters its solves report through (device + pinned host copy, allocated
 * here): like the reference's solver objects it serves one solve at a time — concurrent solves need one handle
 * each (the read-only entry points, e.g. rollout / linearize / quadraticize / strategy costs, may share one). */
ilqg_status ilqg_problem_create(const ilqg_problem_desc* desc, ilqg_problem** out);
void ilqg_problem_destroy(ilqg_problem* p);

/* Bytes of device workspace a solve of `batch` instances needs: the per-instance iterates, linearisations and
 * loop states, the lists of back-tracking instances and the pool of the speculative line search.  A solve
 * allocates nothing. */
ilqg_status ilqg_workspace_bytes(const ilqg_problem* p, int32_t batch, uint64_t* bytes);

/* ------------------------------------------------------------------------ *
 *  Hot-path stages (each also usable on its own; all batched)               *
 * ------------------------------------------------------------------------ */

/* Replaces ILQSolver::CurrentOperatingPoint (src/ilq_solver.cpp:174-206):
 * closed-loop rollout from x0 under u_i = u_ref_i - P_i (x - x_ref) - alpha_i
 * with the RK4(2 sub-steps) integrator of
 * src/multi_player_dynamical_system.cpp:52-77.
 *  x0 [B][n]; xs_ref [B][T][n]; us_ref [B][T][m]; P [B][T][m*n]; alpha [B][T][m]
 *  alpha_scale [B] or NULL (=1): step size applied to alpha on the fly (the
 *  reference scales alphas destructively, ilq_solver.cpp:66-72,314,339)
 *  xs [B][T][n], us [B][T][m] outputs.  active [B] int32 mask or NULL. */
ilqg_status ilqg_rollout_batch(const ilqg_problem* p, int32_t batch,
                               const void* x0, const void* xs_ref,
                               const void* us_ref, const void* P,
                               const void* alpha, const void* alpha_scale,
                               void* xs, void* us, const int32_t* active,
                               void* stream);

/* Replaces ILQSolver::ComputeLinearization (src/ilq_solver.cpp:437-455) ->
 * ConcatenatedDynamicalSystem::Linearize (src/concatenated_dynamical_system.cpp:86-107).
 *  A [B][T][n*n], Bm [B][T][n*m].
 * The stage entry points (this one, ilqg_quadraticize_batch, ilqg_total_costs_batch, ilqg_rollout_batch) run the same
 * compile-time-dimensioned kernels as the solves: a problem whose (n, N, m_i) is not among the library's
 * instantiations, or whose players' control dimensions differ, gets ILQG_ERR_UNSUPPORTED here too. */
ilqg_status ilqg_linearize_batch(const ilqg_problem* p, int32_t batch,
                                 const void* xs, const void* us, void* A,
                                 void* Bm, const int32_t* active, void* stream);

/* Replaces ILQSolver::ComputeCostQuadraticization (src/ilq_solver.cpp:471-490)
 * -> PlayerCost::Quadraticize (src/player_cost.cpp:194-225).
 *  lambdas [B][num_constraints][T] or NULL; mu [B] or NULL (constraint.h:98-117)
 *  t_extreme [B][N] int32 or NULL: time of extreme cost for MAX/MIN players.
 *  Q,l,R,r: layouts of ilqg_lq_feedback_batch, pair order = ilqg_problem_pairs. */
ilqg_status ilqg_quadraticize_batch(const ilqg_problem* p, int32_t batch,
                                    const void* xs, const void* us,
                                    const void* lambdas, const void* mu,
                                    const int32_t* t_extreme, void* Q, void* l,
                                    void* R, void* r, const int32_t* active,
                                    void* stream);

/* The (i,j) control blocks PlayerCost::Quadraticize creates for this problem. */
ilqg_status ilqg_problem_pairs(const ilqg_problem* p, ilqg_pair* pairs_host,
                               int32_t* npairs);

/* Replaces ILQSolver::TotalCosts (src/ilq_solver.cpp:220-257).
 *  costs [B][N]; t_extreme [B][N] int32 (written for MAX/MIN players). */
ilqg_status ilqg_total_costs_batch(const ilqg_problem* p, int32_t batch,
                                   const void* xs, const void* us, void* costs,
                                   int32_t* t_extreme, const int32_t* active,
                                   void* stream);

/* ------------------------------------------------------------------------ *
 *  Whole iterative-LQ solve                                                 *
 * ------------------------------------------------------------------------ */

/* Replaces ILQSolver::Solve (src/ilq_solver.cpp:76-172) for `batch`
 * independent instances that share one Problem definition and differ in x0
 * (and warm start).
 *  x0        [B][n]        Problem::InitialState per instance
 *  xs, us    [B][T][n|m]   in: warm-start operating point; out: final one
 *  P, alpha  [B][T][..]    in: warm-start strategies; out: final strategies
 *                          (alpha carries the accepted step scaling, as the
 *                          strategies the reference logs do)
 *  total_costs [B][N]      ILQSolver::TotalCosts of the final iterate
 *  iters     [B] int32     outer iterations performed
 *  status    [B] int32     1 = success flag true (src/ilq_solver.cpp:169),
 *                          0 = line-search failure (:146-155)
 *  converged [B] int32     has_converged at exit
 *  workspace               device scratch of ilqg_workspace_bytes() bytes
 *  fixed_iters > 0 runs exactly that many outer iterations per instance
 *  ignoring convergence (throughput benchmarking); 0 = reference semantics. */
ilqg_status ilqg_ilq_solve_batch(ilqg_problem* p, int32_t batch, const void* x0,
                                 void* xs, void* us, void* P, void* alpha,
                                 void* total_costs, int32_t* iters,
                                 int32_t* status, int32_t* converged,
                                 void* workspace, int32_t fixed_iters,
                                 void* stream);

/* Everything a solve call can be told beyond its buffers (ilqg_solve_batch_ex).  The scheduling choices select among
 * device schedules of the SAME arithmetic: split_trial / handoff / probe / counted / compact_rows / round_bursts are
 * bit-identical with each other; single_wave_sweep and adjoint_expected_decrease run the same recursion as a different
 * instruction stream (sums in another order), i.e. the same results to rounding — and ILQG_CHOICE_AUTO picks them from
 * the batch size (five or more instances per CU), so an instance's last bits, and with them a line-search decision that
 * sits on a rounding threshold, can depend on the size of the batch it is solved in.  Set `deterministic` (or pin the two
 * choices ON / OFF) for results that must not; ilqg_problem_last_schedule says what the last solve ran.
 * single_wave_sweep = ON is a request: shapes the single-wave form is not built for (players with one control, n >= 16,
 * compact rows wider than its staging row) run the player-parallel sweep, and ilqg_problem_last_schedule reports it.
 * ilqg_default_solve_options fills the defaults: reference semantics, every choice ILQG_CHOICE_AUTO. */
typedef struct {
  int32_t fixed_iters;          /* > 0: exactly that many outer iterations per instance, convergence not tested   */
  int32_t augmented_lagrangian; /* 1: AugmentedLagrangianSolver::Solve around the inner solves                     */
  int32_t resume;               /* 1: GameSolver::Solve called again on the same solver object: the workspace
                                      holds the previous call's state (last_merit_function_value_)               */
  int32_t deterministic;        /* 1: every ILQG_CHOICE_AUTO below resolves to a schedule that does not depend on the
                                      batch size (the player-parallel sweep and its forward-pass expected decrease at
                                      every batch): an instance's result is the same bits whatever batch it is solved in
                                      (tests/test_gpu_single_wave.py).  Costs the large-batch throughput form.  (This
                                      field was reserved0 = 0 up to ABI 6: same layout.)                              */
  const int32_t* active;        /* [B] device int32 or NULL: instances with 0 are skipped, their buffers untouched.
                                     A free-running solve (which waits for the device every round anyway) counts the
                                     mask first and lets every ILQG_CHOICE_AUTO below go by the instances taking part,
                                     not by B; fixed_iters > 0 without the augmented Lagrangian stays asynchronous and
                                     goes by B.                                                                        */
  const void* forced_steps;     /* [B][fixed_iters] device (problem dtype) or NULL.  Test mode: iteration q of
                                      instance b scales its strategies by forced_steps[b][q], integrates and
                                      quadraticises once and ACCEPTS, whatever CheckArmijoCondition says — the
                                      loop's only data-dependent branch is gone, so iterates can be compared one
                                      by one with a CPU run given the same steps.  Needs fixed_iters > 0, no AL.   */
  int32_t split_trial;          /* ilqg_choice: the trial pass as three launches (integrate / rows / decide)        */
  int32_t handoff;              /* ilqg_choice: back-tracking instances leave the fused kernel for split passes     */
  int32_t probe;                /* ilqg_choice: speculative line search over the next step sizes of listed instances */
  int32_t counted;              /* ilqg_choice: host counts the rounds of a fixed-iteration solve as well           */
  int32_t compact_rows;         /* ilqg_choice: the row stage hands the sweep only the touched words of [Q|l|R|r]   */
  int32_t round_bursts;         /* ilqg_choice: free-running solves read their counters back once per burst of rounds */
  int32_t generic_kernels;      /* ilqg_choice: ILQG_CHOICE_ON runs the run-time-dimensioned kernels (what every shape
                                     without a specialised instantiation runs on) for this problem too */
  int32_t probe_first;          /* speculative line search: step sizes probed per listed instance in the first round of
                                     a tail (doubling every round up to 128, as many as the pool holds); 0 = the library's
                                     choice.  Same decisions whatever the value.                                       */
  int32_t single_wave_sweep;    /* ilqg_choice: the one-tile feedback sweep with one wave per instance (twice the instances
                                     per CU; the library picks it for batches of five or more instances per CU)           */
  int32_t adjoint_expected_decrease; /* ilqg_choice: the single-wave sweep forms ILQSolver::ExpectedDecrease itself, by an
                                     adjoint recursion inside the sweep, instead of leaving scratch rows for a forward pass
                                     in the next trial pass (AUTO: where the trial pass is split and has no wave to spare
                                     for one).  Same value up to the order of summation.                                */
  int32_t static_rows;          /* ilqg_choice: a problem whose row program matches a registered structure
                                     (ilqg_problem_row_program) runs the linearise / quadraticise stage of the fused trial
                                     kernel as straight-line code compiled for that structure instead of interpreting the
                                     program (AUTO: on).  Bit-identical.                                                */
  int32_t padded_sweep;         /* ilqg_choice: a solve on the run-time-dimensioned kernels runs its Riccati sweep on the
                                     specialised sweep of the smallest instantiated shape the game embeds in (same player
                                     count, at least its states and its widest control; added states are inert, added
                                     controls have a zero column of B and a unit entry of R_ii and solve to exact zeros).
                                     AUTO: on for problems without an instantiation of their own; ON where no shape holds
                                     the game is ILQG_ERR_UNSUPPORTED.  The same recursion on another elimination order:
                                     results to rounding.  (This field was reserved1 = 0 up to ABI 7: same layout.)        */
  int32_t probe_lanes;          /* ilqg_choice: the probing rollouts of the speculative line search with a lane per
                                     (candidate, subsystem) — 64 / N candidates of an instance per wavefront, the RK4 stages
                                     of a step one after the other in the lane — instead of two candidates per wavefront
                                     with a lane per stage: an eighth of the instructions per rollout, a longer chain per
                                     step (AUTO: rounds with more rollouts than the chip holds at once, eight or more
                                     candidates per instance).  Bit-identical trajectories.  (ABI 8)                      */
  int32_t reserved2;
  const struct ilqg_iterate_log* iterate_log; /* NULL, or where every logged iterate of the solve goes (below)          */
  double max_runtime;           /* > 0: the anytime exit of ILQSolver::Solve (src/ilq_solver.cpp:123-124) on the host's
                                     clock, seconds — once it has passed, instances leave the loop at their next
                                     iteration boundary with success = 1 and the iterate they hold (a line search in
                                     flight is finished first, as there).  <= 0: run to the iteration bounds.  Needs a
                                     free-running solve (fixed_iters = 0, no forced steps).  With the augmented-Lagrangian
                                     loop (src/augmented_lagrangian_solver.cpp:85-110,193) every inner solve of a constrained
                                     problem gets max_runtime / max_solver_iters, and the outer loop has its own clock: it
                                     starts at that allowance, adds the wall time of every outer iteration (its own
                                     LoopTimer, kept in the problem handle) and starts no further inner solve once
                                     elapsed >= max_runtime - RuntimeUpperBound() — a batch shares the clock. */
} ilqg_solve_options;

/* What SolverLog::AddSolverIterate receives per iterate (include/ilqgames/utils/solver_log.h:65-74: the reference deep-
 * copies operating point, strategies and total costs at src/ilq_solver.cpp:111 and :164).  On the device the copies are
 * optional: a solve given a log writes iterate q of instance b — q = 0 is the initial rollout, then one per accepted
 * step; with the augmented-Lagrangian loop the iterates of successive inner solves follow one another as in
 * src/augmented_lagrangian_solver.cpp:94,185 — to slot q of that instance while q < capacity, and counts them.
 * All arrays are device memory in the problem's dtype. */
typedef struct ilqg_iterate_log {
  void* xs;       /* [B][capacity][T][n]                                                         */
  void* us;       /* [B][capacity][T][m]                                                         */
  void* costs;    /* [B][capacity][N]     ILQSolver::TotalCosts of the iterate                    */
  void* P;        /* [B][capacity][T][m*n] or NULL: the strategies the iterate was played with    */
  void* alpha;    /* [B][capacity][T][m] or NULL (alpha carries the accepted step, as logged)     */
  int32_t* count; /* [B] out: iterates the solve produced (may exceed capacity: those were dropped) */
  int32_t capacity;
} ilqg_iterate_log;
void ilqg_default_solve_options(ilqg_solve_options* o);

/* ilqg_ilq_solve_batch / ilqg_al_solve_batch / ilqg_solve_again_batch are this call with the matching options. */
ilqg_status ilqg_solve_batch_ex(ilqg_problem* p, int32_t batch, const void* x0, void* xs, void* us, void* P,
                                void* alpha, void* total_costs, int32_t* iters, int32_t* status, int32_t* converged,
                                void* workspace, const ilqg_solve_options* options, void* stream);

/* The loop state a solve left in its workspace, per instance (what ILQSolver keeps in members / locals):
 * last_merit_function_value_ (ilq_solver.h:189), the last expected decrease (:303), the step size of the last
 * accepted trial and the number of steps the solve's line searches rejected in total.  Outputs are device arrays [B] (problem dtype; int32
 * for backtracks), any may be NULL.  `augmented_lagrangian` must be what the solve was called with. */
ilqg_status ilqg_solve_state_batch(const ilqg_problem* p, int32_t batch, const void* workspace,
                                   int32_t augmented_lagrangian, void* last_merit, void* expected_decrease,
                                   void* step, int32_t* backtracks, void* stream);

/* Which device schedule the problem's last solve ran (bits below), so that a caller can tell two runs apart whose results
 * differ in the last bits because ILQG_CHOICE_AUTO chose differently for their batch sizes. */
#define ILQG_SCHEDULE_SINGLE_WAVE_SWEEP 1   /* one wave per instance (else: one wave per player)                  */
#define ILQG_SCHEDULE_ADJOINT_DECREASE 2    /* ExpectedDecrease by the sweep's adjoint recursion                  */
#define ILQG_SCHEDULE_SPLIT_TRIAL 4         /* trial pass as rollout / rows / decision kernels                    */
#define ILQG_SCHEDULE_COMPACT_ROWS 8        /* compact rows between the row stage and the sweep                   */
#define ILQG_SCHEDULE_COUNTED 16            /* host-counted rounds (hand-off, speculative line search)            */
#define ILQG_SCHEDULE_GENERIC 32            /* the run-time-dimensioned kernels                                   */
#define ILQG_SCHEDULE_OPEN_LOOP 64          /* LQOpenLoopSolver's sweep                                           */
#define ILQG_SCHEDULE_STATIC_ROWS 128       /* the row stage ran as straight-line code for a registered structure */
#define ILQG_SCHEDULE_PADDED_SWEEP 256      /* run-time-dimensioned solve, its sweep on a specialised kernel (padded_sweep) */
ilqg_status ilqg_problem_last_schedule(const ilqg_problem* p, int32_t* schedule_out);

/* The row program ilqg_problem_create compiled the problem's dynamics and cost list into (csrc/ilqg_rowprog.hpp: passes,
 * ops, slot ids, word maps; host memory, int32 words) and the id of the registered structure it matches (0: none — the
 * row stage interprets it).  words_out may be NULL to ask for the size only. */
ilqg_status ilqg_problem_row_program(const ilqg_problem* p, int32_t* words_out, int32_t capacity, int32_t* num_words,
                                     int32_t* static_id);
/* The same straight from a description, on the host alone: everything ilqg_problem_create validates and builds before
 * it uploads tables, without a device (the one entry point that works without one). */
ilqg_status ilqg_row_program_build(const ilqg_problem_desc* desc, int32_t* words_out, int32_t capacity, int32_t* num_words,
                                   int32_t* static_id);

/* Replaces AugmentedLagrangianSolver::Solve (src/augmented_lagrangian_solver.cpp:72-210) with
 * max_runtime = infinity: inner ilqg_ilq_solve_batch calls capped at
 * params.unconstrained_solver_max_iters, multiplier update lambda <- max(0, lambda + mu g) at the
 * final operating point (with the reference's TimeIndex aliasing), mu *= geometric_mu_scaling,
 * warm restart from the last successful inner solve, down-scaling of lambda and mu after a failed
 * one, until the SolverLog would hold max_solver_iters iterates or max g <= constraint_error_tolerance.
 * One (lambda, mu) state per instance replaces the reference's process-global Constraint::mu_.
 *  iters [B]  = number of SolverLog iterates;  status [B] = overall success flag.
 * Same buffers as ilqg_ilq_solve_batch; workspace from ilqg_workspace_bytes. */
ilqg_status ilqg_al_solve_batch(ilqg_problem* p, int32_t batch, const void* x0,
                                void* xs, void* us, void* P, void* alpha,
                                void* total_costs, int32_t* iters,
                                int32_t* status, int32_t* converged,
                                void* workspace, void* stream);

/* Replaces Problem::SetUpNextRecedingHorizon (src/problem.cpp:127-186) with Problem::SyncToExistingProblem
 * (:64-125) and the integrators of src/multi_player_integrable_system.cpp:76-130, for `batch` plans that share
 * one time base: the measured state is integrated forward under the stored strategies by ~planner_runtime, the
 * nearest plan state is found (first subsystem's position distance), the plan is shifted to start there, the
 * tail gets zero strategies / controls and is re-propagated.
 *  x0          [B][n]   measured state at absolute time t0
 *  t0, planner_runtime  the reference's arguments;  plan_t0 = OperatingPoint::t0 of the stored plan
 *  xs, us, P, alpha     in: stored plan (previous solution);  out: warm start of the next solve
 *  x0_next     [B][n]   out: Problem::x0_ of the next solve (Stitch of the nearest plan state and the integrated one)
 *  first_step  [B] int32 out: first_timestep_in_new_problem
 *  new_plan_t0_host     out (host): OperatingPoint::t0 of the shifted plan
 * Invalid times (t0 before the plan, t0 + planner_runtime past its horizon: the reference's CHECKs) return
 * ILQG_ERR_INVALID. */
ilqg_status ilqg_receding_horizon_shift_batch(const ilqg_problem* p, int32_t batch, const void* x0, double t0,
                                              double planner_runtime, double plan_t0, void* xs, void* us,
                                              void* P, void* alpha, void* x0_next, int32_t* first_step,
                                              double* new_plan_t0_host, void* stream);

/* ------------------------------------------------------------------------ *
 *  Receding-horizon harness (examples/receding_horizon_simulator.h:58-60)   *
 * ------------------------------------------------------------------------ *
 * The pieces RecedingHorizonSimulator (src/receding_horizon_simulator.cpp:64-137) strings together, for
 * `batch` instances that share the simulator's clock but each keep their own stored plan.  A stored plan is
 * what SolutionSplicer holds (solver/solution_splicer.h:57-86): T .. T+5 rows, its own start time:
 *  plan_xs/us/P/alpha [B][plan_rows][n | m | m*n | m]   plan_rows >= T + 5
 *  plan_len [B] int32 (device)    rows in use (0 = not constructed yet)
 *  plan_t0  [B] double (device)   OperatingPoint::t0 of the stored plan
 *  active   [B] int32 (device)    1 = instance still running.  Where the reference would leave the loop
 *                                 (ContainsTime false) or CHECK-abort on the times, the instance is
 *                                 cleared instead and left untouched; the batch never aborts. */

/* Replaces MultiPlayerIntegrableSystem::Integrate(t0, t, x0, operating_point, strategies)
 * (src/multi_player_integrable_system.cpp:54-74 with :76-155): x [B][n] is advanced from t_from to t_to under
 * the stored plan's strategies.  Instances whose plan does not satisfy SolutionSplicer::ContainsTime(must_contain)
 * (solution_splicer.h:66-71) are deactivated first — the simulator's loop exits (:89-91, :126). */
ilqg_status ilqg_plan_integrate_batch(const ilqg_problem* p, int32_t batch, int32_t plan_rows, const void* plan_xs,
                                      const void* plan_us, const void* plan_P, const void* plan_alpha,
                                      const int32_t* plan_len, const double* plan_t0, double t_from, double t_to,
                                      double must_contain, void* x, int32_t* active, void* stream);

/* Replaces Problem::OverwriteSolution(stored plan) + Problem::SetUpNextRecedingHorizon(x, t, planner_runtime)
 * (src/problem.cpp:127-193, receding_horizon_simulator.cpp:98-103) per instance: like
 * ilqg_receding_horizon_shift_batch, but every instance has its own plan length / start time and the result goes
 * to the next solve's buffers.
 *  x          [B][n]       measured state at time t
 *  xs, us, P, alpha [B][T][..]  out: warm start of the next solve (must not alias the plan)
 *  x0_next    [B][n]       out: Problem::InitialState of the next solve
 *  solve_t0   [B] double   out (device): OperatingPoint::t0 of the next solve
 *  first_step [B] int32    out: first_timestep_in_new_problem, -1 where the times were invalid */
ilqg_status ilqg_receding_horizon_sync_batch(const ilqg_problem* p, int32_t batch, int32_t plan_rows,
                                             const void* plan_xs, const void* plan_us, const void* plan_P,
                                             const void* plan_alpha, const int32_t* plan_len, const double* plan_t0,
                                             const void* x, double t, double planner_runtime, void* xs, void* us,
                                             void* P, void* alpha, void* x0_next, double* solve_t0,
                                             int32_t* first_step, int32_t* active, void* stream);

/* Replaces SolutionSplicer::SolutionSplicer(log) (src/solution_splicer.cpp:56-58; instances with plan_len == 0)
 * and SolutionSplicer::Splice(log) (:60-129; instances with converged != 0, receding_horizon_simulator.cpp:133):
 * up to five rows of the old plan in front of the solution's start are kept, the solution follows.
 *  xs, us, P, alpha [B][T][..] the solve's final operating point / strategies, solve_t0 [B] its start time
 *  converged, active [B] int32 (device), either may be NULL (= all ones) */
ilqg_status ilqg_solution_splice_batch(const ilqg_problem* p, int32_t batch, int32_t plan_rows, void* plan_xs,
                                       void* plan_us, void* plan_P, void* plan_alpha, int32_t* plan_len,
                                       double* plan_t0, const void* xs, const void* us, const void* P,
                                       const void* alpha, const double* solve_t0, const int32_t* converged,
                                       const int32_t* active, void* stream);

/* GameSolver::Solve called AGAIN on the same solver object (the simulator reuses one, :75,108): like
 * ilqg_ilq_solve_batch / ilqg_al_solve_batch (augmented_lagrangian != 0), except that
 * ILQSolver::last_merit_function_value_ (solver/ilq_solver.h:189) starts from what the previous call on this
 * `workspace` left instead of infinity.  The workspace must come from a previous solve of the same kind and
 * batch.  Constraint multipliers are indexed relative to the start of each window, as in a first solve.
 * active [B] int32 (device, nullable): instances with 0 are skipped, their buffers and outputs left as they are. */
ilqg_status ilqg_solve_again_batch(ilqg_problem* p, int32_t batch, const void* x0, void* xs, void* us, void* P,
                                   void* alpha, void* total_costs, int32_t* iters, int32_t* status,
                                   int32_t* converged, void* workspace, int32_t augmented_lagrangian,
                                   const int32_t* active, void* stream);

/* ------------------------------------------------------------------------ *
 *  Equilibrium checks                                                       *
 * ------------------------------------------------------------------------ */

/* Replaces ComputeStrategyCosts (src/compute_strategy_costs.cpp:61-106): the cost every player accumulates when
 * the strategies (P, alpha) are played from x0 against the operating point (xs, us) — closed loop, or with
 * open_loop != 0 as u = u_ref - alpha with state costs taken at the next state (PlayerCost::EvaluateOffset,
 * src/player_cost.cpp:175-190).  euler != 0: one-step Euler integration
 * (MultiPlayerIntegrableSystem::IntegrateUsingEuler), else the default RK4.  costs [B][N]. */
ilqg_status ilqg_strategy_costs_batch(const ilqg_problem* p, int32_t batch, const void* x0, const void* xs,
                                      const void* us, const void* P, const void* alpha, int32_t open_loop,
                                      int32_t euler, void* costs, void* stream);

/* Replaces NumericalCheckLocalNashEquilibrium (src/check_local_nash_equilibrium.cpp:60-133): every entry of every
 * alpha_i[k], k < T-1, is moved by -/+ max_perturbation in turn (Euler integration, as there) and the mover's cost
 * compared with the nominal one — 2 m (T-1) rollouts per instance, all in one launch.
 *  is_nash [B] int32   1 = no unilateral move lowered its mover's cost
 *  margin  [B]         out (nullable): min over moves of (moved cost - nominal cost) of the mover */
ilqg_status ilqg_check_local_nash_batch(const ilqg_problem* p, int32_t batch, const void* x0, const void* xs,
                                        const void* us, const void* P, const void* alpha, double max_perturbation,
                                        int32_t open_loop, int32_t* is_nash, void* margin, void* stream);

/* Replaces CheckSufficientLocalNashEquilibrium (src/check_local_nash_equilibrium.cpp:144-201): every player's full
 * PlayerCost::Quadraticize at every step of the operating point (xs, us) has Q_i and all its R_ij without an
 * eigenvalue below -1e-4 (decided by a Cholesky factorisation of the matrix shifted by 1e-4, not by a spectrum).
 *  is_psd [B] int32 (device) */
ilqg_status ilqg_check_sufficient_nash_batch(const ilqg_problem* p, int32_t batch, const void* xs, const void* us,
                                             int32_t* is_psd, void* stream);

/* Diagnostics: out = X^T Y + C for 16x16 column-major device matrices, computed through the
 * MFMA accumulator-layout path the LQ sweep is built on (pins the gfx950 register layouts). */
ilqg_status ilqg_selftest_mfma(int32_t dtype, const void* X, const void* Y, const void* C, void* out, void* stream);

/* Diagnostics: a streaming 16-byte-per-lane copy of `bytes` (a multiple of 16; both buffers 16-byte aligned, device
 * memory) on `stream` — the copy kernel SURVEY.md 8(d) asks the roofline's measured-bandwidth denominator to come from
 * (bench.py times it with HIP events: 2 x bytes per launch / time). */
ilqg_status ilqg_copy_bandwidth(void* dst, const void* src, size_t bytes, void* stream);

/* Last HIP / validation error text of the calling thread. */
const char* ilqg_last_error(void);

/* Scratch of the stand-alone stage entry points that take no workspace argument (ilqg_lq_feedback_batch /
 * ilqg_lq_openloop_batch when delta_x or costates are asked for, ilqg_total_costs_batch, ilqg_check_*_nash_batch):
 * by default the library keeps one grow-only device allocation per calling thread for them.  A caller that owns all
 * device memory hands a buffer of its own here (per calling thread; NULL returns to the default): nothing is
 * allocated afterwards, and a call that needs more than `bytes` fails with ILQG_ERR_INVALID and the size it needs in
 * ilqg_last_error().  The solves (ilqg_*_solve_batch*) keep their memory in the workspace, with one exception: a solve
 * on the run-time-dimensioned kernels whose sweep runs on the padded specialised kernel (ilqg_solve_options::padded_sweep)
 * stages the padded rows of every instance in this scratch (batch x T x ~(N + 3) n'^2 elements). */
ilqg_status ilqg_set_scratch(void* device_buffer, size_t bytes);

/* Library / device introspection (used by the loader to fail loudly). */
#define ILQG_ABI_VERSION 8 /* 8: ilqg_solve_options::padded_sweep (was reserved1) / probe_lanes (new, with reserved2: the struct grew by
                                 eight bytes), ILQG_SCHEDULE_PADDED_SWEEP;
                              7: ilqg_solve_options::deterministic (was reserved0) / static_rows, ilqg_copy_bandwidth, ilqg_problem_row_program, ilqg_row_program_build;
                              6: ilqg_problem_last_schedule;
                              5: ilqg_solve_options::iterate_log / max_runtime, run-time-dimensioned kernels behind every entry
                              point (any n <= 32, N <= 8, m_i), the affine constraints (ilqg_problem_desc::dense_params);
                              4: ilqg_cost_term::idx_extra / value2, cost kinds 12-21, dynamics kinds 10-12;
                              3: ilqg_solve_options / ilqg_solve_batch_ex / ilqg_solve_state_batch, ilqg_dims::sweep_formulation;
                              the workspace holds every device buffer a solve uses */
int32_t ilqg_abi_version(void);
ilqg_status ilqg_device_info(char* name_out, int32_t name_len, int32_t* num_cus);

#ifdef __cplusplus
}
#endif
#endif /* ILQG_H_ */
