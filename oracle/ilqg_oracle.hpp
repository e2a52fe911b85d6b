// ilqg_oracle.hpp — CPU restatement of the iLQGames hot path.  TEST INFRASTRUCTURE ONLY.
//
// This file is the parity oracle and the CPU baseline for the MI355X kernels in
// ilqgames_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
// `cpu_baseline` leg may build, link or call it; the product path never does.
//
// It follows the reference C++ (paths relative to the reference repo root)
// function by function; every routine cites the lines it restates.  It is
// templated on the scalar type: S=float mirrors the reference's arithmetic
// (types.h:68-69 — Eigen MatrixXf/VectorXf everywhere), S=double is the ground
// truth the fp64 device path is compared against.
//
// Pinning: the reference C++ cannot be built in this image (Eigen3/glog/gflags
// absent), so the oracle is pinned against
//   * the reference's importable numpy LQ solver python/solve_lq_game.py
//     (fixtures under tests/golden/, generator tests/golden/make_golden.py),
//   * the known-answer tables of test/test_polyline2.cpp, test/test_line_segment2.cpp,
//   * the properties of test/test_lq_solver.cpp, test_linearization.cpp,
//     test_quadraticization.cpp, test_player_cost.cpp re-expressed in tests/.
// Bitwise parity with Eigen's own summation order / QR is NOT pinned.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

#include "../include/ilqg.h"

namespace oracle {

// ---------------------------------------------------------------------------
// Minimal dense column-major matrix (Eigen's default storage order).
// ---------------------------------------------------------------------------
template <class S>
struct Mat {
  int r = 0, c = 0;
  std::vector<S> d;
  Mat() {}
  Mat(int rows, int cols) : r(rows), c(cols), d(size_t(rows) * cols, S(0)) {}
  S& operator()(int i, int j) { return d[size_t(i) + size_t(j) * r]; }
  S operator()(int i, int j) const { return d[size_t(i) + size_t(j) * r]; }
  static Mat Identity(int n, S scale = S(1)) {
    Mat m(n, n);
    for (int i = 0; i < n; i++) m(i, i) = scale;
    return m;
  }
  void setZero() { std::fill(d.begin(), d.end(), S(0)); }
};
template <class S>
using Vec = std::vector<S>;

template <class S>
Mat<S> matmul(const Mat<S>& a, const Mat<S>& b) {
  Mat<S> o(a.r, b.c);
  for (int j = 0; j < b.c; j++)
    for (int k = 0; k < a.c; k++) {
      const S bkj = b(k, j);
      for (int i = 0; i < a.r; i++) o(i, j) += a(i, k) * bkj;
    }
  return o;
}
template <class S>
Mat<S> matmulTN(const Mat<S>& a, const Mat<S>& b) {  // a^T b
  Mat<S> o(a.c, b.c);
  for (int j = 0; j < b.c; j++)
    for (int i = 0; i < a.c; i++) {
      S s = 0;
      for (int k = 0; k < a.r; k++) s += a(k, i) * b(k, j);
      o(i, j) = s;
    }
  return o;
}
template <class S>
Vec<S> matvec(const Mat<S>& a, const Vec<S>& x) {
  Vec<S> o(a.r, S(0));
  for (int k = 0; k < a.c; k++)
    for (int i = 0; i < a.r; i++) o[i] += a(i, k) * x[k];
  return o;
}
template <class S>
Vec<S> matvecT(const Mat<S>& a, const Vec<S>& x) {  // a^T x
  Vec<S> o(a.c, S(0));
  for (int i = 0; i < a.c; i++) {
    S s = 0;
    for (int k = 0; k < a.r; k++) s += a(k, i) * x[k];
    o[i] = s;
  }
  return o;
}

// ---------------------------------------------------------------------------
// Householder QR solve, restating Eigen's HouseholderQR (unblocked path) +
// solve: the call `S_.householderQr().solve(Y_)` of
// src/lq_feedback_solver.cpp:180 and src/lq_open_loop_solver.cpp:131,144,148,165.
// Eigen is an un-vendored dependency of the reference (cmake/Dependencies.cmake:5,
// any Eigen >= 2.91); the algorithm restated is Eigen 3.3's
// householder_qr_inplace_unblocked + MatrixBase::makeHouseholder +
// applyHouseholderOnTheLeft, then back substitution on R.
// ---------------------------------------------------------------------------
template <class S>
struct HouseholderQR {
  Mat<S> qr;
  Vec<S> tau;
  void compute(const Mat<S>& a) {
    qr = a;
    const int rows = a.r, cols = a.c, size = std::min(rows, cols);
    tau.assign(size, S(0));
    for (int k = 0; k < size; k++) {
      const int rem = rows - k;
      // makeHouseholderInPlace on qr.col(k).tail(rem)
      S tailsq = 0;
      for (int i = k + 1; i < rows; i++) tailsq += qr(i, k) * qr(i, k);
      const S c0 = qr(k, k);
      S beta, t;
      if (rem == 1 || tailsq <= std::numeric_limits<S>::min()) {
        t = 0;
        beta = c0;
        for (int i = k + 1; i < rows; i++) qr(i, k) = 0;
      } else {
        beta = std::sqrt(c0 * c0 + tailsq);
        if (c0 >= S(0)) beta = -beta;
        for (int i = k + 1; i < rows; i++) qr(i, k) = qr(i, k) / (c0 - beta);
        t = (beta - c0) / beta;
      }
      tau[k] = t;
      qr(k, k) = beta;
      // apply H_k to the trailing columns
      for (int j = k + 1; j < cols; j++) applyReflector(k, t, qr, j);
    }
  }
  // x(:, j) <- H_k x(:, j), H_k = I - tau v v^T, v = [1; qr(k+1:, k)]
  void applyReflector(int k, S t, Mat<S>& x, int j) const {
    const int rows = qr.r;
    if (rows - k == 1) {
      x(k, j) *= (S(1) - t);
      return;
    }
    if (t == S(0)) return;
    S tmp = 0;
    for (int i = k + 1; i < rows; i++) tmp += qr(i, k) * x(i, j);
    tmp += x(k, j);
    x(k, j) -= t * tmp;
    for (int i = k + 1; i < rows; i++) x(i, j) -= t * qr(i, k) * tmp;
  }
  Mat<S> solve(const Mat<S>& b) const {
    Mat<S> c = b;
    const int n = qr.c, size = (int)tau.size();
    for (int k = 0; k < size; k++)
      for (int j = 0; j < c.c; j++) applyReflector(k, tau[k], c, j);
    // back substitution with R = upper triangle of qr
    Mat<S> x(n, b.c);
    for (int j = 0; j < b.c; j++)
      for (int i = n - 1; i >= 0; i--) {
        S s = c(i, j);
        for (int k2 = i + 1; k2 < n; k2++) s -= qr(i, k2) * x(k2, j);
        x(i, j) = s / qr(i, i);
      }
    return x;
  }
};

// LDLT solve for the small SPD R_ii blocks (Eigen::LDLT, lq_open_loop_solver.cpp:124-126).
// Symmetric-pivoting order of Eigen's LDLT is not restated (unpinned; R_ii is
// diagonal-dominant in every config) — plain LDL^T.
template <class S>
struct LDLT {
  Mat<S> L;
  Vec<S> D;
  void compute(const Mat<S>& a) {
    const int n = a.r;
    L = Mat<S>::Identity(n);
    D.assign(n, S(0));
    for (int j = 0; j < n; j++) {
      S dj = a(j, j);
      for (int k = 0; k < j; k++) dj -= L(j, k) * L(j, k) * D[k];
      D[j] = dj;
      for (int i = j + 1; i < n; i++) {
        S s = a(i, j);
        for (int k = 0; k < j; k++) s -= L(i, k) * L(j, k) * D[k];
        L(i, j) = s / dj;
      }
    }
  }
  Mat<S> solve(const Mat<S>& b) const {
    const int n = L.r;
    Mat<S> x = b;
    for (int c = 0; c < b.c; c++) {
      for (int i = 0; i < n; i++)
        for (int k = 0; k < i; k++) x(i, c) -= L(i, k) * x(k, c);
      for (int i = 0; i < n; i++) x(i, c) /= D[i];
      for (int i = n - 1; i >= 0; i--)
        for (int k = i + 1; k < n; k++) x(i, c) -= L(k, i) * x(k, c);
    }
    return x;
  }
};

template <class S>
inline S sgn(S x) {  // types.h:147-155
  return S((S(0) < x) - (x < S(0)));
}

// ---------------------------------------------------------------------------
// Geometry: LineSegment2 / Polyline2
// ---------------------------------------------------------------------------
template <class S>
struct Segment2 {  // include/ilqgames/geometry/line_segment2.h:52-91
  S p1x, p1y, p2x, p2y, length, ux, uy;
  Segment2() : p1x(0), p1y(0), p2x(1), p2y(1) { init(); }
  Segment2(S ax, S ay, S bx, S by) : p1x(ax), p1y(ay), p2x(bx), p2y(by) { init(); }
  void init() {
    const S dx = p1x - p2x, dy = p1y - p2y;
    length = std::sqrt(dx * dx + dy * dy);
    ux = (p2x - p1x) / length;
    uy = (p2y - p1y) / length;
  }
  // src/line_segment2.cpp:48-54
  bool Side(S qx, S qy) const {
    const S rx = qx - p1x, ry = qy - p1y;
    const S cross = rx * uy - ux * ry;
    return cross > S(0);
  }
  // src/line_segment2.cpp:56-100
  void ClosestPoint(S qx, S qy, S* cx, S* cy, bool* is_endpoint, S* ssd) const {
    const S rx = qx - p1x, ry = qy - p1y;
    const S dot = rx * ux + ry * uy;
    const S cross = rx * uy - ux * ry;
    const S csign = sgn(cross);
    if (dot < S(0)) {
      *is_endpoint = true;
      *ssd = csign * (rx * rx + ry * ry);
      *cx = p1x;
      *cy = p1y;
      return;
    } else if (dot > length) {
      *is_endpoint = true;
      const S ex = qx - p2x, ey = qy - p2y;
      *ssd = csign * (ex * ex + ey * ey);
      *cx = p2x;
      *cy = p2y;
      return;
    }
    *is_endpoint = false;
    *ssd = csign * cross * cross;
    *cx = p1x + dot * ux;
    *cy = p1y + dot * uy;
  }
};

template <class S>
struct Polyline2 {  // src/polyline2.cpp:52-63
  std::vector<Segment2<S>> segs;
  Polyline2() {}
  Polyline2(const float* pts, int npts) {
    for (int i = 1; i < npts; i++)
      segs.emplace_back(S(pts[2 * (i - 1)]), S(pts[2 * (i - 1) + 1]), S(pts[2 * i]),
                        S(pts[2 * i + 1]));
  }
  // src/polyline2.cpp:68-103 (the cumulative lengths of the constructor, :52-63, formed on the way)
  void PointAt(S route_pos, S* px, S* py) const {
    std::vector<S> cumulative(1, S(0));
    for (const auto& s : segs) cumulative.push_back(cumulative.back() + s.length);
    auto upper = std::upper_bound(cumulative.begin(), cumulative.end(), route_pos);
    if (upper == cumulative.end()) upper--;  // off the end of the route: the reference warns and clamps
    upper--;
    const size_t idx = size_t(upper - cumulative.begin());
    const S remaining = route_pos - cumulative[idx];
    *px = segs[idx].p1x + remaining * segs[idx].ux;
    *py = segs[idx].p1y + remaining * segs[idx].uy;
  }
  // src/polyline2.cpp:105-174
  void ClosestPoint(S qx, S qy, S* cx, S* cy, bool* is_vertex, Segment2<S>* segment,
                    S* signed_sq, bool* is_endpoint) const {
    S closest = std::numeric_limits<S>::infinity();
    S bx = 0, by = 0;
    int seg_idx = 0;
    bool vertex = false;
    const int nseg = (int)segs.size();
    for (int counter = 0; counter < nseg; counter++) {
      const Segment2<S>& s = segs[counter];
      S px, py, cur;
      bool seg_end;
      s.ClosestPoint(qx, qy, &px, &py, &seg_end, &cur);
      if (std::abs(cur) < std::abs(closest)) {
        const bool at_second = (px == s.p2x && py == s.p2y);
        const bool at_first = (px == s.p1x && py == s.p1y);
        if (seg_end && (counter > 0 || at_second) && (counter < nseg - 1 || at_first)) {
          const Segment2<S> shortcut =
              at_first ? Segment2<S>(segs[counter - 1].p1x, segs[counter - 1].p1y, s.p2x, s.p2y)
                       : Segment2<S>(s.p1x, s.p1y, segs[counter + 1].p2x, segs[counter + 1].p2y);
          cur *= shortcut.Side(qx, qy) ? sgn(cur) : -sgn(cur);
        }
        closest = cur;
        bx = px;
        by = py;
        vertex = seg_end;
        seg_idx = counter;
      }
    }
    if (segment) *segment = segs[seg_idx];
    if (signed_sq) *signed_sq = closest;
    if (is_vertex) *is_vertex = vertex;
    if (is_endpoint) {
      auto same = [](S ax, S ay, S cx2, S cy2) {
        const S dx = ax - cx2, dy = ay - cy2;
        return dx * dx + dy * dy < S(1e-4f);  // constants::kSmallNumber, types.h:115
      };
      *is_endpoint = same(bx, by, segs.front().p1x, segs.front().p1y) ||
                     same(bx, by, segs.back().p2x, segs.back().p2y);
    }
    *cx = bx;
    *cy = by;
  }
};

// ---------------------------------------------------------------------------
// Problem: flattened Problem/PlayerCost/ConcatenatedDynamicalSystem
// ---------------------------------------------------------------------------
template <class S>
struct Problem {
  int N = 0, n = 0, m = 0, T = 0;
  double dt = 0.1;
  std::vector<ilqg_subsystem> subs;
  std::vector<int> xoff, uoff;  // subsystem_start_dims_, concatenated_dynamical_system.cpp:62-66
  std::vector<ilqg_player_cost> pcs;
  std::vector<ilqg_cost_term> terms;
  std::vector<Polyline2<S>> polylines;
  std::vector<ilqg_pair> pairs;          // fixed (i,j) block order
  std::vector<int> pair_from_cost;       // 1 if a control COST (not only a constraint) touches it
  std::vector<int> roff, rgoff;          // offsets of pair blocks inside R / r rows
  int Rsz = 0, rsz = 0;
  int num_constraints = 0;
  ilqg_solver_params params;
  std::vector<float> dense;  // coefficients of the affine constraints (ilqg_problem_desc::dense_params)

  explicit Problem(const ilqg_problem_desc& d) {
    if (d.num_dense_params > 0 && d.dense_params) dense.assign(d.dense_params, d.dense_params + d.num_dense_params);
    N = d.num_players;
    T = d.T;
    dt = d.dt;
    params = d.params;
    xoff.push_back(0);
    uoff.push_back(0);
    for (int i = 0; i < N; i++) {
      subs.push_back(d.subsystems[i]);
      pcs.push_back(d.player_costs[i]);
      xoff.push_back(xoff.back() + d.subsystems[i].xdim);
      uoff.push_back(uoff.back() + d.subsystems[i].udim);
    }
    n = xoff.back();
    m = uoff.back();
    terms.assign(d.terms, d.terms + d.num_terms);
    for (int p = 0; p < d.num_polylines; p++) {
      const int b = d.polyline_offsets[p], e = d.polyline_offsets[p + 1];
      polylines.emplace_back(d.polyline_points + 2 * b, e - b);
    }
    // Pair table: per player, first-touch order over control costs then control
    // constraints (player_cost.cpp:59-86 creates a block at first touch).
    for (int i = 0; i < N; i++) {
      for (int pass = 0; pass < 2; pass++)
        for (const auto& t : terms) {
          if (t.player != i) continue;
          if (pass == 0 && t.role != ILQG_ROLE_CONTROL_COST) continue;
          if (pass == 1 && t.role != ILQG_ROLE_CONTROL_CONSTRAINT) continue;
          int found = -1;
          for (size_t q = 0; q < pairs.size(); q++)
            if (pairs[q].i == i && pairs[q].j == t.arg) found = (int)q;
          if (found < 0) {
            pairs.push_back({i, t.arg});
            pair_from_cost.push_back(pass == 0 ? 1 : 0);
          }
        }
    }
    for (const auto& pr : pairs) {
      roff.push_back(Rsz);
      rgoff.push_back(rsz);
      Rsz += subs[pr.j].udim * subs[pr.j].udim;
      rsz += subs[pr.j].udim;
    }
    for (const auto& t : terms)
      if (t.constraint_slot >= 0) num_constraints = std::max(num_constraints, t.constraint_slot + 1);
  }
  // LQ-only problem: dimensions + (i,j) block table, no dynamics/cost models.
  Problem(const ilqg_dims& d, const ilqg_pair* prs, int npairs) {
    N = d.num_players;
    T = d.T;
    n = d.n;
    xoff.push_back(0);
    uoff.push_back(0);
    for (int i = 0; i < N; i++) {
      ilqg_subsystem s{};
      s.udim = d.udim[i];
      subs.push_back(s);
      xoff.push_back(0);
      uoff.push_back(uoff.back() + d.udim[i]);
    }
    m = uoff.back();
    ilqg_default_params_local(&params);
    for (int q = 0; q < npairs; q++) {
      pairs.push_back(prs[q]);
      pair_from_cost.push_back(1);
      roff.push_back(Rsz);
      rgoff.push_back(rsz);
      Rsz += d.udim[prs[q].j] * d.udim[prs[q].j];
      rsz += d.udim[prs[q].j];
    }
  }
  static void ilqg_default_params_local(ilqg_solver_params* p) {  // solver_params.h:50-84
    p->convergence_tolerance = 1e-1f;
    p->max_solver_iters = 1000;
    p->linesearch = 1;
    p->initial_alpha_scaling = 0.5f;
    p->geometric_alpha_scaling = 0.5f;
    p->max_backtracking_steps = 10;
    p->expected_decrease_fraction = 0.1f;
    p->open_loop = 0;
    p->unconstrained_solver_max_iters = 10;
    p->geometric_mu_scaling = 1.1f;
    p->geometric_mu_downscaling = 0.5f;
    p->geometric_lambda_downscaling = 0.5f;
    p->constraint_error_tolerance = 1e-1f;
  }
  int udim(int i) const { return subs[i].udim; }
  int pairIndex(int i, int j) const {
    for (size_t q = 0; q < pairs.size(); q++)
      if (pairs[q].i == i && pairs[q].j == j) return (int)q;
    return -1;
  }
};

// Per-instance augmented-Lagrangian state: what the reference keeps in
// Constraint::lambdas_ (constraint.h:136) and the process-global
// Constraint::mu_ (src/constraint.cpp:61) — one copy per instance here.
template <class S>
struct ALState {
  std::vector<S> lambdas;  // [num_constraints][T]
  S mu = S(10);            // constants::kDefaultMu, types.h:128-129
  double t_init = 0.0;     // RelativeTimeTracker::initial_time_
  double dt = 0.1;
  int T = 0;
  ALState() {}
  ALState(int nc, int T_, double dt_) : lambdas(size_t(nc) * T_, S(0)), dt(dt_), T(T_) {}
  // RelativeTimeTracker::TimeIndex, relative_time_tracker.h:69-72 (double arithmetic,
  // truncation aliases k=43,81,86,91 onto k-1 for dt=0.1 — reproduced, not fixed).
  size_t TimeIndex(double t) const { return static_cast<size_t>((t - t_init) / dt); }
  S& lambda(int slot, double t) { return lambdas[size_t(slot) * T + TimeIndex(t)]; }
  S lambda(int slot, double t) const { return lambdas[size_t(slot) * T + TimeIndex(t)]; }
};

// ---------------------------------------------------------------------------
// Dynamics
// ---------------------------------------------------------------------------
// SinglePlayer{Unicycle4D,Car5D,Car6D}::Evaluate —
// single_player_unicycle_4d.h:90-100, single_player_car_5d.h:100-111,
// single_player_car_6d.h:102-114.
template <class S>
void EvaluateSubsystem(const ilqg_subsystem& s, const S* x, const S* u, S* xdot, S next_param = S(0)) {
  const S L = S(s.param0);
  switch (s.kind) {
    case ILQG_DYN_UNICYCLE_4D:
      xdot[0] = x[3] * std::cos(x[2]);
      xdot[1] = x[3] * std::sin(x[2]);
      xdot[2] = u[0];
      xdot[3] = u[1];
      break;
    case ILQG_DYN_UNICYCLE_4D_DISTURBED:  // two_player_unicycle_4d.h:105-118; u[2..3] = the other player's (dx, dy)
      xdot[0] = x[3] * std::cos(x[2]) + u[2];
      xdot[1] = x[3] * std::sin(x[2]) + u[3];
      xdot[2] = u[0];
      xdot[3] = u[1];
      break;
    case ILQG_DYN_PLANAR_DISTURBANCE:  // no state of its own
      break;
    case ILQG_DYN_AIR_3D_EVADER: {  // air_3d.h:112-125; u[1] = the pursuer's turn rate, next_param = its speed
      const S ve = L, vp = next_param;
      xdot[0] = -ve + vp * std::cos(x[2]) + u[0] * x[1];
      xdot[1] = vp * std::sin(x[2]) - u[0] * x[0];
      xdot[2] = u[1] - u[0];
      break;
    }
    case ILQG_DYN_AIR_3D_PURSUER:  // no state of its own
      break;
    case ILQG_DYN_DUBINS_CAR:  // single_player_dubins_car.h:94-103
      xdot[0] = L * std::cos(x[2]);
      xdot[1] = L * std::sin(x[2]);
      xdot[2] = u[0];
      break;
    case ILQG_DYN_POINT_MASS_2D:  // single_player_point_mass_2d.h:90-99
      xdot[0] = x[2];
      xdot[1] = x[3];
      xdot[2] = u[0];
      xdot[3] = u[1];
      break;
    case ILQG_DYN_CAR_5D:
      xdot[0] = x[4] * std::cos(x[2]);
      xdot[1] = x[4] * std::sin(x[2]);
      xdot[2] = (x[4] / L) * std::tan(x[3]);
      xdot[3] = u[0];
      xdot[4] = u[1];
      break;
    case ILQG_DYN_CAR_6D:
      xdot[0] = x[4] * std::cos(x[2]);
      xdot[1] = x[4] * std::sin(x[2]);
      xdot[2] = (x[4] / L) * std::tan(x[3]);
      xdot[3] = u[0];
      xdot[4] = x[5];
      xdot[5] = u[1];
      break;
    case ILQG_DYN_UNICYCLE_5D:  // single_player_unicycle_5d.h:92-103
      xdot[0] = x[3] * std::cos(x[2]);
      xdot[1] = x[3] * std::sin(x[2]);
      xdot[2] = u[0];
      xdot[3] = u[1];
      xdot[4] = x[3];
      break;
    case ILQG_DYN_CAR_7D: {  // single_player_car_7d.h:104-120 (sec_phi is a float made from a double quotient there)
      xdot[0] = x[4] * std::cos(x[2]);
      xdot[1] = x[4] * std::sin(x[2]);
      xdot[2] = (x[4] / L) * std::tan(x[3]);
      xdot[3] = u[0];
      xdot[4] = u[1];
      const S sec_phi = S(1.0 / double(std::cos(x[3])));
      xdot[5] = u[0] * sec_phi * sec_phi / L;
      xdot[6] = x[4];
      break;
    }
    case ILQG_DYN_DELAYED_DUBINS_CAR:  // single_player_delayed_dubins_car.h:103-113
      xdot[0] = L * std::cos(x[2]);
      xdot[1] = L * std::sin(x[2]);
      xdot[2] = x[3];
      xdot[3] = u[0];
      break;
  }
}

// ConcatenatedDynamicalSystem::Evaluate, src/concatenated_dynamical_system.cpp:69-84
template <class S>
Vec<S> Evaluate(const Problem<S>& p, const Vec<S>& x, const Vec<S>& u) {
  Vec<S> xdot(p.n);
  for (int i = 0; i < p.N; i++)
    EvaluateSubsystem(p.subs[i], x.data() + p.xoff[i], u.data() + p.uoff[i], xdot.data() + p.xoff[i],
                      i + 1 < p.N ? S(p.subs[i + 1].param0) : S(0));
  return xdot;
}

// MultiPlayerDynamicalSystem::Integrate, src/multi_player_dynamical_system.cpp:52-77.
// RK4 with 2 sub-steps; `dt` is double in the reference and narrows to the
// matrix scalar when it multiplies an Eigen expression.
template <class S>
Vec<S> Integrate(const Problem<S>& p, double t0, double interval, const Vec<S>& x0, const Vec<S>& u,
                 bool euler = false) {
  Vec<S> x = x0;
  const int n = p.n;
  if (euler) {
    const Vec<S> f = Evaluate(p, x0, u);
    for (int i = 0; i < n; i++) x[i] += S(interval) * f[i];
    return x;
  }
  const double dt = interval / 2.0;
  const S h = S(dt);
  for (double t = t0; t < t0 + interval - 0.5 * dt; t += dt) {
    Vec<S> k1 = Evaluate(p, x, u), xt(n);
    for (int i = 0; i < n; i++) { k1[i] = h * k1[i]; xt[i] = x[i] + S(0.5) * k1[i]; }
    Vec<S> k2 = Evaluate(p, xt, u);
    for (int i = 0; i < n; i++) { k2[i] = h * k2[i]; xt[i] = x[i] + S(0.5) * k2[i]; }
    Vec<S> k3 = Evaluate(p, xt, u);
    for (int i = 0; i < n; i++) { k3[i] = h * k3[i]; xt[i] = x[i] + k3[i]; }
    Vec<S> k4 = Evaluate(p, xt, u);
    for (int i = 0; i < n; i++) {
      k4[i] = h * k4[i];
      x[i] += (k1[i] + S(2.0) * (k2[i] + k3[i]) + k4[i]) / S(6.0);
    }
  }
  return x;
}

// ConcatenatedDynamicalSystem::Linearize (src/concatenated_dynamical_system.cpp:86-107)
// on top of LinearDynamicsApproximation's (I, 0) init
// (linear_dynamics_approximation.h:62-68) and the per-model `+=` Jacobians
// (single_player_unicycle_4d.h:102-116, single_player_car_5d.h:113-133,
// single_player_car_6d.h:116-138).  Mixed float*double products are kept
// exactly as the reference writes them (time::kTimeStep is double).
template <class S>
void Linearize(const Problem<S>& p, const Vec<S>& x, const Vec<S>& u, Mat<S>* A, Mat<S>* B) {
  const int n = p.n;
  const double dt = p.dt;
  *A = Mat<S>::Identity(n);
  *B = Mat<S>(n, p.m);
  for (int i = 0; i < p.N; i++) {
    const int o = p.xoff[i], uo = p.uoff[i];
    const S* xs = x.data() + o;
    const ilqg_subsystem& s = p.subs[i];
    if (s.kind == ILQG_DYN_PLANAR_DISTURBANCE) {  // two_player_unicycle_4d.h:135-136: Bs[1](px, dx) = Bs[1](py, dy) = dt
      const int po = p.xoff[i - 1];
      (*B)(po + 0, uo + 0) = S(dt);
      (*B)(po + 1, uo + 1) = S(dt);
      continue;
    }
    if (s.kind == ILQG_DYN_AIR_3D_EVADER) {  // air_3d.h:127-149 (the pursuer's column: the next row)
      const S vp = S(p.subs[i + 1].param0), w1 = u[uo];
      const S ctheta = S(double(std::cos(xs[2])) * dt), stheta = S(double(std::sin(xs[2])) * dt);
      (*A)(o + 0, o + 1) += S(double(w1) * dt);
      (*A)(o + 0, o + 2) -= vp * stheta;
      (*A)(o + 1, o + 0) -= S(double(w1) * dt);
      (*A)(o + 1, o + 2) += vp * ctheta;
      (*B)(o + 0, uo) = S(double(xs[1]) * dt);
      (*B)(o + 1, uo) = S(double(-xs[0]) * dt);
      (*B)(o + 2, uo) = S(-dt);
      continue;
    }
    if (s.kind == ILQG_DYN_AIR_3D_PURSUER) {
      (*B)(p.xoff[i - 1] + 2, uo) = S(dt);
      continue;
    }
    if (s.kind == ILQG_DYN_POINT_MASS_2D) {  // single_player_point_mass_2d.h:101-110
      (*A)(o + 0, o + 2) += S(dt);
      (*A)(o + 1, o + 3) += S(dt);
      (*B)(o + 2, uo + 0) = S(dt);
      (*B)(o + 3, uo + 1) = S(dt);
      continue;
    }
    if (s.kind == ILQG_DYN_DUBINS_CAR) {  // single_player_dubins_car.h:105-117
      const S v = S(s.param0);
      const S ctheta = S(double(std::cos(xs[2])) * dt), stheta = S(double(std::sin(xs[2])) * dt);
      (*A)(o + 0, o + 2) += -v * stheta;
      (*A)(o + 1, o + 2) += v * ctheta;
      (*B)(o + 2, uo + 0) = S(dt);
      continue;
    }
    if (s.kind == ILQG_DYN_DELAYED_DUBINS_CAR) {  // single_player_delayed_dubins_car.h:115-127
      const S v = S(s.param0);
      const S ctheta = S(double(std::cos(xs[2])) * dt), stheta = S(double(std::sin(xs[2])) * dt);
      (*A)(o + 0, o + 2) += -v * stheta;
      (*A)(o + 1, o + 2) += v * ctheta;
      (*A)(o + 2, o + 3) += S(dt);
      (*B)(o + 3, uo + 0) = S(dt);
      continue;
    }
    const bool unicycle = s.kind == ILQG_DYN_UNICYCLE_4D || s.kind == ILQG_DYN_UNICYCLE_4D_DISTURBED ||
                          s.kind == ILQG_DYN_UNICYCLE_5D;
    const int vidx = unicycle ? 3 : 4;
    const S ctheta = S(double(std::cos(xs[2])) * dt);
    const S stheta = S(double(std::sin(xs[2])) * dt);
    (*A)(o + 0, o + 2) += -xs[vidx] * stheta;
    (*A)(o + 0, o + vidx) += ctheta;
    (*A)(o + 1, o + 2) += xs[vidx] * ctheta;
    (*A)(o + 1, o + vidx) += stheta;
    if (unicycle) {
      (*B)(o + 2, uo + 0) = S(dt);
      (*B)(o + 3, uo + 1) = S(dt);
      if (s.kind == ILQG_DYN_UNICYCLE_5D) (*A)(o + 4, o + 3) += S(dt);  // single_player_unicycle_5d.h:117
    } else {
      const S L = S(s.param0);
      const S cphi = std::cos(xs[3]);
      const S tphi = std::tan(xs[3]);
      (*A)(o + 2, o + 3) += S(double(xs[4]) * dt / double(L * cphi * cphi));
      (*A)(o + 2, o + 4) += S(double(tphi) * dt / double(L));
      if (s.kind == ILQG_DYN_CAR_5D) {
        (*B)(o + 3, uo + 0) = S(dt);
        (*B)(o + 4, uo + 1) = S(dt);
      } else if (s.kind == ILQG_DYN_CAR_7D) {  // single_player_car_7d.h:122-152: the all-double products of the
        const S den = cphi * cphi * L;         // curvature row narrow once, at the store
        (*A)(o + 5, o + 3) += S(2.0 * dt * double(u[uo]) * double(tphi) / double(den));
        (*A)(o + 6, o + 4) += S(dt);
        (*B)(o + 3, uo + 0) = S(dt);
        (*B)(o + 4, uo + 1) = S(dt);
        (*B)(o + 5, uo + 0) = S(dt / double(den));
      } else {
        (*A)(o + 4, o + 5) += S(dt);
        (*B)(o + 3, uo + 0) = S(dt);
        (*B)(o + 5, uo + 1) = S(dt);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Costs
// ---------------------------------------------------------------------------
// Constraint::Mu(lambda, g), constraint.h:112-117
template <class S>
inline S ConstraintMu(S lambda, S g, S mu, bool is_equality = false) {
  if (!is_equality && g <= S(1e-4f) && std::abs(lambda) <= S(1e-4f)) return S(0);
  return mu;
}

// The per-step nominal of the two time-dependent costs at step k (t = RelativeTime(k) = double(k) * dt, Time = double;
// relative_time_tracker.h:63-65): NominalPathLengthCost's t * nominal_speed (a double product of a double and a
// float, nominal_path_length_cost.cpp:53) and RouteProgressCost's route point (the route position is a float made from
// that double expression, route_progress_cost.cpp:57-59, RelativeTimeTracker's initial time = 0).
// Returned as doubles: the path-length nominal IS a double there; the route point is an S pair (exact in double).
template <class S>
inline void TimeNominal(const Problem<S>& p, const ilqg_cost_term& c, int k, double* nx, double* ny) {
  const double t = double(k) * p.dt;
  if (c.kind == ILQG_COST_NOMINAL_PATH_LENGTH) {
    *nx = t * double(c.value);
    *ny = 0.0;
  } else {
    const S route_pos = S(double(c.value2) + (t - 0.0) * double(c.value));
    S px, py;
    p.polylines[c.polyline].PointAt(route_pos, &px, &py);
    *nx = double(px);
    *ny = double(py);
  }
}

// Cost::Evaluate for every in-scope kind.  `step`: the time step the term is evaluated at (the time-dependent kinds).
template <class S>
S EvaluateTerm(const Problem<S>& p, int ti, const S* v, int dim, int step = 0) {
  const ilqg_cost_term& c = p.terms[ti];
  const S w = S(c.weight), val = S(c.value);
  const bool oriented = c.flags & ILQG_FLAG_ORIENTED;
  switch (c.kind) {
    case ILQG_COST_NOMINAL_PATH_LENGTH: {  // src/nominal_path_length_cost.cpp:50-56
      double nom, unused;
      TimeNominal(p, c, step, &nom, &unused);
      const S delta = S(double(v[c.idx[0]]) - nom);
      return S(0.5) * w * delta * delta;
    }
    case ILQG_COST_ROUTE_PROGRESS: {  // src/route_progress_cost.cpp:52-64
      double px, py;
      TimeNominal(p, c, step, &px, &py);
      const S dx = v[c.idx[0]] - S(px), dy = v[c.idx[1]] - S(py);
      return S(0.5) * w * (dx * dx + dy * dy);
    }
    case ILQG_COST_QUADRATIC: {  // src/quadratic_cost.cpp:51-63
      if (c.idx[0] >= 0) {
        const S delta = v[c.idx[0]] - val;
        return S(0.5) * w * delta * delta;
      }
      S sq = 0;
      for (int i = 0; i < dim; i++) sq += (v[i] - val) * (v[i] - val);
      return S(0.5) * w * sq;
    }
    case ILQG_COST_SEMIQUADRATIC: {  // src/semiquadratic_cost.cpp:51-59
      const S diff = v[c.idx[0]] - val;
      if ((diff > S(0) && oriented) || (diff < S(0) && !oriented)) return S(0.5) * w * diff * diff;
      return S(0);
    }
    case ILQG_COST_QUADRATIC_POLYLINE2: {  // src/quadratic_polyline2_cost.cpp:52-69
      S cx, cy, ssd;
      bool endp;
      p.polylines[c.polyline].ClosestPoint(v[c.idx[0]], v[c.idx[1]], &cx, &cy, nullptr, nullptr, &ssd,
                                           &endp);
      if (endp) ssd = S(0);
      return S(0.5) * w * std::abs(ssd);
    }
    case ILQG_COST_SEMIQUADRATIC_POLYLINE2: {  // src/semiquadratic_polyline2_cost.cpp:52-74
      S cx, cy, ssd;
      bool endp;
      p.polylines[c.polyline].ClosestPoint(v[c.idx[0]], v[c.idx[1]], &cx, &cy, nullptr, nullptr, &ssd,
                                           &endp);
      if (endp) return S(0);
      const S sst = sgn(val) * val * val;  // semiquadratic_polyline2_cost.h:66
      const bool active = (ssd > sst && oriented) || (ssd < sst && !oriented);
      if (!active) return S(0);
      const S sd = sgn(ssd) * std::sqrt(std::abs(ssd));
      const S diff = sd - val;
      return S(0.5) * w * diff * diff;
    }
    case ILQG_COST_PROXIMITY: {  // src/proximity_cost.cpp:52-61
      const S dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      const S dsq = dx * dx + dy * dy;
      if (dsq >= val * val) return S(0);
      const S gap = val - std::sqrt(dsq);
      return S(0.5) * w * gap * gap;
    }
    case ILQG_COST_SIGNED_DISTANCE: {  // src/signed_distance_cost.cpp:51-63
      const S dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      const S cost = val - std::hypot(dx, dy);
      return oriented ? cost : -cost;
    }
    case ILQG_COST_QUADRATIC_DIFFERENCE: {  // src/quadratic_difference_cost.cpp:51-59
      S total = 0;
      for (int ii = 0; ii < 2; ii++) {
        const S diff = v[c.idx[ii]] - v[c.idx[2 + ii]];
        total += diff * diff;
      }
      return S(0.5) * w * total;
    }
    case ILQG_COST_POLYLINE2_SIGNED_DISTANCE: {  // src/polyline2_signed_distance_cost.cpp:52-65
      S cx, cy, ssd;
      bool is_vertex, is_endpoint;
      Segment2<S> seg;
      p.polylines[c.polyline].ClosestPoint(v[c.idx[0]], v[c.idx[1]], &cx, &cy, &is_vertex, &seg, &ssd, &is_endpoint);
      if (!oriented) ssd *= S(-1);
      return sgn(ssd) * std::sqrt(std::abs(ssd)) - val;
    }
    case ILQG_COST_ORIENTATION: {  // src/orientation_cost.cpp:50-58 (the wrap is computed in double there)
      const S diff = S(std::fmod(double(v[c.idx[0]] - val) + M_PI, M_PI * 2.0) - M_PI);
      return S(0.5) * w * diff * diff;
    }
    case ILQG_COST_QUADRATIC_NORM: {  // src/quadratic_norm_cost.cpp:50-57
      const S diff = std::hypot(v[c.idx[0]], v[c.idx[1]]) - val;
      return S(0.5) * w * diff * diff;
    }
    case ILQG_COST_SEMIQUADRATIC_NORM: {  // src/semiquadratic_norm_cost.cpp:50-59
      const S diff = std::hypot(v[c.idx[0]], v[c.idx[1]]) - val;
      if ((diff > S(0) && oriented) || (diff < S(0) && !oriented)) return S(0.5) * w * diff * diff;
      return S(0);
    }
    case ILQG_COST_RELATIVE_DISTANCE: {  // src/relative_distance_cost.cpp:50-54
      return w * std::hypot(v[c.idx[0]] - v[c.idx[2]], v[c.idx[1]] - v[c.idx[3]]);
    }
    case ILQG_COST_LOCALLY_CONVEX_PROXIMITY: {  // src/locally_convex_proximity_cost.cpp:50-60
      const S dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      if (dx * dx >= val * val || dy * dy >= val * val) return S(0);
      const S delta_x = val - std::abs(dx), delta_y = val - std::abs(dy);
      return S(0.5) * w * std::min(delta_x * delta_x, delta_y * delta_y);
    }
    case ILQG_COST_WEIGHTED_CONVEX_PROXIMITY: {  // src/weighted_convex_proximity_cost.cpp:50-61
      const S dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      const S v1 = v[c.idx_extra[0]], v2 = v[c.idx_extra[1]];
      const S vv = v1 * v1 + v2 * v2;
      if (dx * dx >= val * val || dy * dy >= val * val) return S(0);
      const S delta_x = val - std::abs(dx), delta_y = val - std::abs(dy);
      return S(0.5) * w * vv * std::min(delta_x * delta_x, delta_y * delta_y);
    }
    case ILQG_COST_CURVATURE: {  // src/curvature_cost.cpp:50-53, curvature_cost.h: omega / v
      const S curvature = v[c.idx[0]] / v[c.idx[1]];
      return S(0.5) * w * curvature * curvature;
    }
    case ILQG_CONSTRAINT_POLYLINE2_SIGNED_DISTANCE: {  // src/polyline2_signed_distance_constraint.cpp:52-63
      S cx, cy, ssd;
      p.polylines[c.polyline].ClosestPoint(v[c.idx[0]], v[c.idx[1]], &cx, &cy, nullptr, nullptr, &ssd, nullptr);
      const S value = sgn(ssd) * std::sqrt(std::abs(ssd)) - val;
      return oriented ? value : -value;
    }
    case ILQG_COST_EXTREME_VALUE: {  // src/extreme_value_cost.cpp:51-85
      const bool is_min = c.flags & ILQG_FLAG_IS_MIN;
      S ext = is_min ? std::numeric_limits<S>::infinity() : -std::numeric_limits<S>::infinity();
      for (int q = 0; q < c.child_count; q++) {
        const S value = EvaluateTerm(p, c.child_begin + q, v, dim);
        if ((is_min && value < ext) || (!is_min && value > ext)) ext = value;
      }
      return ext;
    }
    case ILQG_CONSTRAINT_PROXIMITY: {  // src/proximity_constraint.cpp:56-62
      const S dx = v[c.idx[0]] - v[c.idx[2]], dy = v[c.idx[1]] - v[c.idx[3]];
      const S value = std::hypot(dx, dy) - val;
      return oriented ? value : -value;
    }
    case ILQG_CONSTRAINT_SINGLE_DIMENSION:  // single_dimension_constraint.h:68-70
      return oriented ? v[c.idx[0]] - val : val - v[c.idx[0]];
    case ILQG_CONSTRAINT_AFFINE_SCALAR: {  // affine_scalar_constraint.h:63-66: a^T x - b
      const float* a = p.dense.data() + c.polyline;
      S s = S(0);
      for (int i = 0; i < dim; i++) s += S(a[i]) * v[i];
      return s - S(a[dim]);
    }
    case ILQG_CONSTRAINT_AFFINE_VECTOR: {  // affine_vector_constraint.h:70-73: |A x - b|
      const float* A = p.dense.data() + c.polyline;
      const float* b = A + dim * dim;
      S sq = S(0);
      for (int i = 0; i < dim; i++) {
        S d = S(0);
        for (int j = 0; j < dim; j++) d += S(A[i + dim * j]) * v[j];
        d -= S(b[i]);
        sq += d * d;
      }
      return std::sqrt(sq);
    }
  }
  return S(0);
}

// Constraint::ModifyDerivatives, src/constraint.cpp:63-89
template <class S>
void ModifyDerivatives(S lambda, S mu_in, S g, S* dx, S* ddx, S* dy = nullptr, S* ddy = nullptr,
                       S* dxdy = nullptr) {
  const S mu = ConstraintMu(lambda, g, mu_in);
  const S new_dx = lambda * *dx + mu * g * *dx;
  const S new_ddx = lambda * *ddx + mu * (*dx * *dx + g * *ddx);
  if (dy) {
    const S new_dy = lambda * *dy + mu * g * *dy;
    const S new_ddy = lambda * *ddy + mu * (*dy * *dy + g * *ddy);
    const S new_dxdy = lambda * *dxdy + mu * (*dy * *dx + g * *dxdy);
    *dy = new_dy;
    *ddy = new_ddy;
    *dxdy = new_dxdy;
  }
  *dx = new_dx;
  *ddx = new_ddx;
}

// Cost::Quadraticize (accumulating) for every in-scope kind.
template <class S>
void QuadraticizeTerm(const Problem<S>& p, int ti, double t, const S* v, int dim, Mat<S>* hess,
                      Vec<S>* grad, const ALState<S>* al) {
  // FinalTimeCost::Quadraticize (cost/final_time_cost.h:73-77): nothing before the threshold step
  if (p.terms[ti].first_step > 0 && std::llround(t / p.dt) < p.terms[ti].first_step) return;
  const ilqg_cost_term& c = p.terms[ti];
  const S w = S(c.weight), val = S(c.value);
  const bool oriented = c.flags & ILQG_FLAG_ORIENTED;
  Mat<S>& H = *hess;
  Vec<S>& G = *grad;
  switch (c.kind) {
    case ILQG_COST_NOMINAL_PATH_LENGTH: {  // src/nominal_path_length_cost.cpp:58-76
      double nom, unused;
      TimeNominal(p, c, int(std::llround(t / p.dt)), &nom, &unused);
      const S delta = S(double(v[c.idx[0]]) - nom);
      G[c.idx[0]] += w * delta;
      H(c.idx[0], c.idx[0]) += w;
      return;
    }
    case ILQG_COST_ROUTE_PROGRESS: {  // src/route_progress_cost.cpp:66-108
      double px, py;
      TimeNominal(p, c, int(std::llround(t / p.dt)), &px, &py);
      const int xi = c.idx[0], yi = c.idx[1];
      G[xi] += w * (v[xi] - S(px));
      G[yi] += w * (v[yi] - S(py));
      H(xi, xi) += w;
      H(yi, yi) += w;
      H(xi, yi) += S(0);
      H(yi, xi) += S(0);
      return;
    }
    case ILQG_COST_QUADRATIC: {  // src/quadratic_cost.cpp:65-94
      if (c.idx[0] >= 0) {
        const int d = c.idx[0];
        const S delta = v[d] - val;
        G[d] += w * delta;
        H(d, d) += w;
      } else {
        for (int i = 0; i < dim; i++) {
          G[i] += w * (v[i] - val);
          H(i, i) = H(i, i) + w;
        }
      }
      return;
    }
    case ILQG_COST_SEMIQUADRATIC: {  // src/semiquadratic_cost.cpp:63-85
      const int d = c.idx[0];
      const S diff = v[d] - val;
      if ((diff < S(0) && oriented) || (diff > S(0) && !oriented)) return;
      G[d] += w * diff;
      H(d, d) += w;
      return;
    }
    case ILQG_COST_QUADRATIC_POLYLINE2: {  // src/quadratic_polyline2_cost.cpp:71-126
      const int xi = c.idx[0], yi = c.idx[1];
      const S px = v[xi], py = v[yi];
      S cx, cy;
      bool is_vertex, is_endpoint;
      Segment2<S> seg;
      p.polylines[c.polyline].ClosestPoint(px, py, &cx, &cy, &is_vertex, &seg, nullptr, &is_endpoint);
      if (is_endpoint) return;
      S ddx = w, ddy = w, dxdy = 0;
      S dx = w * (px - cx), dy = w * (py - cy);
      if (!is_vertex) {
        const S relx = px - seg.p1x, rely = py - seg.p1y;
        ddx = w * seg.uy * seg.uy;
        ddy = w * seg.ux * seg.ux;
        dxdy = -w * seg.ux * seg.uy;
        const S w_cross = w * (relx * seg.uy - rely * seg.ux);
        dx = w_cross * seg.uy;
        dy = -w_cross * seg.ux;
      }
      G[xi] += dx;
      G[yi] += dy;
      H(xi, xi) += ddx;
      H(yi, yi) += ddy;
      H(xi, yi) += dxdy;
      H(yi, xi) += dxdy;
      return;
    }
    case ILQG_COST_SEMIQUADRATIC_POLYLINE2: {  // src/semiquadratic_polyline2_cost.cpp:76-142
      const int xi = c.idx[0], yi = c.idx[1];
      const S px = v[xi], py = v[yi];
      S cx, cy, ssd;
      bool is_vertex, is_endpoint;
      Segment2<S> seg;
      p.polylines[c.polyline].ClosestPoint(px, py, &cx, &cy, &is_vertex, &seg, &ssd, &is_endpoint);
      const S sst = sgn(val) * val * val;
      const bool active = (ssd > sst && oriented) || (ssd < sst && !oriented);
      if (!active) return;
      if (is_endpoint) return;
      S ddx = w, ddy = w, dxdy = 0;
      S scaling = std::sqrt(std::abs(ssd));
      scaling = (scaling - std::abs(val)) / scaling;
      S dx = w * scaling * (px - cx);
      S dy = w * scaling * (py - cy);
      if (!is_vertex) {
        const S relx = px - seg.p1x, rely = py - seg.p1y;
        ddx = w * seg.uy * seg.uy;
        ddy = w * seg.ux * seg.ux;
        dxdy = -w * seg.ux * seg.uy;
        const S w_cross = w * (relx * seg.uy - rely * seg.ux - val);
        dx = w_cross * seg.uy;
        dy = -w_cross * seg.ux;
      }
      G[xi] += dx;
      G[yi] += dy;
      H(xi, xi) += ddx;
      H(yi, yi) += ddy;
      H(xi, yi) += dxdy;
      H(yi, xi) += dxdy;
      return;
    }
    case ILQG_COST_PROXIMITY: {  // src/proximity_cost.cpp:63-122
      const int x1 = c.idx[0], y1 = c.idx[1], x2 = c.idx[2], y2 = c.idx[3];
      const S dx = v[x1] - v[x2], dy = v[y1] - v[y2];
      const S dsq = dx * dx + dy * dy;
      if (dsq >= val * val) return;
      const S delta = std::sqrt(dsq);
      const S gap = val - delta;
      const S wd = w / delta;
      const S dxd = dx / delta, dyd = dy / delta;
      const S ddx1 = -wd * gap * dx;
      const S ddy1 = -wd * gap * dy;
      const S hxx = wd * (dxd * (gap * dxd + dx) - gap);
      const S hyy = wd * (dyd * (gap * dyd + dy) - gap);
      const S hxy = wd * (dxd * (gap * dyd + dy));
      G[x1] += ddx1; G[x2] -= ddx1; G[y1] += ddy1; G[y2] -= ddy1;
      H(x1, x1) += hxx; H(x1, x2) -= hxx; H(x2, x1) -= hxx; H(x2, x2) += hxx;
      H(y1, y1) += hyy; H(y1, y2) -= hyy; H(y2, y1) -= hyy; H(y2, y2) += hyy;
      H(x1, y1) += hxy; H(y1, x1) += hxy;
      H(x1, y2) -= hxy; H(y2, x1) -= hxy;
      H(x2, y1) -= hxy; H(y1, x2) -= hxy;
      H(x2, y2) += hxy; H(y2, x2) += hxy;
      return;
    }
    case ILQG_COST_QUADRATIC_DIFFERENCE: {  // src/quadratic_difference_cost.cpp:61-91
      for (int ii = 0; ii < 2; ii++) {
        const int d1 = c.idx[ii], d2 = c.idx[2 + ii];
        const S dx = w * (v[d1] - v[d2]);
        H(d1, d1) += w; H(d2, d2) += w; H(d1, d2) += -w; H(d2, d1) += -w;
        G[d1] += dx; G[d2] += -dx;
      }
      return;
    }
    case ILQG_COST_POLYLINE2_SIGNED_DISTANCE: {  // src/polyline2_signed_distance_cost.cpp:67-126
      const int xi = c.idx[0], yi = c.idx[1];
      const S px = v[xi], py = v[yi];
      S cx, cy, ssd;
      bool is_vertex, is_endpoint;
      Segment2<S> seg;
      p.polylines[c.polyline].ClosestPoint(px, py, &cx, &cy, &is_vertex, &seg, &ssd, &is_endpoint);
      if (!oriented) ssd *= S(-1);
      const S sign = sgn(ssd);
      const S distance = std::sqrt(std::abs(ssd));
      const S ex = px - cx, ey = py - cy;
      S dx = sign * ex / distance, dy = sign * ey / distance;
      const S denom = ssd * distance;
      S ddx = ey * ey / denom, ddy = ex * ex / denom, dxdy = -ex * ey / denom;
      if (!is_vertex) {  // as written there: the segment normal, whatever the orientation flag
        dx = seg.uy;
        dy = -seg.ux;
        ddx = ddy = dxdy = S(0);
      }
      G[xi] += dx; G[yi] += dy;
      H(xi, xi) += ddx; H(yi, yi) += ddy; H(xi, yi) += dxdy; H(yi, xi) += dxdy;
      return;
    }
    case ILQG_COST_SIGNED_DISTANCE: {  // src/signed_distance_cost.cpp:65-113
      const int x1 = c.idx[0], y1 = c.idx[1], x2 = c.idx[2], y2 = c.idx[3];
      const S s = oriented ? S(1) : S(-1);
      const S ex = v[x1] - v[x2], ey = v[y1] - v[y2];
      const S norm = std::hypot(ex, ey);
      const S norm3 = norm * norm * norm;
      const S dx1 = -s * ex / norm, dy1 = -s * ey / norm;
      const S ddx1 = -s * ey * ey / norm3, ddy1 = -s * ex * ex / norm3;
      const S dxy = s * ex * ey / norm3;
      G[x1] += dx1; G[y1] += dy1; G[x2] -= dx1; G[y2] -= dy1;
      H(x1, x1) += ddx1; H(y1, y1) += ddy1; H(x1, y1) += dxy; H(y1, x1) += dxy;
      H(x2, x2) += ddx1; H(y2, y2) += ddy1; H(x2, y2) += dxy; H(y2, x2) += dxy;
      H(x1, x2) -= ddx1; H(x1, y2) -= dxy; H(y1, x2) -= dxy; H(y1, y2) -= ddy1;
      H(x2, x1) -= ddx1; H(x2, y1) -= dxy; H(y2, x1) -= dxy; H(y2, y1) -= ddy1;
      return;
    }
    case ILQG_COST_EXTREME_VALUE: {  // src/extreme_value_cost.cpp:58-85
      const bool is_min = c.flags & ILQG_FLAG_IS_MIN;
      S ext = is_min ? std::numeric_limits<S>::infinity() : -std::numeric_limits<S>::infinity();
      int best = c.child_begin;
      for (int q = 0; q < c.child_count; q++) {
        const S value = EvaluateTerm(p, c.child_begin + q, v, dim);
        if ((is_min && value < ext) || (!is_min && value > ext)) {
          ext = value;
          best = c.child_begin + q;
        }
      }
      QuadraticizeTerm(p, best, t, v, dim, hess, grad, al);
      return;
    }
    case ILQG_CONSTRAINT_PROXIMITY: {  // src/proximity_constraint.cpp:64-116
      const int x1 = c.idx[0], y1 = c.idx[1], x2 = c.idx[2], y2 = c.idx[3];
      const S dx = v[x1] - v[x2], dy = v[y1] - v[y2];
      const S prox = std::hypot(dx, dy);
      const S sign = oriented ? S(1) : S(-1);
      const S g = sign * (prox - val);
      const S rdx = dx / prox, rdy = dy / prox;
      S gx = sign * rdx, gy = sign * rdy;
      S hxx = sign * (S(1) - rdx * rdx) / prox;
      S hyy = sign * (S(1) - rdy * rdy) / prox;
      S hxy = -sign * rdx * rdy / prox;
      const S lambda = al ? al->lambda(c.constraint_slot, t) : S(0);
      const S mu = al ? al->mu : S(10);
      ModifyDerivatives(lambda, mu, g, &gx, &hxx, &gy, &hyy, &hxy);
      G[x1] += gx; G[x2] -= gx; G[y1] += gy; G[y2] -= gy;
      H(x1, x1) += hxx; H(x1, x2) -= hxx; H(x2, x1) -= hxx; H(x2, x2) += hxx;
      H(y1, y1) += hyy; H(y1, y2) -= hyy; H(y2, y1) -= hyy; H(y2, y2) += hyy;
      H(x1, y1) += hxy; H(x1, y2) -= hxy; H(x2, y1) -= hxy; H(x2, y2) += hxy;
      H(y1, x1) += hxy; H(y1, x2) -= hxy; H(y2, x1) -= hxy; H(y2, x2) += hxy;
      return;
    }
    case ILQG_CONSTRAINT_SINGLE_DIMENSION: {  // single_dimension_constraint.h:74-97
      const int d = c.idx[0];
      const S sign = oriented ? S(1) : S(-1);
      const S g = sign * (v[d] - val);
      S dx = sign, ddx = 0;
      const S lambda = al ? al->lambda(c.constraint_slot, t) : S(0);
      const S mu = al ? al->mu : S(10);
      ModifyDerivatives(lambda, mu, g, &dx, &ddx);
      G[d] += dx;
      H(d, d) += ddx;
      return;
    }
    case ILQG_CONSTRAINT_AFFINE_SCALAR: {  // affine_scalar_constraint.h:70-88
      // grad += lambda a + mu (hess_of_sq x - b a), hess += mu hess_of_sq, hess_of_sq = a a^T formed in the
      // constructor (a matrix of the reference's scalars: every product a_i a_j is rounded before it meets x)
      const float* a = p.dense.data() + c.polyline;
      const S b = S(a[dim]);
      const S g = EvaluateTerm(p, ti, v, dim);
      const S lambda = al ? al->lambda(c.constraint_slot, t) : S(0);
      const S mu = ConstraintMu(lambda, g, al ? al->mu : S(10), (c.flags & ILQG_FLAG_EQUALITY) != 0);
      for (int i = 0; i < dim; i++) {
        S hx = S(0);
        for (int j = 0; j < dim; j++) hx += S(S(a[i]) * S(a[j])) * v[j];
        G[i] += lambda * S(a[i]) + mu * (hx - b * S(a[i]));
        for (int j = 0; j < dim; j++) H(i, j) += mu * S(S(a[i]) * S(a[j]));
      }
      return;
    }
    case ILQG_CONSTRAINT_AFFINE_VECTOR: {  // affine_vector_constraint.h:77-101, as written there
      const float* A = p.dense.data() + c.polyline;
      const float* b = A + dim * dim;
      Vec<S> delta(dim), atd(dim);
      S sq = S(0);
      for (int i = 0; i < dim; i++) {
        S d = S(0);
        for (int j = 0; j < dim; j++) d += S(A[i + dim * j]) * v[j];
        delta[i] = d - S(b[i]);
        sq += delta[i] * delta[i];
      }
      const S value = std::sqrt(sq);
      for (int i = 0; i < dim; i++) {  // A^T delta
        S d = S(0);
        for (int j = 0; j < dim; j++) d += S(A[j + dim * i]) * delta[j];
        atd[i] = d;
      }
      const S lambda = al ? al->lambda(c.constraint_slot, t) : S(0);
      const S mu = ConstraintMu(lambda, value, al ? al->mu : S(10), (c.flags & ILQG_FLAG_EQUALITY) != 0);
      const S lv = lambda / value;
      for (int i = 0; i < dim; i++) {
        G[i] += (mu + lv) * atd[i];
        for (int j = 0; j < dim; j++) {
          S ata = S(0), aat = S(0);  // ATA_ = A^T A, AAT_ = A A^T (constructor; :60-61)
          for (int q = 0; q < dim; q++) {
            ata += S(A[q + dim * i]) * S(A[q + dim * j]);
            aat += S(A[i + dim * q]) * S(A[j + dim * q]);
          }
          H(i, j) += lv * (aat - atd[i] * atd[j] / (value * value)) + mu * ata;
        }
      }
      return;
    }
    case ILQG_COST_ORIENTATION: {  // src/orientation_cost.cpp:60-80
      const int d = c.idx[0];
      const S diff = S(std::fmod(double(v[d] - val) + M_PI, M_PI * 2.0) - M_PI);
      G[d] += w * diff;
      H(d, d) += w;
      return;
    }
    case ILQG_COST_QUADRATIC_NORM:        // src/quadratic_norm_cost.cpp:59-94
    case ILQG_COST_SEMIQUADRATIC_NORM: {  // src/semiquadratic_norm_cost.cpp:61-99
      const int d1 = c.idx[0], d2 = c.idx[1];
      const S x = v[d1], y = v[d2];
      const bool semi = c.kind == ILQG_COST_SEMIQUADRATIC_NORM;
      const S norm_sq = x * x + y * y;
      const S norm = semi ? std::hypot(x, y) : std::sqrt(norm_sq);
      if (semi && ((norm > val && !oriented) || (norm < val && oriented))) return;
      const S norm3 = semi ? norm * (norm * norm) : norm * norm_sq;
      const S dx = -w * x * (S(-1) + val / norm), dy = -w * y * (S(-1) + val / norm);
      const S ddx = w - (val * y * y * w) / norm3, ddy = w - (val * x * x * w) / norm3;
      const S dxdy = val * x * y * w / norm3;
      G[d1] += dx; G[d2] += dy;
      H(d1, d1) += ddx; H(d2, d2) += ddy; H(d1, d2) += dxdy; H(d2, d1) += dxdy;
      return;
    }
    case ILQG_COST_RELATIVE_DISTANCE: {  // src/relative_distance_cost.cpp:56-104
      const int x1 = c.idx[0], y1 = c.idx[1], x2 = c.idx[2], y2 = c.idx[3];
      const S ex = v[x1] - v[x2], ey = v[y1] - v[y2];
      const S dist = std::hypot(ex, ey);
      const S dist3 = dist * dist * dist;
      const S ddx = w * ey * ey / dist3, ddy = w * ex * ex / dist3, dxdy = -w * ex * ey / dist3;
      H(x1, x1) += ddx; H(x1, y1) += dxdy; H(y1, x1) += dxdy; H(y1, y1) += ddy;
      H(x2, x2) += ddx; H(x2, y2) += dxdy; H(y2, x2) += dxdy; H(y2, y2) += ddy;
      H(x1, x2) -= ddx; H(x1, y2) -= dxdy; H(y1, x2) -= dxdy; H(y1, y2) -= ddy;
      H(x2, x1) -= ddx; H(x2, y1) -= dxdy; H(y2, x1) -= dxdy; H(y2, y1) -= ddy;
      const S dx = w * ex / dist, dy = w * ey / dist;
      G[x1] += dx; G[y1] += dy; G[x2] -= dx; G[y2] -= dy;
      return;
    }
    case ILQG_COST_LOCALLY_CONVEX_PROXIMITY: {  // src/locally_convex_proximity_cost.cpp:62-108
      const int x1 = c.idx[0], y1 = c.idx[1], x2 = c.idx[2], y2 = c.idx[3];
      const S dx = v[x1] - v[x2], dy = v[y1] - v[y2];
      if (dx * dx >= val * val || dy * dy >= val * val) return;
      const S delta_x = val - std::abs(dx), delta_y = val - std::abs(dy);
      if (delta_x * delta_x < delta_y * delta_y) {  // as written there: the gradient carries no sign(dx)
        const S dx1 = -w * delta_x;
        G[x1] += dx1; G[x2] -= dx1;
        H(x1, x1) += w; H(x2, x2) += w; H(x1, x2) -= w; H(x2, x1) -= w;
      } else {
        const S dy1 = -w * delta_y;
        G[y1] += dy1; G[y2] -= dy1;
        H(y1, y1) += w; H(y2, y2) += w; H(y1, y2) -= w; H(y2, y1) -= w;
      }
      return;
    }
    case ILQG_COST_WEIGHTED_CONVEX_PROXIMITY: {  // src/weighted_convex_proximity_cost.cpp:63-158, as written there
      const int x1 = c.idx[0], y1 = c.idx[1], x2 = c.idx[2], y2 = c.idx[3], i1 = c.idx_extra[0], i2 = c.idx_extra[1];
      const S dx = v[x1] - v[x2], dy = v[y1] - v[y2];
      const S v1 = v[i1], v2 = v[i2];
      const S vv = v1 * v1 + v2 * v2;
      if (dx * dx >= val * val || dy * dy >= val * val) return;
      const S delta_x = val - std::abs(dx), delta_y = val - std::abs(dy);
      const bool is_x_active = delta_x * delta_x < delta_y * delta_y;
      const int p1 = is_x_active ? x1 : y1, p2 = is_x_active ? x2 : y2;  // the two branches differ in the axis only
      const S delta = is_x_active ? delta_x : delta_y, d = is_x_active ? dx : dy;
      const S dp1 = -w * delta * vv;
      const S dv1 = -w * v1 * delta * delta, dv2 = -w * v2 * delta * delta;
      const S ddp1 = w, ddv1 = w * delta * delta, ddv2 = ddv1, dv1dv2 = S(0);
      const S dp1dv1 = S(-2.0) * w * v1 * sgn(d), dp1dv2 = S(-2.0) * w * v2 * sgn(d);
      H(p1, p1) += ddp1; H(p1, p2) -= ddp1; H(p2, p1) -= ddp1; H(p2, p2) += ddp1;
      H(p1, i1) += dp1dv1; H(p1, i2) += dp1dv2; H(p2, i1) -= dp1dv1; H(p2, i2) -= dp1dv2;
      H(i1, p1) += dp1dv1; H(i1, p2) -= dp1dv1; H(i1, i1) += ddv1; H(i1, i2) += dv1dv2;
      H(i2, p1) += dp1dv2; H(i2, p2) -= dp1dv2; H(i2, i1) += dv1dv2; H(i2, i2) += ddv2;
      G[p1] += dp1; G[p2] -= dp1; G[i1] += dv1; G[i2] += dv2;
      return;
    }
    case ILQG_COST_CURVATURE: {  // src/curvature_cost.cpp:55-86
      const int oi = c.idx[0], vi = c.idx[1];
      const S vel = v[vi], omega = v[oi];
      const S one_over_vsq = S(1) / (vel * vel);
      const S weight_over_vsq = w * one_over_vsq;
      const S weight_omega_over_vsq = omega * weight_over_vsq;
      G[oi] += weight_omega_over_vsq;
      G[vi] += -weight_omega_over_vsq * omega / vel;
      H(oi, oi) += weight_over_vsq;
      H(oi, vi) += S(-2) * weight_omega_over_vsq / vel;
      H(vi, oi) += S(-2) * weight_omega_over_vsq / vel;
      H(vi, vi) += S(3) * weight_omega_over_vsq * omega * one_over_vsq;
      return;
    }
    case ILQG_CONSTRAINT_POLYLINE2_SIGNED_DISTANCE: {  // src/polyline2_signed_distance_constraint.cpp:65-144
      const int xi = c.idx[0], yi = c.idx[1];
      const S x = v[xi], y = v[yi];
      S cx, cy, ssd;
      bool is_vertex;
      Segment2<S> seg;
      p.polylines[c.polyline].ClosestPoint(x, y, &cx, &cy, &is_vertex, &seg, &ssd, nullptr);
      const S s = sgn(ssd);
      const S sign = oriented ? S(1) : S(-1);
      const S sd = sgn(ssd) * std::sqrt(std::abs(ssd));
      const S g = oriented ? sd - val : val - sd;
      S dx = sign * seg.uy, ddx = 0, dy = -sign * seg.ux, ddy = 0, dxdy = 0;
      if (is_vertex) {
        const S px = cx, py = cy;
        const S rx = x - px, ry = y - py;
        const S d_sq = rx * rx + ry * ry;
        const S d = std::sqrt(d_sq);
        dx = sign * s * rx / d;
        ddx = sign * s * (d_sq - px * px - x * x + S(2) * px * x) / (d_sq * d);
        dxdy = -sign * s * rx * ry / (d_sq * d);
        dy = sign * s * ry / d;
        ddy = sign * s * (d_sq - py * py - y * y + S(2) * py * y) / (d_sq * d);
      }
      const S lambda = al ? al->lambda(c.constraint_slot, t) : S(0);
      const S mu = al ? al->mu : S(10);
      ModifyDerivatives(lambda, mu, g, &dx, &ddx, &dy, &ddy, &dxdy);
      G[xi] += dx; G[yi] += dy;
      H(xi, xi) += ddx; H(xi, yi) += dxdy; H(yi, xi) += dxdy; H(yi, yi) += ddy;
      return;
    }
  }
}

// QuadraticCostApproximation for one (k, player): Q, l and the pair blocks.
template <class S>
struct Quad {
  Mat<S> Q;
  Vec<S> l;
  std::vector<Mat<S>> R;   // per pair of the problem table (unused pairs stay empty)
  std::vector<Vec<S>> r;
  std::vector<char> has;   // control.find(j) != end
};

// PlayerCost::Quadraticize / QuadraticizeControlCosts, src/player_cost.cpp:194-225
// (with AccumulateControlCostsBase :59-86 creating sigma_u*I blocks at first touch).
template <class S>
Quad<S> QuadraticizePlayer(const Problem<S>& p, int i, double t, const Vec<S>& x, const Vec<S>& u,
                           bool control_only, const ALState<S>* al) {
  Quad<S> q;
  q.Q = Mat<S>::Identity(p.n, S(p.pcs[i].state_regularization));
  q.l.assign(p.n, S(0));
  const int np = (int)p.pairs.size();
  q.R.resize(np);
  q.r.resize(np);
  q.has.assign(np, 0);
  auto touch = [&](int j) {
    const int pi = p.pairIndex(i, j);
    if (!q.has[pi]) {
      q.R[pi] = Mat<S>::Identity(p.udim(j), S(p.pcs[i].control_regularization));
      q.r[pi].assign(p.udim(j), S(0));
      q.has[pi] = 1;
    }
    return pi;
  };
  const int nt = (int)p.terms.size();
  if (!control_only)
    for (int ti = 0; ti < nt; ti++)
      if (p.terms[ti].player == i && p.terms[ti].role == ILQG_ROLE_STATE_COST)
        QuadraticizeTerm(p, ti, t, x.data(), p.n, &q.Q, &q.l, al);
  for (int ti = 0; ti < nt; ti++)
    if (p.terms[ti].player == i && p.terms[ti].role == ILQG_ROLE_CONTROL_COST) {
      const int j = p.terms[ti].arg, pi = touch(j);
      QuadraticizeTerm(p, ti, t, &u[p.uoff[j]], p.udim(j), &q.R[pi], &q.r[pi], al);
    }
  if (!control_only) {
    for (int ti = 0; ti < nt; ti++)
      if (p.terms[ti].player == i && p.terms[ti].role == ILQG_ROLE_STATE_CONSTRAINT)
        QuadraticizeTerm(p, ti, t, x.data(), p.n, &q.Q, &q.l, al);
    for (int ti = 0; ti < nt; ti++)
      if (p.terms[ti].player == i && p.terms[ti].role == ILQG_ROLE_CONTROL_CONSTRAINT) {
        const int j = p.terms[ti].arg, pi = touch(j);
        QuadraticizeTerm(p, ti, t, &u[p.uoff[j]], p.udim(j), &q.R[pi], &q.r[pi], al);
      }
  }
  return q;
}

// PlayerCost::Evaluate(t, x, us), src/player_cost.cpp:128-144 (constraints excluded).
// k_state / k_control: the time steps the state / control costs are evaluated AT (FinalTimeCost::Evaluate,
// cost/final_time_cost.h:66-69, returns 0 before its threshold step); the default is "late enough for every term".
template <class S>
S EvaluatePlayer(const Problem<S>& p, int i, const Vec<S>& x, const Vec<S>& u, int k_state = 1 << 30,
                 int k_control = 1 << 30) {
  S total = 0;
  const int nt = (int)p.terms.size();
  for (int ti = 0; ti < nt; ti++)
    if (p.terms[ti].player == i && p.terms[ti].role == ILQG_ROLE_STATE_COST && k_state >= p.terms[ti].first_step)
      total += EvaluateTerm(p, ti, x.data(), p.n, k_state);
  for (int ti = 0; ti < nt; ti++)
    if (p.terms[ti].player == i && p.terms[ti].role == ILQG_ROLE_CONTROL_COST &&
        k_control >= p.terms[ti].first_step) {
      const int j = p.terms[ti].arg;
      total += EvaluateTerm(p, ti, &u[p.uoff[j]], p.udim(j), k_control);
    }
  return total;
}

// ---------------------------------------------------------------------------
// Containers for one instance
// ---------------------------------------------------------------------------
template <class S>
struct Trajectory {  // OperatingPoint, operating_point.h:55-85 (us stacked per step)
  std::vector<Vec<S>> xs, us;
  Trajectory() {}
  Trajectory(int T, int n, int m) : xs(T, Vec<S>(n, S(0))), us(T, Vec<S>(m, S(0))) {}
};
template <class S>
struct Strategies {  // vector<Strategy>, strategy.h:59-85 — stacked (m x n) gain per step
  std::vector<Mat<S>> P;
  std::vector<Vec<S>> alpha;
  Strategies() {}
  Strategies(int T, int n, int m) : P(T, Mat<S>(m, n)), alpha(T, Vec<S>(m, S(0))) {}
};
template <class S>
struct LQInputs {
  std::vector<Mat<S>> A, B;           // [T]
  std::vector<std::vector<Quad<S>>> q;  // [T][N]
};

// ---------------------------------------------------------------------------
// LQFeedbackSolver::Solve, src/lq_feedback_solver.cpp:71-244
// ---------------------------------------------------------------------------
template <class S>
void SolveLQFeedback(const Problem<S>& p, const LQInputs<S>& in, const Vec<S>& x0,
                     bool adaptive_regularization, Strategies<S>* out, std::vector<Vec<S>>* delta_xs,
                     std::vector<std::vector<Vec<S>>>* costates) {
  const int T = (int)in.A.size(), N = p.N, n = p.n, m = p.m;
  *out = Strategies<S>(T, n, m);
  std::vector<std::vector<Mat<S>>> Zs(T, std::vector<Mat<S>>(N));
  std::vector<std::vector<Vec<S>>> zetas(T, std::vector<Vec<S>>(N));
  for (int i = 0; i < N; i++) {  // :102-105
    Zs[T - 1][i] = in.q[T - 1][i].Q;
    zetas[T - 1][i] = in.q[T - 1][i].l;
  }
  for (int k = T - 2; k >= 0; k--) {  // :110
    const Mat<S>& A = in.A[k];
    const Mat<S>& B = in.B[k];
    const auto& quad = in.q[k];
    Mat<S> Sm(m, m), Y(m, n + 1);
    for (int i = 0; i < N; i++) {
      const int ro = p.uoff[i], mi = p.udim(i);
      // BiZi = B_i^T Z_i  (:128)
      Mat<S> Bi(n, mi);
      for (int c = 0; c < mi; c++)
        for (int rr = 0; rr < n; rr++) Bi(rr, c) = B(rr, ro + c);
      const Mat<S> BiZi = matmulTN(Bi, Zs[k + 1][i]);
      const Mat<S> Srow = matmul(BiZi, B);  // all column blocks at once (:131-149)
      const int pii = p.pairIndex(i, i);
      for (int a = 0; a < mi; a++)
        for (int c = 0; c < m; c++) {
          S v = Srow(a, c);
          if (c >= ro && c < ro + mi) v = v + quad[i].R[pii](a, c - ro);
          Sm(ro + a, c) = v;
        }
      const Mat<S> Yrow = matmul(BiZi, A);  // :152-153
      const Vec<S> yz = matvecT(Bi, zetas[k + 1][i]);  // :154-157
      for (int a = 0; a < mi; a++) {
        for (int c = 0; c < n; c++) Y(ro + a, c) = Yrow(a, c);
        Y(ro + a, n) = yz[a] + quad[i].r[pii][a];
      }
    }
    if (adaptive_regularization) {  // :163-176 (column-wise, in place, sequential)
      for (int c = 0; c < m; c++) {
        S l1 = 0;
        for (int rr = 0; rr < m; rr++) l1 += std::abs(Sm(rr, c));
        const S radius = l1 - std::abs(Sm(c, c));
        const S eval_lo = Sm(c, c) - radius;
        const S min_eval = S(1e-3f);
        if (eval_lo < min_eval) Sm(c, c) += radius + min_eval;
      }
    }
    HouseholderQR<S> qr;  // :180
    qr.compute(Sm);
    const Mat<S> X = qr.solve(Y);
    Mat<S>& P = out->P[k];
    Vec<S>& alpha = out->alpha[k];
    for (int a = 0; a < m; a++) {
      for (int c = 0; c < n; c++) P(a, c) = X(a, c);
      alpha[a] = X(a, n);
    }
    // F = A - sum B_i P_i ; beta = - sum B_i alpha_i  (:189-194)
    Mat<S> F = A;
    const Mat<S> BP = matmul(B, P);
    for (size_t e = 0; e < F.d.size(); e++) F.d[e] -= BP.d[e];
    Vec<S> beta = matvec(B, alpha);
    for (auto& b : beta) b = -b;
    for (int i = 0; i < N; i++) {  // :198-212
      Vec<S> tmp = matvec(Zs[k + 1][i], beta);
      for (int e = 0; e < n; e++) tmp[e] += zetas[k + 1][i][e];
      Vec<S> zeta = matvecT(F, tmp);
      for (int e = 0; e < n; e++) zeta[e] += quad[i].l[e];
      Mat<S> Z = matmulTN(F, matmul(Zs[k + 1][i], F));
      for (size_t e = 0; e < Z.d.size(); e++) Z.d[e] += quad[i].Q.d[e];
      for (size_t q = 0; q < p.pairs.size(); q++) {
        if (p.pairs[q].i != i || !quad[i].has[q]) continue;
        const int j = p.pairs[q].j, jo = p.uoff[j], mj = p.udim(j);
        Mat<S> Pj(mj, n);
        Vec<S> aj(mj);
        for (int a = 0; a < mj; a++) {
          aj[a] = alpha[jo + a];
          for (int c = 0; c < n; c++) Pj(a, c) = P(jo + a, c);
        }
        Vec<S> w = matvec(quad[i].R[q], aj);
        for (int a = 0; a < mj; a++) w[a] -= quad[i].r[q][a];
        const Vec<S> add = matvecT(Pj, w);
        for (int e = 0; e < n; e++) zeta[e] += add[e];
        const Mat<S> PRP = matmulTN(Pj, matmul(quad[i].R[q], Pj));
        for (size_t e = 0; e < Z.d.size(); e++) Z.d[e] += PRP.d[e];
      }
      Zs[k][i] = Z;
      zetas[k][i] = zeta;
    }
  }
  if (delta_xs) {  // forward pass, :217-241 (feedback term deliberately absent)
    delta_xs->assign(T, Vec<S>(n, S(0)));
    if (costates) costates->assign(T, std::vector<Vec<S>>(N, Vec<S>(n, S(0))));
    Vec<S> xstar = x0;
    for (int k = 0; k < T; k++) {
      (*delta_xs)[k] = xstar;
      if (costates)
        for (int i = 0; i < N; i++)
          if (k < T - 1) {
            Vec<S> c = matvec(Zs[k + 1][i], xstar);
            for (int e = 0; e < n; e++) c[e] = -c[e] - zetas[k + 1][i][e];
            (*costates)[k][i] = c;
          }
      Vec<S> nx = matvec(in.A[k], xstar);
      const Vec<S> ba = matvec(in.B[k], out->alpha[k]);
      for (int e = 0; e < n; e++) nx[e] -= ba[e];
      xstar = nx;
    }
  }
}

// ---------------------------------------------------------------------------
// LQOpenLoopSolver::Solve, src/lq_open_loop_solver.cpp:73-195
// ---------------------------------------------------------------------------
template <class S>
void SolveLQOpenLoop(const Problem<S>& p, const LQInputs<S>& in, const Vec<S>& x0, Strategies<S>* out,
                     std::vector<Vec<S>>* delta_xs, std::vector<std::vector<Vec<S>>>* costates) {
  const int T = (int)in.A.size(), N = p.N, n = p.n, m = p.m;
  *out = Strategies<S>(T, n, m);
  std::vector<std::vector<Mat<S>>> Ms(T, std::vector<Mat<S>>(N)), warpedB(T, std::vector<Mat<S>>(N));
  std::vector<std::vector<Vec<S>>> ms(T, std::vector<Vec<S>>(N)), warpedr(T, std::vector<Vec<S>>(N));
  std::vector<HouseholderQR<S>> qrs(T);
  std::vector<Vec<S>> inter(T, Vec<S>(n, S(0)));
  for (int i = 0; i < N; i++) {  // :105-108
    ms[T - 1][i] = in.q[T - 1][i].l;
    Ms[T - 1][i] = in.q[T - 1][i].Q;
  }
  for (int k = T - 2; k >= 0; k--) {  // :113-151
    const Mat<S>& A = in.A[k];
    const Mat<S>& B = in.B[k];
    const auto& quad = in.q[k];
    Mat<S> Lam = Mat<S>::Identity(n);
    std::vector<Mat<S>> Bis(N);
    for (int i = 0; i < N; i++) {
      const int ro = p.uoff[i], mi = p.udim(i), pii = p.pairIndex(i, i);
      Mat<S> Bi(n, mi), BiT(mi, n);
      for (int c = 0; c < mi; c++)
        for (int rr = 0; rr < n; rr++) { Bi(rr, c) = B(rr, ro + c); BiT(c, rr) = B(rr, ro + c); }
      Bis[i] = Bi;
      LDLT<S> chol;
      chol.compute(quad[i].R[pii]);
      warpedB[k][i] = chol.solve(BiT);
      Mat<S> rcol(mi, 1);
      for (int a = 0; a < mi; a++) rcol(a, 0) = quad[i].r[pii][a];
      const Mat<S> wr = chol.solve(rcol);
      warpedr[k][i].assign(mi, S(0));
      for (int a = 0; a < mi; a++) warpedr[k][i][a] = wr(a, 0);
      const Mat<S> add = matmul(matmul(Bi, warpedB[k][i]), Ms[k + 1][i]);
      for (size_t e = 0; e < Lam.d.size(); e++) Lam.d[e] += add.d[e];
    }
    qrs[k].compute(Lam);
    for (int i = 0; i < N; i++) {
      Vec<S> w = matvec(warpedB[k][i], ms[k + 1][i]);
      for (int a = 0; a < p.udim(i); a++) w[a] += warpedr[k][i][a];
      const Vec<S> bw = matvec(Bis[i], w);
      for (int e = 0; e < n; e++) inter[k][e] -= bw[e];
    }
    const Mat<S> LinvA = qrs[k].solve(A);
    Mat<S> ic(n, 1);
    for (int e = 0; e < n; e++) ic(e, 0) = inter[k][e];
    const Mat<S> Linvc = qrs[k].solve(ic);
    Vec<S> lc(n);
    for (int e = 0; e < n; e++) lc[e] = Linvc(e, 0);
    for (int i = 0; i < N; i++) {
      Mat<S> M = matmulTN(A, matmul(Ms[k + 1][i], LinvA));
      for (size_t e = 0; e < M.d.size(); e++) M.d[e] += quad[i].Q.d[e];
      Vec<S> tmp = matvec(Ms[k + 1][i], lc);
      for (int e = 0; e < n; e++) tmp[e] += ms[k + 1][i][e];
      Vec<S> mm = matvecT(A, tmp);
      for (int e = 0; e < n; e++) mm[e] += quad[i].l[e];
      Ms[k][i] = M;
      ms[k][i] = mm;
    }
  }
  if (delta_xs) delta_xs->assign(T, Vec<S>(n, S(0)));
  if (costates) costates->assign(T, std::vector<Vec<S>>(N, Vec<S>(n, S(0))));
  Vec<S> xstar = x0;
  for (int k = 0; k < T - 1; k++) {  // :156-185
    if (delta_xs) (*delta_xs)[k] = xstar;
    Vec<S> rhs = matvec(in.A[k], xstar);
    Mat<S> rc(n, 1);
    for (int e = 0; e < n; e++) rc(e, 0) = rhs[e] + inter[k][e];
    const Mat<S> sol = qrs[k].solve(rc);
    for (int e = 0; e < n; e++) xstar[e] = sol(e, 0);
    for (int i = 0; i < N; i++) {
      Vec<S> it = matvec(Ms[k + 1][i], xstar);
      for (int e = 0; e < n; e++) it[e] += ms[k + 1][i][e];
      const Vec<S> a = matvec(warpedB[k][i], it);
      for (int q = 0; q < p.udim(i); q++) out->alpha[k][p.uoff[i] + q] = a[q] + warpedr[k][i][q];
      if (costates) (*costates)[k][i] = matvecT(in.A[k], it);
    }
  }
  if (delta_xs) delta_xs->back() = xstar;  // :188-192
}

// ---------------------------------------------------------------------------
// ILQSolver pieces, src/ilq_solver.cpp
// ---------------------------------------------------------------------------
// ILQSolver::CurrentOperatingPoint, :174-206 (Strategy::operator(), strategy.h:73-76)
template <class S>
void Rollout(const Problem<S>& p, const Vec<S>& x0, const Trajectory<S>& last, const Strategies<S>& st,
             Trajectory<S>* cur, bool euler = false) {
  const int T = p.T, n = p.n, m = p.m;
  if ((int)cur->xs.size() != T) *cur = Trajectory<S>(T, n, m);
  Vec<S> x = x0;
  for (int k = 0; k < T; k++) {
    const double t = double(k) * p.dt;
    Vec<S> dx(n);
    for (int e = 0; e < n; e++) dx[e] = x[e] - last.xs[k][e];
    cur->xs[k] = x;
    const Vec<S> Pdx = matvec(st.P[k], dx);
    for (int a = 0; a < m; a++) cur->us[k][a] = last.us[k][a] - Pdx[a] - st.alpha[k][a];
    if (k < T - 1) x = Integrate(p, t, p.dt, x, cur->us[k], euler);
  }
}

// ILQSolver::ComputeLinearization, :437-455
template <class S>
void ComputeLinearization(const Problem<S>& p, const Trajectory<S>& op, LQInputs<S>* lq) {
  lq->A.resize(p.T);
  lq->B.resize(p.T);
  for (int k = 0; k < p.T; k++) Linearize(p, op.xs[k], op.us[k], &lq->A[k], &lq->B[k]);
}

// ILQSolver::ComputeCostQuadraticization, :471-490
template <class S>
void ComputeQuadraticization(const Problem<S>& p, const Trajectory<S>& op, const std::vector<int>& t_extreme,
                             const ALState<S>* al, LQInputs<S>* lq) {
  lq->q.assign(p.T, std::vector<Quad<S>>(p.N));
  for (int k = 0; k < p.T; k++) {
    const double t = double(k) * p.dt;
    for (int i = 0; i < p.N; i++) {
      const bool full = p.pcs[i].structure == ILQG_SUM || t_extreme[i] == k;
      lq->q[k][i] = QuadraticizePlayer(p, i, t, op.xs[k], op.us[k], !full, al);
    }
  }
}

// ILQSolver::TotalCosts, :220-257
template <class S>
void TotalCosts(const Problem<S>& p, const Trajectory<S>& op, Vec<S>* costs, std::vector<int>* t_extreme) {
  costs->assign(p.N, S(0));
  for (int i = 0; i < p.N; i++) {
    if (p.pcs[i].structure == ILQG_MAX) (*costs)[i] = -std::numeric_limits<S>::infinity();
    if (p.pcs[i].structure == ILQG_MIN) (*costs)[i] = std::numeric_limits<S>::infinity();
  }
  for (int k = 0; k < p.T; k++)
    for (int i = 0; i < p.N; i++) {
      const S c = EvaluatePlayer(p, i, op.xs[k], op.us[k], k, k);
      if (p.pcs[i].structure == ILQG_SUM)
        (*costs)[i] += c;
      else if (p.pcs[i].structure == ILQG_MAX && c > (*costs)[i]) {
        (*costs)[i] = c;
        (*t_extreme)[i] = k;
      } else if (p.pcs[i].structure == ILQG_MIN && c < (*costs)[i]) {
        (*costs)[i] = c;
        (*t_extreme)[i] = k;
      }
    }
}

// ILQSolver::MeritFunction (:400-435) minus the re-quadraticization it triggers.
template <class S>
S MeritFromQuad(const Problem<S>& p, const LQInputs<S>& lq) {
  S merit = 0;
  for (int k = 0; k < p.T; k++)
    for (int i = 0; i < p.N; i++) {
      const Quad<S>& q = lq.q[k][i];
      const int pii = p.pairIndex(i, i);
      S s = 0;
      for (S v : q.r[pii]) s += v * v;
      merit += s;
      if (k > 0) {
        S s2 = 0;
        for (S v : q.l) s2 += v * v;
        merit += s2;
      }
    }
  return S(0.5) * merit;
}

// ILQSolver::ExpectedDecrease, :364-398
template <class S>
S ExpectedDecrease(const Problem<S>& p, const LQInputs<S>& lq, const Strategies<S>& st,
                   const std::vector<Vec<S>>& delta_xs) {
  S ed = 0;
  for (int k = 0; k < p.T; k++)
    for (int i = 0; i < p.N; i++) {
      const Quad<S>& q = lq.q[k][i];
      const int pii = p.pairIndex(i, i), mi = p.udim(i), uo = p.uoff[i];
      // neg_ui^T * R_ii * r_ii   (Eigen evaluates (a^T R) r left to right)
      S acc = 0;
      for (int c = 0; c < mi; c++) {
        S aR = 0;
        for (int a = 0; a < mi; a++) aR += st.alpha[k][uo + a] * q.R[pii](a, c);
        acc += aR * q.r[pii][c];
      }
      ed -= acc;
      if (k > 0) {
        S acc2 = 0;
        for (int c = 0; c < p.n; c++) {
          S dQ = 0;
          for (int a = 0; a < p.n; a++) dQ += delta_xs[k][a] * q.Q(a, c);
          acc2 += dQ * q.l[c];
        }
        ed -= acc2;
      }
    }
  return ed;
}

// Per-iteration record (what tests compare against the device path).
template <class S>
struct IterLog {
  S merit, expected_decrease, step;
  int backtracks;
  Vec<S> costs;
};

// Persistent solver state: ILQSolver members last_merit_function_value_,
// expected_decrease_ (ilq_solver.h:73-74,189-190) survive across Solve() calls.
template <class S>
struct ILQState {
  S last_merit = std::numeric_limits<S>::infinity();
  S expected_decrease = std::numeric_limits<S>::infinity();
  std::vector<int> t_extreme;  // PlayerCost::time_of_extreme_cost_, player_cost.h:70
};

// ILQSolver::Solve, :76-172, with ModifyLQStrategies :289-348 inlined.
// `fixed_iters` > 0: run exactly that many outer iterations (ignore has_converged).
// `forced_steps` (test mode, needs fixed_iters > 0): iteration q takes the step forced_steps[q] — the strategies are
// scaled by it, the trajectory is rolled out and quadraticised ONCE and accepted whatever CheckArmijoCondition
// says; merit / convergence bookkeeping as after an accepted step.  This removes the only data-dependent branch of
// the loop, so that a device solve can be compared iterate by iterate.
// If `raw` is non-null it receives the unscaled LQ strategies of the LAST LQ solve
// (the object P_t / alpha_t parity is defined on).
// One CheckArmijoCondition call (src/ilq_solver.cpp:350-362), for diagnosis: which inequality was tested with what.
template <class S>
struct ArmijoTrace {
  int iteration, backtrack, accepted;
  S step, last_merit, merit, expected_decrease, scaled;  // scaled = fraction * step * expected_decrease
};

template <class S>
bool SolveILQ(const Problem<S>& p, const Vec<S>& x0, Trajectory<S>* op_io, Strategies<S>* st_io,
              ILQState<S>* state, const ALState<S>* al, int fixed_iters, std::vector<IterLog<S>>* log,
              Vec<S>* final_costs, int* iters_out, int* converged_out, Strategies<S>* raw = nullptr,
              int* logged_out = nullptr, const S* forced_steps = nullptr, std::vector<ArmijoTrace<S>>* trace = nullptr) {
  const ilqg_solver_params& prm = p.params;
  if ((int)state->t_extreme.size() != p.N) state->t_extreme.assign(p.N, 0);
  Trajectory<S> last_op = *op_io, cur_op = *op_io;
  cur_op.xs[0] = x0;
  last_op.xs[0] = x0;
  Strategies<S> strategies = *st_io;
  int num_iterations = 0;
  bool has_converged = false;
  Vec<S> total_costs;
  std::swap(last_op, cur_op);  // :100
  Rollout(p, x0, last_op, strategies, &cur_op);
  TotalCosts(p, cur_op, &total_costs, &state->t_extreme);
  // What the returned SolverLog ends with (log->AddSolverIterate, :111,164): the
  // last ACCEPTED iterate; a failed line search returns the log as it stood.
  Trajectory<S> logged_op = cur_op;
  Strategies<S> logged_st = strategies;
  Vec<S> logged_costs = total_costs;
  LQInputs<S> lq;
  ComputeQuadraticization(p, cur_op, state->t_extreme, al, &lq);
  std::vector<Vec<S>> delta_xs;
  const int max_iters = fixed_iters > 0 ? fixed_iters : prm.max_solver_iters;
  bool ok = true;
  while (num_iterations < max_iters && (fixed_iters > 0 || !has_converged)) {
    num_iterations++;
    ComputeLinearization(p, cur_op, &lq);
    Vec<S> dx0(p.n);
    for (int e = 0; e < p.n; e++) dx0[e] = x0[e] - cur_op.xs[0][e];
    if (prm.open_loop)
      SolveLQOpenLoop(p, lq, dx0, &strategies, &delta_xs, (std::vector<std::vector<Vec<S>>>*)nullptr);
    else
      SolveLQFeedback(p, lq, dx0, true, &strategies, &delta_xs, (std::vector<std::vector<Vec<S>>>*)nullptr);
    if (raw) *raw = strategies;
    // ---- ModifyLQStrategies ----
    state->expected_decrease = ExpectedDecrease(p, lq, strategies, delta_xs);
    auto scale = [&](S s) {
      for (auto& a : strategies.alpha)
        for (auto& v : a) v *= s;
    };
    const S first_step = forced_steps ? forced_steps[num_iterations - 1] : S(prm.initial_alpha_scaling);
    scale(first_step);
    const Trajectory<S> last = cur_op;
    S step = first_step;
    Rollout(p, last.xs[0], last, strategies, &cur_op);
    int backtracks = 0;
    S merit = std::numeric_limits<S>::quiet_NaN();
    bool accepted = !prm.linesearch;
    if (forced_steps && prm.linesearch) {
      ComputeQuadraticization(p, cur_op, state->t_extreme, al, &lq);
      merit = MeritFromQuad(p, lq);
      has_converged = (merit <= state->last_merit) &&
                      std::abs(state->last_merit - merit) < S(prm.convergence_tolerance);
      state->last_merit = merit;
      accepted = true;
    } else if (prm.linesearch) {
      for (int bb = 0; bb < prm.max_backtracking_steps; bb++) {
        ComputeQuadraticization(p, cur_op, state->t_extreme, al, &lq);  // MeritFunction :405
        merit = MeritFromQuad(p, lq);
        const S scaled = S(prm.expected_decrease_fraction) * step * state->expected_decrease;
        if (trace)
          trace->push_back({num_iterations, bb, (state->last_merit - merit >= scaled) ? 1 : 0, step, state->last_merit, merit,
                            state->expected_decrease, scaled});
        if (state->last_merit - merit >= scaled) {  // CheckArmijoCondition :350-362
          has_converged = (merit <= state->last_merit) &&
                          std::abs(state->last_merit - merit) < S(prm.convergence_tolerance);
          state->last_merit = merit;
          accepted = true;
          break;
        }
        scale(S(prm.geometric_alpha_scaling));
        step *= S(prm.geometric_alpha_scaling);
        Rollout(p, last.xs[0], last, strategies, &cur_op);
        backtracks++;
      }
    }
    if (!accepted) {  // :146-155
      ok = false;
      break;
    }
    TotalCosts(p, cur_op, &total_costs, &state->t_extreme);
    if (log) log->push_back({merit, state->expected_decrease, step, backtracks, total_costs});
    logged_op = cur_op;
    logged_st = strategies;
    logged_costs = total_costs;
  }
  *op_io = logged_op;
  *st_io = logged_st;
  if (final_costs) *final_costs = logged_costs;
  if (iters_out) *iters_out = num_iterations;
  if (converged_out) *converged_out = has_converged ? 1 : 0;
  // SolverLog entries this call produced: iterate 0 + one per accepted iteration (:111,164)
  if (logged_out) *logged_out = 1 + (ok ? num_iterations : num_iterations - 1);
  return ok;
}

// ---------------------------------------------------------------------------
// AugmentedLagrangianSolver::Solve, src/augmented_lagrangian_solver.cpp:72-210, with
// max_runtime = infinity (the reference's per-call time budget is wall-clock).
// One ALState per instance replaces Constraint::lambdas_ / the process-global Constraint::mu_.
// Returns overall success; *logged_out = log->NumIterates() at exit.
// ---------------------------------------------------------------------------
template <class S>
bool SolveAL(const Problem<S>& p, const Vec<S>& x0, Trajectory<S>* op_io, Strategies<S>* st_io,
             Vec<S>* final_costs, int* logged_out, S* max_err_out, int* converged_out = nullptr,
             ILQState<S>* state_io = nullptr) {
  ilqg_solver_params inner = p.params;
  inner.max_solver_iters = p.params.unconstrained_solver_max_iters;  // augmented_lagrangian_solver.h:80-84
  Problem<S> pin = p;
  pin.params = inner;
  // the inner ILQSolver is a member of the AL solver (augmented_lagrangian_solver.h:80-84): its
  // last_merit_function_value_ outlives this call when the same solver object is used again
  ILQState<S> fresh_state;
  ILQState<S>& state = state_io ? *state_io : fresh_state;
  ALState<S> al(p.num_constraints, p.T, p.dt);
  Trajectory<S> warm_op = *op_io, res_op = *op_io;
  Strategies<S> warm_st = *st_io, res_st = *st_io;
  Vec<S> costs;
  int logged = 0, it = 0, conv = 0, lg = 0;
  bool success = true;
  bool inner_ok = SolveILQ(pin, x0, &res_op, &res_st, &state, &al, 0, (std::vector<IterLog<S>>*)nullptr, &costs,
                           &it, &conv, (Strategies<S>*)nullptr, &lg);
  logged += lg;
  success = success && inner_ok;
  S max_err = std::numeric_limits<S>::infinity();
  if (p.num_constraints > 0) {
    while (logged < p.params.max_solver_iters && max_err > S(p.params.constraint_error_tolerance)) {
      max_err = -std::numeric_limits<S>::infinity();
      // multiplier update at the last logged operating point (:116-140); terms are visited per
      // player in PlayerCost order: state constraints then control constraints, k outer.
      for (int i = 0; i < p.N; i++)
        for (int k = 0; k < p.T; k++) {
          const double t = 0.0 + p.dt * double(float(k));
          for (int role = ILQG_ROLE_STATE_CONSTRAINT; role <= ILQG_ROLE_CONTROL_CONSTRAINT; role++)
            for (size_t ti = 0; ti < p.terms.size(); ti++) {
              const ilqg_cost_term& c = p.terms[ti];
              if (c.player != i || c.role != role) continue;
              // FinalTimeConstraint::Evaluate (constraint/final_time_constraint.h:66-70): 0 before its threshold
              const S err = k < c.first_step ? S(0)
                            : (role == ILQG_ROLE_STATE_CONSTRAINT)
                                ? EvaluateTerm(p, (int)ti, res_op.xs[k].data(), p.n)
                                : EvaluateTerm(p, (int)ti, &res_op.us[k][p.uoff[c.arg]], p.udim(c.arg));
              max_err = std::max(max_err, err);
              S& lam = al.lambda(c.constraint_slot, t);  // Constraint::IncrementLambda, constraint.h:98-102
              lam = (c.flags & ILQG_FLAG_EQUALITY) ? lam + al.mu * err : std::max(S(0), lam + al.mu * err);
            }
        }
      al.mu *= S(p.params.geometric_mu_scaling);  // :143
      if (inner_ok) {                              // :151-154
        warm_op = res_op;
        warm_st = res_st;
      }
      res_op = warm_op;
      res_st = warm_st;
      inner_ok = SolveILQ(pin, x0, &res_op, &res_st, &state, &al, 0, (std::vector<IterLog<S>>*)nullptr, &costs,
                          &it, &conv, (Strategies<S>*)nullptr, &lg);
      if (!inner_ok) {  // :166-178
        for (auto& l : al.lambdas) l *= S(p.params.geometric_lambda_downscaling);
        al.mu *= S(p.params.geometric_mu_downscaling);
      }
      success = success && inner_ok;
      logged += lg;
    }
    if (max_err > S(p.params.constraint_error_tolerance)) success = false;  // :188-191
  }
  *op_io = res_op;
  *st_io = res_st;
  if (final_costs) *final_costs = costs;
  if (logged_out) *logged_out = logged;
  if (max_err_out) *max_err_out = max_err;
  if (converged_out) *converged_out = (inner_ok && conv) ? 1 : 0;  // SolverLog::WasConverged of the merged log
  return success;
}


// ---------------------------------------------------------------------------------------------
// Receding-horizon warm start: Problem::SyncToExistingProblem (src/problem.cpp:64-125) and
// Problem::SetUpNextRecedingHorizon (:127-186), with the integrators of
// src/multi_player_integrable_system.cpp:76-130.
// The time bookkeeping (which step the measured state falls in, how long to integrate) depends only on
// (t0, planner_runtime, plan t0), so it is computed once for a batch; everything state-dependent is per
// instance.
// ---------------------------------------------------------------------------------------------
struct RecedingHorizonTimes {
  int itn_step;          // timestep IntegrateToNextTimeStep reads the strategy from (:106-110)
  double itn_remaining;  // its integration interval (:111-112)
  float itn_frac;        // interpolation weight of x0_ref (:117)
  int integrate_begin, integrate_end;  // whole steps [begin, end) integrated afterwards (:95-101); begin == end: none
  double new_plan_t0;    // OperatingPoint::t0 of the shifted plan (:92, :102)
};
inline RecedingHorizonTimes RecedingHorizonTimesOf(double t0, double planner_runtime, double plan_t0, double dt) {
  RecedingHorizonTimes r;
  // SyncToExistingProblem :75-85
  const float kRoundingError = 0.9f;
  const double relative_t0 = t0 - plan_t0;
  size_t current_timestep = static_cast<size_t>(relative_t0 / dt);
  double remaining = (current_timestep + 1) * dt - relative_t0;
  if (remaining < kRoundingError * dt) {
    current_timestep += 1;
    remaining = dt - remaining;
  }
  // IntegrateToNextTimeStep recomputes its own step and interval (multi_player_integrable_system.cpp:104-113)
  const size_t itn_step = static_cast<size_t>((relative_t0 + 1e-4f) / dt);
  r.itn_step = int(itn_step);
  r.itn_remaining = dt * (itn_step + 1) - relative_t0;
  r.itn_frac = float(r.itn_remaining / dt);
  r.new_plan_t0 = t0 + remaining;
  r.integrate_begin = r.integrate_end = int(current_timestep) + 1;
  if (remaining <= planner_runtime) {
    const size_t num_steps = static_cast<size_t>(1e-4f + (planner_runtime - remaining) / dt);
    r.integrate_end = int(current_timestep + num_steps);
    if (r.integrate_end < r.integrate_begin) r.integrate_end = r.integrate_begin;  // Integrate(a, b): empty loop
    r.new_plan_t0 += dt * double(num_steps);
  }
  return r;
}

// Strategy::operator() for all players at step k (strategy.h:73-76): u = u_ref - P (x - x_ref) - alpha
template <class S>
inline Vec<S> ApplyStrategies(const Problem<S>& p, const Strategies<S>& st, int k, const Vec<S>& x, const Vec<S>& x_ref,
                              const Vec<S>& u_ref) {
  Vec<S> dx(p.n);
  for (int e = 0; e < p.n; e++) dx[e] = x[e] - x_ref[e];
  const Vec<S> Pdx = matvec(st.P[k], dx);
  Vec<S> u(p.m);
  for (int a = 0; a < p.m; a++) u[a] = u_ref[a] - Pdx[a] - st.alpha[k][a];
  return u;
}

// Returns first_timestep_in_new_problem; x0_next = the next solve's initial state; op / st are shifted in place.
template <class S>
int RecedingHorizonShift(const Problem<S>& p, const RecedingHorizonTimes& tm, const Vec<S>& x0, Trajectory<S>* op,
                         Strategies<S>* st, Vec<S>* x0_next) {
  const int T = p.T, n = p.n, m = p.m;
  // IntegrateToNextTimeStep (:115-129)
  const int ks = tm.itn_step;
  Vec<S> x0_ref(n);
  if (ks + 1 < T) {
    const S frac = S(tm.itn_frac);
    for (int e = 0; e < n; e++) x0_ref[e] = frac * op->xs[ks][e] + (S(1) - frac) * op->xs[ks + 1][e];
  } else {
    x0_ref = op->xs[T - 1];
  }
  Vec<S> x = Integrate(p, 0.0, tm.itn_remaining, x0, ApplyStrategies(p, *st, ks, x0, x0_ref, op->us[ks]), false);
  // Integrate(initial_timestep, final_timestep, ...) (:76-93)
  for (int kk = tm.integrate_begin; kk < tm.integrate_end; kk++)
    x = Integrate(p, 0.0, p.dt, x, ApplyStrategies(p, *st, kk, x, op->xs[kk], op->us[kk]), false);
  // nearest plan state: ConcatenatedDynamicalSystem::DistanceBetween looks at the first subsystem only
  // (concatenated_dynamical_system.cpp:109-113); the car / unicycle models measure squared position distance
  // (TwoPlayerUnicycle4D overrides it the same way: px, py only, two_player_unicycle_4d.h:141-147) — SinglePlayerDubinsCar
  // alone inherits the default, the squared norm of its whole state (single_player_dynamical_system.h:69)
  const int dist_dims = p.subs[0].kind == ILQG_DYN_DUBINS_CAR ? 3 : (p.subs[0].kind == ILQG_DYN_DELAYED_DUBINS_CAR ? 4 : 2);
  int first = 0;
  S best = std::numeric_limits<S>::infinity();
  for (int k = 0; k < T; k++) {
    S d = 0;
    for (int e = 0; e < dist_dims; e++) d += (x[e] - op->xs[k][e]) * (x[e] - op->xs[k][e]);
    if (d < best) {  // std::min_element keeps the first minimum
      best = d;
      first = k;
    }
  }
  // Stitch (concatenated_dynamical_system.h:75-84): ego block from the plan, the rest from the integration
  *x0_next = x;
  const int ego = p.xoff[1] - p.xoff[0];
  for (int e = 0; e < ego; e++) (*x0_next)[e] = op->xs[first][e];
  // shift (:136-157), then zero strategies / controls and re-propagate the tail (:170-184)
  const int keep = T - first;
  for (int kk = 0; kk < keep; kk++) {
    op->xs[kk] = op->xs[kk + first];
    op->us[kk] = op->us[kk + first];
    st->P[kk] = st->P[kk + first];
    st->alpha[kk] = st->alpha[kk + first];
  }
  for (int kk = keep; kk < T; kk++) {
    st->P[kk].setZero();
    for (auto& v : st->alpha[kk]) v = S(0);
    for (auto& v : op->us[kk]) v = S(0);
    op->xs[kk] = Integrate(p, 0.0, p.dt, op->xs[kk - 1], op->us[kk - 1], false);
  }
  return first;
}

// ---------------------------------------------------------------------------------------------
// Receding-horizon harness: the pieces RecedingHorizonSimulator (src/receding_horizon_simulator.cpp:64-137)
// strings together, on plans that carry their own length and start time (SolutionSplicer keeps up to
// kNumPreviousTimeStepsToSave = 5 rows of the old plan in front of the new one).
// ---------------------------------------------------------------------------------------------
template <class S>
struct Plan {  // (OperatingPoint, vector<Strategy>) as SolutionSplicer / Problem hold them
  Trajectory<S> op;
  Strategies<S> st;
  double t0 = 0.0;
  int len() const { return int(op.xs.size()); }
  void resize(int L) {
    op.xs.resize(L);
    op.us.resize(L);
    st.P.resize(L);
    st.alpha.resize(L);
  }
};

// SolutionSplicer::ContainsTime (solver/solution_splicer.h:66-71)
template <class S>
inline bool PlanContainsTime(const Plan<S>& pl, double t, double dt) {
  return pl.t0 <= t && pl.t0 + pl.len() * dt >= t;
}

// MultiPlayerIntegrableSystem::IntegrateToNextTimeStep (src/multi_player_integrable_system.cpp:95-130)
template <class S>
Vec<S> IntegrateToNextTimeStep(const Problem<S>& p, double t0, const Vec<S>& x0, const Plan<S>& pl) {
  const double relative_t0 = t0 - pl.t0;
  const size_t ks = static_cast<size_t>((relative_t0 + 1e-4f) / p.dt);
  const double remaining = p.dt * (ks + 1) - relative_t0;
  const float frac = float(remaining / p.dt);
  Vec<S> x0_ref(p.n);
  if (int(ks) + 1 < pl.len()) {
    for (int e = 0; e < p.n; e++) x0_ref[e] = S(frac) * pl.op.xs[ks][e] + (S(1) - S(frac)) * pl.op.xs[ks + 1][e];
  } else {
    x0_ref = pl.op.xs.back();
  }
  return Integrate(p, t0, remaining, x0, ApplyStrategies(p, pl.st, int(ks), x0, x0_ref, pl.op.us[ks]), false);
}

// Integrate(initial_timestep, final_timestep, ...) (:76-93)
template <class S>
Vec<S> IntegrateSteps(const Problem<S>& p, size_t begin, size_t end, Vec<S> x, const Plan<S>& pl) {
  for (size_t kk = begin; kk < end; kk++)
    x = Integrate(p, pl.t0 + kk * p.dt, p.dt, x, ApplyStrategies(p, pl.st, int(kk), x, pl.op.xs[kk], pl.op.us[kk]), false);
  return x;
}

// IntegrateFromPriorTimeStep (:132-155)
template <class S>
Vec<S> IntegrateFromPriorTimeStep(const Problem<S>& p, double t, const Vec<S>& x0, const Plan<S>& pl) {
  const double relative_t = t - pl.t0;
  const size_t ks = static_cast<size_t>(relative_t / p.dt);
  const double remaining = relative_t - p.dt * ks;
  return Integrate(p, pl.t0 + p.dt * ks, remaining, x0,
                   ApplyStrategies(p, pl.st, int(ks), x0, pl.op.xs[ks], pl.op.us[ks]), false);
}

// Integrate(t0, t, x0, operating_point, strategies) (:54-74).  As written there: when t0 sits exactly on
// the plan's start no partial step is taken and the whole steps begin at current_timestep + 1.
template <class S>
Vec<S> IntegrateInterval(const Problem<S>& p, double t0, double t, const Vec<S>& x0, const Plan<S>& pl) {
  const size_t current = static_cast<size_t>((t0 - pl.t0) / p.dt);
  const size_t final_step = static_cast<size_t>((t - pl.t0) / p.dt);
  Vec<S> x = x0;
  if (t0 > pl.t0) x = IntegrateToNextTimeStep(p, t0, x0, pl);
  x = IntegrateSteps(p, current + 1, final_step, x, pl);
  return IntegrateFromPriorTimeStep(p, t, x, pl);
}

// Integrate(t0, t, ...)'s own CHECKs (:57-58, :111-112, :141-142) plus the rows it reads: false = abort / out of range.
template <class S>
inline bool IntegrateIntervalValid(const Problem<S>& p, const Plan<S>& pl, double t0, double t) {
  if (t < t0 || t0 < pl.t0) return false;
  const size_t itn = static_cast<size_t>(((t0 - pl.t0) + 1e-4f) / p.dt);
  const size_t final_step = static_cast<size_t>((t - pl.t0) / p.dt);
  return int(itn) < pl.len() && int(final_step) < pl.len();
}

// What the reference CHECKs before it touches the plan (src/problem.cpp:68-70,
// multi_player_integrable_system.cpp:111-112,141-142): false = the reference would abort.
template <class S>
inline bool RecedingHorizonTimesValid(const Problem<S>& p, const Plan<S>& pl, double t0, double planner_runtime) {
  if (planner_runtime < 0.0 || planner_runtime + t0 > pl.t0 + p.dt * p.T || t0 < pl.t0) return false;
  const RecedingHorizonTimes tm = RecedingHorizonTimesOf(t0, planner_runtime, pl.t0, p.dt);
  return tm.itn_step < pl.len() && tm.integrate_end <= pl.len();
}

// Problem::SetUpNextRecedingHorizon (src/problem.cpp:127-186) on a stored plan of any length >= T:
// the plan becomes the next solve's warm start (length T, t0 = the new problem's start), *x0_next its
// initial state.  Returns first_timestep_in_new_problem.
template <class S>
int SetUpNextRecedingHorizon(const Problem<S>& p, const Vec<S>& x0, double t0, double planner_runtime, Plan<S>* pl,
                             Vec<S>* x0_next) {
  const int T = p.T;
  const RecedingHorizonTimes tm = RecedingHorizonTimesOf(t0, planner_runtime, pl->t0, p.dt);
  Vec<S> x = IntegrateToNextTimeStep(p, t0, x0, *pl);
  x = IntegrateSteps(p, size_t(tm.integrate_begin), size_t(tm.integrate_end), x, *pl);
  // nearest plan state in the first subsystem's DistanceBetween (see RecedingHorizonShift above)
  const int dist_dims = p.subs[0].kind == ILQG_DYN_DUBINS_CAR ? 3 : (p.subs[0].kind == ILQG_DYN_DELAYED_DUBINS_CAR ? 4 : 2);
  int first = 0;
  S best = std::numeric_limits<S>::infinity();
  for (int k = 0; k < pl->len(); k++) {
    S d = 0;
    for (int e = 0; e < dist_dims; e++) d += (x[e] - pl->op.xs[k][e]) * (x[e] - pl->op.xs[k][e]);
    if (d < best) {
      best = d;
      first = k;
    }
  }
  *x0_next = x;
  const int ego = p.xoff[1] - p.xoff[0];
  for (int e = 0; e < ego; e++) (*x0_next)[e] = pl->op.xs[first][e];
  const int end = std::min(first + T, pl->len());  // timestep_iterator_end (:139-140)
  for (int kk = first; kk < end; kk++) {
    pl->op.xs[kk - first] = pl->op.xs[kk];
    pl->op.us[kk - first] = pl->op.us[kk];
    pl->st.P[kk - first] = pl->st.P[kk];
    pl->st.alpha[kk - first] = pl->st.alpha[kk];
  }
  pl->resize(T);  // :160-168 (a shorter plan cannot occur: CHECK_GE at :159)
  for (int kk = end - first; kk < T; kk++) {
    pl->st.P[kk] = Mat<S>(p.m, p.n);
    pl->st.alpha[kk].assign(p.m, S(0));
    pl->op.us[kk].assign(p.m, S(0));
    pl->op.xs[kk] = Integrate(p, (kk - 1) * p.dt, p.dt, pl->op.xs[kk - 1], pl->op.us[kk - 1], false);
  }
  pl->t0 = tm.new_plan_t0;
  return first;
}

// SolutionSplicer::Splice (src/solution_splicer.cpp:60-129): keep up to 5 rows of the old plan that
// precede the new solution's start, then the new solution.
template <class S>
void SplicePlan(const Problem<S>& p, const Plan<S>& solution, Plan<S>* pl) {
  const int T = p.T;
  if (solution.t0 < pl->t0) return;  // CHECK_GE (:61): the reference aborts; a batch leaves the plan alone
  const size_t current = static_cast<size_t>(1e-4 + (solution.t0 - pl->t0) / p.dt);
  if (int(current) > pl->len()) return;  // a gap between plan and solution: rows the reference never has
  const size_t kSave = 5;
  const size_t initial = (int(current) < int(kSave)) ? 0 : current - kSave;
  for (size_t kk = initial; kk < current; kk++) {
    pl->op.xs[kk - initial] = pl->op.xs[kk];
    pl->op.us[kk - initial] = pl->op.us[kk];
    pl->st.P[kk - initial] = pl->st.P[kk];
    pl->st.alpha[kk - initial] = pl->st.alpha[kk];
  }
  const int spliced = int(current - initial) + T;
  pl->resize(spliced);
  pl->t0 += initial * p.dt;
  for (int kk = 0; kk < T; kk++) {
    const size_t at = current + kk - initial;
    pl->op.xs[at] = solution.op.xs[kk];
    pl->op.us[at] = solution.op.us[kk];
    pl->st.P[at] = solution.st.P[kk];
    pl->st.alpha[at] = solution.st.alpha[kk];
  }
}

// One solver invocation of the harness, as a test can compare it.
template <class S>
struct MpcRecord {
  double t_call;      // the simulator's clock when the solver was called
  Vec<S> x_measured;  // true state handed to SetUpNextRecedingHorizon (the initial state for record 0)
  Vec<S> x0;          // Problem::InitialState of the solve
  double plan_t0;     // OperatingPoint::t0 of the solve
  int first_step;     // first_timestep_in_new_problem (-1 for record 0)
  Plan<S> solution;   // SolverLog::FinalOperatingPoint / FinalStrategies
  int iters, ok, converged;
  int max_backtracks;  // deepest line search of the solve (-1: not recorded, AL mode)
};

// RecedingHorizonSimulator (src/receding_horizon_simulator.cpp:64-137) with the wall clock replaced by a
// fixed `solve_time` per call and no per-call time budget.  `extra_time` is the
// reference's kExtraTime = 0.25.  use_al: AugmentedLagrangianSolver, else ILQSolver; either way ONE solver
// object serves every call, so ILQSolver::last_merit_function_value_ carries over (ilq_solver.h:189).
// A solve whose times the reference would CHECK-abort on ends the run.  Constraint multipliers are indexed
// relative to the start of each window, as in the first solve (after the first ResetInitialTime the
// reference's TimeIndex CHECK_GE fails for any constrained problem, so there is nothing else to follow).
template <class S>
std::vector<MpcRecord<S>> RecedingHorizonSimulate(const Problem<S>& p, const Vec<S>& x_init, double final_time,
                                                  double planner_runtime, double extra_time, double solve_time,
                                                  bool use_al, Plan<S>* final_plan, Vec<S>* final_x,
                                                  int max_records = 1 << 30) {
  std::vector<MpcRecord<S>> recs;
  ILQState<S> state;
  ALState<S> no_al(p.num_constraints, p.T, p.dt);
  auto solve = [&](const Vec<S>& x0, Plan<S>* pl, MpcRecord<S>* r) {
    Vec<S> costs;
    int it = 0, conv = 0;
    bool ok;
    r->max_backtracks = -1;
    if (use_al) {
      S maxerr;
      ok = SolveAL(p, x0, &pl->op, &pl->st, &costs, &it, &maxerr, &conv, &state);
    } else {
      std::vector<IterLog<S>> log;
      ok = SolveILQ(p, x0, &pl->op, &pl->st, &state, &no_al, 0, &log, &costs, &it, &conv);
      r->max_backtracks = ok ? 0 : p.params.max_backtracking_steps;
      for (const auto& e : log) r->max_backtracks = std::max(r->max_backtracks, e.backtracks);
    }
    r->iters = it;
    r->ok = ok ? 1 : 0;
    r->converged = conv;
  };
  // first call: the problem as Problem::Initialize leaves it (zero operating point and strategies, t0 = 0)
  Plan<S> problem;
  problem.op = Trajectory<S>(p.T, p.n, p.m);
  problem.st = Strategies<S>(p.T, p.n, p.m);
  problem.t0 = 0.0;
  MpcRecord<S> r0;
  r0.t_call = 0.0;
  r0.x_measured = x_init;
  r0.x0 = x_init;
  r0.plan_t0 = 0.0;
  r0.first_step = -1;
  r0.solution = problem;
  solve(x_init, &r0.solution, &r0);
  recs.push_back(r0);
  Plan<S> splicer = r0.solution;  // SolutionSplicer(const SolverLog&), :56-58
  Vec<S> x = x_init;
  double t = splicer.t0;
  while (int(recs.size()) < max_records && r0.ok) {  // CHECK(success) after the first call (:77)
    t += extra_time;
    if (t >= final_time || !PlanContainsTime(splicer, t + planner_runtime + p.dt, p.dt)) break;
    if (!IntegrateIntervalValid(p, splicer, t - extra_time, t)) break;
    x = IntegrateInterval(p, t - extra_time, t, x, splicer);
    problem = splicer;  // Problem::OverwriteSolution
    if (!RecedingHorizonTimesValid(p, problem, t, planner_runtime)) break;
    MpcRecord<S> r;
    r.t_call = t;
    r.x_measured = x;
    r.first_step = SetUpNextRecedingHorizon(p, x, t, planner_runtime, &problem, &r.x0);
    r.plan_t0 = problem.t0;
    r.solution = problem;
    solve(r.x0, &r.solution, &r);
    recs.push_back(r);
    t += solve_time;
    if (t >= final_time || !PlanContainsTime(splicer, t, p.dt)) break;
    if (!IntegrateIntervalValid(p, splicer, t - solve_time, t)) break;
    x = IntegrateInterval(p, t - solve_time, t, x, splicer);
    if (r.converged) SplicePlan(p, r.solution, &splicer);
  }
  if (final_plan) *final_plan = splicer;
  if (final_x) *final_x = x;
  return recs;
}

// ---------------------------------------------------------------------------------------------
// Equilibrium checks: ComputeStrategyCosts (src/compute_strategy_costs.cpp:61-106) and
// NumericalCheckLocalNashEquilibrium (src/check_local_nash_equilibrium.cpp:60-133).
// ---------------------------------------------------------------------------------------------
// Cost of every player when the strategies are played from x0 (closed loop: against the operating point;
// open loop: u = u_ref - alpha), one-step Euler or the default RK4 integration.
template <class S>
Vec<S> ComputeStrategyCosts(const Problem<S>& p, const Vec<S>& x0, const Trajectory<S>& op, const Strategies<S>& st,
                            bool open_loop, bool euler) {
  Vec<S> x = x0, total(p.N, S(0));
  double t = 0.0;
  const int steps = open_loop ? p.T - 1 : p.T;
  const Vec<S> zero(p.n, S(0));
  for (int kk = 0; kk < steps; kk++) {
    const Vec<S> u = open_loop ? ApplyStrategies(p, st, kk, zero, zero, op.us[kk])
                               : ApplyStrategies(p, st, kk, x, op.xs[kk], op.us[kk]);
    const Vec<S> next_x = Integrate(p, t, p.dt, x, u, euler);
    // PlayerCost::EvaluateOffset (player_cost.cpp:175-190): state costs at the next state, control costs now
    for (int i = 0; i < p.N; i++)
      total[i] += EvaluatePlayer(p, i, open_loop ? next_x : x, u, open_loop ? kk + 1 : kk, kk);
    x = next_x;
    t += p.dt;
  }
  return total;
}

// Every entry of every alpha_i[k], k < T-1, moved by -/+ max_perturbation in turn (Euler integration, as the
// reference forces): a local Nash equilibrium iff no such unilateral move lowers the mover's cost.
// *margin = min over moves of (perturbed cost - nominal cost) of the mover.
template <class S>
bool CheckLocalNash(const Problem<S>& p, const Vec<S>& x0, const Trajectory<S>& op, const Strategies<S>& st,
                    S max_perturbation, bool open_loop, S* margin) {
  const Vec<S> nominal = ComputeStrategyCosts(p, x0, op, st, open_loop, true);
  S worst = std::numeric_limits<S>::infinity();
  Strategies<S> lower = st, upper = st;
  for (int i = 0; i < p.N; i++)
    for (int kk = 0; kk < p.T - 1; kk++)
      for (int jj = 0; jj < p.udim(i); jj++) {
        const int a = p.uoff[i] + jj;
        lower.alpha[kk][a] -= max_perturbation;
        upper.alpha[kk][a] += max_perturbation;
        const S cl = ComputeStrategyCosts(p, x0, op, lower, open_loop, true)[i];
        const S cu = ComputeStrategyCosts(p, x0, op, upper, open_loop, true)[i];
        worst = std::min(worst, std::min(cl, cu) - nominal[i]);
        lower.alpha[kk][a] = st.alpha[kk][a];
        upper.alpha[kk][a] = st.alpha[kk][a];
      }
  if (margin) *margin = worst;
  return !(worst < S(0));
}

// Smallest eigenvalue of a symmetric matrix by cyclic Jacobi rotations (what SelfAdjointEigenSolver::eigenvalues()
// .minCoeff() returns, src/check_local_nash_equilibrium.cpp:182-194, to rounding).
template <class S>
S MinEigenvalueSymmetric(Mat<S> a) {
  const int n = a.r;
  for (int sweep = 0; sweep < 60; sweep++) {
    S off = 0;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) off += a(i, j) * a(i, j);
    if (off < S(1e-30)) break;
    for (int pi = 0; pi < n; pi++)
      for (int q = pi + 1; q < n; q++) {
        if (a(pi, q) == S(0)) continue;
        const S theta = (a(q, q) - a(pi, pi)) / (S(2) * a(pi, q));
        const S t = (theta >= S(0) ? S(1) : S(-1)) / (std::abs(theta) + std::sqrt(theta * theta + S(1)));
        const S c = S(1) / std::sqrt(t * t + S(1)), sn = t * c;
        for (int k = 0; k < n; k++) {
          const S akp = a(k, pi), akq = a(k, q);
          a(k, pi) = c * akp - sn * akq;
          a(k, q) = sn * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          const S apk = a(pi, k), aqk = a(q, k);
          a(pi, k) = c * apk - sn * aqk;
          a(q, k) = sn * apk + c * aqk;
        }
      }
  }
  S lo = std::numeric_limits<S>::infinity();
  for (int i = 0; i < n; i++) lo = std::min(lo, a(i, i));
  return lo;
}

// CheckSufficientLocalNashEquilibrium (src/check_local_nash_equilibrium.cpp:144-201): every player's full
// PlayerCost::Quadraticize at every step has Q_i and all its R_ij with no eigenvalue below -1e-4.
// *worst = the smallest eigenvalue met.
template <class S>
bool CheckSufficientLocalNash(const Problem<S>& p, const Trajectory<S>& op, S* worst) {
  S lo = std::numeric_limits<S>::infinity();
  for (int k = 0; k < p.T; k++)
    for (int i = 0; i < p.N; i++) {
      const Quad<S> q = QuadraticizePlayer(p, i, double(k) * p.dt, op.xs[k], op.us[k], false, (const ALState<S>*)nullptr);
      lo = std::min(lo, MinEigenvalueSymmetric(q.Q));
      for (size_t pi = 0; pi < q.R.size(); pi++)
        if (q.has[pi]) lo = std::min(lo, MinEigenvalueSymmetric(q.R[pi]));
    }
  if (worst) *worst = lo;
  return !(lo < S(-1e-4f));
}

}  // namespace oracle
