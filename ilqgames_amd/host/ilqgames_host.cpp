// Host-side mirror of the reference API (include/ilqgames/host/api.hpp): flattens Problem objects
// into the C-ABI descriptor and stages trajectories to and from the device.  No numerics of the
// hot path live here; every Solve() ends in a call into libilqg_hip.so.
#include <ilqgames/host/api.hpp>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <sys/stat.h>
#include <sys/time.h>
#include <sys/types.h>

#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstring>
#include <ctime>
#include <fstream>
#include <sstream>
#include <typeinfo>

namespace ilqgames {

// ------------------------------------------------------------------------------------------
// Model index constants (src/single_player_{unicycle_4d,car_5d,car_6d}.cpp)
// ------------------------------------------------------------------------------------------
const Dimension TwoPlayerUnicycle4D::kNumXDims = 4;
const Dimension TwoPlayerUnicycle4D::kPxIdx = 0;
const Dimension TwoPlayerUnicycle4D::kPyIdx = 1;
const Dimension TwoPlayerUnicycle4D::kThetaIdx = 2;
const Dimension TwoPlayerUnicycle4D::kVIdx = 3;
const PlayerIndex TwoPlayerUnicycle4D::kNumPlayers = 2;
const Dimension TwoPlayerUnicycle4D::kNumU1Dims = 2;
const Dimension TwoPlayerUnicycle4D::kOmegaIdx = 0;
const Dimension TwoPlayerUnicycle4D::kAIdx = 1;
const Dimension TwoPlayerUnicycle4D::kNumU2Dims = 2;
const Dimension TwoPlayerUnicycle4D::kDxIdx = 0;
const Dimension TwoPlayerUnicycle4D::kDyIdx = 1;
const Dimension Air3D::kNumXDims = 3;
const Dimension Air3D::kRxIdx = 0;
const Dimension Air3D::kRyIdx = 1;
const Dimension Air3D::kRThetaIdx = 2;
const PlayerIndex Air3D::kNumPlayers = 2;
const Dimension Air3D::kNumU1Dims = 1;
const Dimension Air3D::kOmega1Idx = 0;
const Dimension Air3D::kNumU2Dims = 1;
const Dimension Air3D::kOmega2Idx = 0;
const Dimension SinglePlayerPointMass2D::kNumXDims = 4;
const Dimension SinglePlayerPointMass2D::kPxIdx = 0;
const Dimension SinglePlayerPointMass2D::kPyIdx = 1;
const Dimension SinglePlayerPointMass2D::kVxIdx = 2;
const Dimension SinglePlayerPointMass2D::kVyIdx = 3;
const Dimension SinglePlayerPointMass2D::kNumUDims = 2;
const Dimension SinglePlayerPointMass2D::kAxIdx = 0;
const Dimension SinglePlayerPointMass2D::kAyIdx = 1;
const Dimension SinglePlayerDubinsCar::kNumXDims = 3;
const Dimension SinglePlayerDubinsCar::kPxIdx = 0;
const Dimension SinglePlayerDubinsCar::kPyIdx = 1;
const Dimension SinglePlayerDubinsCar::kThetaIdx = 2;
const Dimension SinglePlayerDubinsCar::kNumUDims = 1;
const Dimension SinglePlayerDubinsCar::kOmegaIdx = 0;
const Dimension SinglePlayerUnicycle4D::kNumXDims = 4;
const Dimension SinglePlayerUnicycle4D::kPxIdx = 0;
const Dimension SinglePlayerUnicycle4D::kPyIdx = 1;
const Dimension SinglePlayerUnicycle4D::kThetaIdx = 2;
const Dimension SinglePlayerUnicycle4D::kVIdx = 3;
const Dimension SinglePlayerUnicycle4D::kNumUDims = 2;
const Dimension SinglePlayerUnicycle4D::kOmegaIdx = 0;
const Dimension SinglePlayerUnicycle4D::kAIdx = 1;

const Dimension SinglePlayerCar5D::kNumXDims = 5;
const Dimension SinglePlayerCar5D::kPxIdx = 0;
const Dimension SinglePlayerCar5D::kPyIdx = 1;
const Dimension SinglePlayerCar5D::kThetaIdx = 2;
const Dimension SinglePlayerCar5D::kPhiIdx = 3;
const Dimension SinglePlayerCar5D::kVIdx = 4;
const Dimension SinglePlayerCar5D::kNumUDims = 2;
const Dimension SinglePlayerCar5D::kOmegaIdx = 0;
const Dimension SinglePlayerCar5D::kAIdx = 1;

// src/single_player_unicycle_5d.cpp:53-64, src/single_player_car_7d.cpp:60-74, src/single_player_delayed_dubins_car.cpp:53-62
const Dimension SinglePlayerUnicycle5D::kNumXDims = 5;
const Dimension SinglePlayerUnicycle5D::kPxIdx = 0;
const Dimension SinglePlayerUnicycle5D::kPyIdx = 1;
const Dimension SinglePlayerUnicycle5D::kThetaIdx = 2;
const Dimension SinglePlayerUnicycle5D::kVIdx = 3;
const Dimension SinglePlayerUnicycle5D::kSIdx = 4;
const Dimension SinglePlayerUnicycle5D::kNumUDims = 2;
const Dimension SinglePlayerUnicycle5D::kOmegaIdx = 0;
const Dimension SinglePlayerUnicycle5D::kAIdx = 1;
const Dimension SinglePlayerCar7D::kNumXDims = 7;
const Dimension SinglePlayerCar7D::kPxIdx = 0;
const Dimension SinglePlayerCar7D::kPyIdx = 1;
const Dimension SinglePlayerCar7D::kThetaIdx = 2;
const Dimension SinglePlayerCar7D::kPhiIdx = 3;
const Dimension SinglePlayerCar7D::kVIdx = 4;
const Dimension SinglePlayerCar7D::kKappaIdx = 5;
const Dimension SinglePlayerCar7D::kSIdx = 6;
const Dimension SinglePlayerCar7D::kNumUDims = 2;
const Dimension SinglePlayerCar7D::kOmegaIdx = 0;
const Dimension SinglePlayerCar7D::kAIdx = 1;
const Dimension SinglePlayerDelayedDubinsCar::kNumXDims = 4;
const Dimension SinglePlayerDelayedDubinsCar::kPxIdx = 0;
const Dimension SinglePlayerDelayedDubinsCar::kPyIdx = 1;
const Dimension SinglePlayerDelayedDubinsCar::kThetaIdx = 2;
const Dimension SinglePlayerDelayedDubinsCar::kOmegaIdx = 3;
const Dimension SinglePlayerDelayedDubinsCar::kNumUDims = 1;
const Dimension SinglePlayerDelayedDubinsCar::kAlphaIdx = 0;
const Dimension SinglePlayerCar6D::kNumXDims = 6;
const Dimension SinglePlayerCar6D::kPxIdx = 0;
const Dimension SinglePlayerCar6D::kPyIdx = 1;
const Dimension SinglePlayerCar6D::kThetaIdx = 2;
const Dimension SinglePlayerCar6D::kPhiIdx = 3;
const Dimension SinglePlayerCar6D::kVIdx = 4;
const Dimension SinglePlayerCar6D::kAIdx = 5;
const Dimension SinglePlayerCar6D::kNumUDims = 2;
const Dimension SinglePlayerCar6D::kOmegaIdx = 0;
const Dimension SinglePlayerCar6D::kJerkIdx = 1;

// ------------------------------------------------------------------------------------------
// Geometry
// ------------------------------------------------------------------------------------------
LineSegment2::LineSegment2(const Point2& point1, const Point2& point2)
    : p1_(point1), p2_(point2), length_((point1 - point2).norm()), unit_direction_((point2 - point1) / length_) {
  CHECK_GT(length_, constants::kSmallNumber);
}

Polyline2::Polyline2(const PointList2& points) : length_(0.0f) {
  CHECK_GT(points.size(), 1);
  cumulative_lengths_.push_back(length_);
  for (size_t ii = 1; ii < points.size(); ii++) {
    segments_.emplace_back(points[ii - 1], points[ii]);
    length_ += segments_.back().Length();
    cumulative_lengths_.push_back(length_);
  }
}

void Polyline2::AddPoint(const Point2& point) {
  CHECK(!segments_.empty());
  segments_.emplace_back(segments_.back().SecondPoint(), point);
  length_ += segments_.back().Length();
  cumulative_lengths_.push_back(length_);
}

PointList2 Polyline2::Points() const {
  PointList2 pts;
  if (segments_.empty()) return pts;
  pts.push_back(segments_.front().FirstPoint());
  for (const auto& s : segments_) pts.push_back(s.SecondPoint());
  return pts;
}

// Lane centre of one arm of the roundabout scene: approach, quarter arc onto the circle, ten
// chords around it, far-away exit point (same construction and float arithmetic as
// src/roundabout_lane_center.cpp:50-106, which the example problem definitions call).
PointList2 RoundaboutLaneCenter(float entrance_angle, float exit_angle, float distance_from_roundabout) {
  const float radius = 12.0f, half_width = 2.5f;
  const Point2 arc_center((radius + half_width) * std::cos(entrance_angle),
                          (radius + half_width) * std::sin(entrance_angle));
  const float arc_start = static_cast<float>(entrance_angle - M_PI_2);  // double subtraction, as there
  const Point2 arc_first = arc_center + half_width * Point2(std::cos(arc_start), std::sin(arc_start));
  PointList2 pts;
  pts.push_back(arc_first + distance_from_roundabout * Point2(std::cos(entrance_angle), std::sin(entrance_angle)));
  pts.push_back(arc_first);
  const int arc_points = 3, circle_points = 10;
  for (int ii = 1; ii <= arc_points; ii++) {
    const float a = static_cast<float>(arc_start - M_PI_2 * static_cast<float>(ii) / arc_points);
    pts.push_back(arc_center + half_width * Point2(std::cos(a), std::sin(a)));
  }
  const Point2 on_circle(radius * std::cos(entrance_angle), radius * std::sin(entrance_angle));
  CHECK_LT((pts.back() - on_circle).norm(), constants::kSmallNumber);
  for (int ii = 1; ii <= circle_points; ii++) {
    const float a = entrance_angle + (exit_angle - entrance_angle) * static_cast<float>(ii) / circle_points;
    pts.emplace_back(radius * std::cos(a), radius * std::sin(a));
  }
  const float far = 1e4f;
  pts.emplace_back(far * std::cos(exit_angle), far * std::sin(exit_angle));
  return pts;
}

// src/draw_shapes.cpp:61-73: the angle is formed in double from a float fraction, the trigonometry narrows to the
// points' float coordinates.
Polyline2 DrawCircle(const Point2& center, float radius, size_t num_segments) {
  CHECK_GT(radius, 0.0);
  PointList2 rim;
  rim.push_back(center + Point2(radius, 0.0f));
  for (size_t step = 1; step <= num_segments; step++) {
    const double angle = 2.0 * M_PI * (static_cast<float>(step) / static_cast<float>(num_segments));
    rim.push_back(center + radius * Point2(static_cast<float>(std::cos(angle)), static_cast<float>(std::sin(angle))));
  }
  return Polyline2(rim);
}

// ------------------------------------------------------------------------------------------
// Cost / constraint descriptions
// ------------------------------------------------------------------------------------------
namespace {
void FillTerm(host::TermDescription* out, int kind, float weight, float value, int flags,
              std::initializer_list<int> idx) {
  out->term = ilqg_cost_term{};
  out->term.kind = kind;
  out->term.weight = weight;
  out->term.value = value;
  out->term.flags = flags;
  out->term.polyline = -1;
  out->term.arg = -1;
  out->term.constraint_slot = -1;
  int a = 0;
  for (int v : idx) out->term.idx[a++] = v;
}
}  // namespace

bool QuadraticCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_QUADRATIC, weight_, nominal_, 0, {dimension_ < 0 ? -1 : dimension_});
  return true;
}
bool SemiquadraticCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_SEMIQUADRATIC, weight_, threshold_, oriented_right_ ? ILQG_FLAG_ORIENTED : 0, {dimension_});
  return true;
}
bool QuadraticPolyline2Cost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_QUADRATIC_POLYLINE2, weight_, 0.0f, 0, {xidx_, yidx_});
  out->polyline = &polyline_;
  return true;
}
bool SemiquadraticPolyline2Cost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_SEMIQUADRATIC_POLYLINE2, weight_, threshold_, oriented_right_ ? ILQG_FLAG_ORIENTED : 0,
           {xidx_, yidx_});
  out->polyline = &polyline_;
  return true;
}
bool ProximityCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_PROXIMITY, weight_, threshold_, 0, {xidx1_, yidx1_, xidx2_, yidx2_});
  return true;
}
bool SignedDistanceCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_SIGNED_DISTANCE, weight_, nominal_, less_is_positive_ ? ILQG_FLAG_ORIENTED : 0,
           {xdim1_, ydim1_, xdim2_, ydim2_});
  return true;
}
bool Polyline2SignedDistanceCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_POLYLINE2_SIGNED_DISTANCE, weight_, nominal_,
           oriented_same_as_polyline_ ? ILQG_FLAG_ORIENTED : 0, {xidx_, yidx_});
  out->polyline = &polyline_;
  return true;
}
bool QuadraticDifferenceCost::Describe(host::TermDescription* out) const {
  if (dims1_.size() != 2) return false;  // other lengths have no device kernel
  FillTerm(out, ILQG_COST_QUADRATIC_DIFFERENCE, weight_, 0.0f, 0, {dims1_[0], dims1_[1], dims2_[0], dims2_[1]});
  return true;
}
bool OrientationCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_ORIENTATION, weight_, nominal_, 0, {dim_});
  return true;
}
bool QuadraticNormCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_QUADRATIC_NORM, weight_, nominal_, 0, {dim1_, dim2_});
  return true;
}
bool SemiquadraticNormCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_SEMIQUADRATIC_NORM, weight_, threshold_, oriented_right_ ? ILQG_FLAG_ORIENTED : 0,
           {dim1_, dim2_});
  return true;
}
bool RelativeDistanceCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_RELATIVE_DISTANCE, weight_, 0.0f, 0,
           {dims1_.first, dims1_.second, dims2_.first, dims2_.second});
  return true;
}
bool LocallyConvexProximityCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_LOCALLY_CONVEX_PROXIMITY, weight_, threshold_, 0, {xidx1_, yidx1_, xidx2_, yidx2_});
  return true;
}
bool WeightedConvexProximityCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_WEIGHTED_CONVEX_PROXIMITY, weight_, threshold_, 0, {xidx1_, yidx1_, xidx2_, yidx2_});
  out->term.idx_extra[0] = vidx1_;
  out->term.idx_extra[1] = vidx2_;
  return true;
}
bool CurvatureCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_CURVATURE, weight_, 0.0f, 0, {omega_idx_, v_idx_});
  return true;
}
bool NominalPathLengthCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_NOMINAL_PATH_LENGTH, weight_, nominal_speed_, 0, {dimension_});
  return true;
}
bool RouteProgressCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_ROUTE_PROGRESS, weight_, nominal_speed_, 0, {xidx_, yidx_});
  out->term.value2 = initial_route_pos_;
  out->polyline = &polyline_;
  return true;
}
bool FinalTimeCost::Describe(host::TermDescription* out) const {
  if (!cost_->Describe(out) || out->term.kind == ILQG_COST_EXTREME_VALUE) return false;
  // the first step whose time ILQSolver hands to Evaluate / Quadraticize (RelativeTime(kk) = kk * kTimeStep,
  // src/ilq_solver.cpp:236,475) passes `t >= threshold_time` (final_time_cost.h:67,75)
  int first = 0;
  while (static_cast<Time>(first) * time::kTimeStep < threshold_time_) first++;
  out->term.first_step = first;
  return true;
}
bool FinalTimeConstraint::Describe(host::TermDescription* out) const {
  if (!constraint_->Describe(out)) return false;
  int first = 0;  // as FinalTimeCost: the first step whose time passes `t >= initial_time + threshold_time` (final_time_constraint.h:67,75)
  while (static_cast<Time>(first) * time::kTimeStep < threshold_time_) first++;
  out->term.first_step = first;
  return true;
}
bool ExtremeValueCost::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_COST_EXTREME_VALUE, weight_, 0.0f, is_min_ ? ILQG_FLAG_IS_MIN : 0, {0});
  out->children = costs_;
  return true;
}
bool ProximityConstraint::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_CONSTRAINT_PROXIMITY, 1.0f, threshold_, keep_within_ ? ILQG_FLAG_ORIENTED : 0,
           {xidx1_, yidx1_, xidx2_, yidx2_});
  return true;
}
bool SingleDimensionConstraint::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_CONSTRAINT_SINGLE_DIMENSION, 1.0f, threshold_, keep_below_ ? ILQG_FLAG_ORIENTED : 0, {dim_});
  return true;
}
bool AffineScalarConstraint::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_CONSTRAINT_AFFINE_SCALAR, 1.0f, 0.0f, IsEquality() ? ILQG_FLAG_EQUALITY : 0, {0});
  out->dense.assign(a_.data(), a_.data() + a_.size());
  out->dense.push_back(b_);
  return true;
}
bool AffineVectorConstraint::Describe(host::TermDescription* out) const {
  if (A_.rows() != A_.cols()) return false;  // the reference's Quadraticize only takes a square A
  FillTerm(out, ILQG_CONSTRAINT_AFFINE_VECTOR, 1.0f, 0.0f, IsEquality() ? ILQG_FLAG_EQUALITY : 0, {0});
  for (std::ptrdiff_t j = 0; j < A_.cols(); j++)  // column-major
    for (std::ptrdiff_t i = 0; i < A_.rows(); i++) out->dense.push_back(A_(i, j));
  out->dense.insert(out->dense.end(), b_.data(), b_.data() + b_.size());
  return true;
}
bool Polyline2SignedDistanceConstraint::Describe(host::TermDescription* out) const {
  FillTerm(out, ILQG_CONSTRAINT_POLYLINE2_SIGNED_DISTANCE, 1.0f, threshold_, keep_left_ ? ILQG_FLAG_ORIENTED : 0,
           {xidx_, yidx_});
  out->polyline = &polyline_;
  return true;
}

// ------------------------------------------------------------------------------------------
// Dynamics / Problem
// ------------------------------------------------------------------------------------------
namespace {
Dimension TotalXDim(const SubsystemList& subsystems) {
  Dimension total = 0;
  for (const auto& s : subsystems) total += CHECK_NOTNULL(s.get())->XDim();
  return total;
}
}  // namespace

ConcatenatedDynamicalSystem::ConcatenatedDynamicalSystem(const SubsystemList& subsystems)
    : MultiPlayerDynamicalSystem(TotalXDim(subsystems)), subsystems_(subsystems) {
  Dimension start = 0;
  for (const auto& s : subsystems_) {
    subsystem_start_dims_.push_back(start);
    start += s->XDim();
  }
}

std::vector<Dimension> ConcatenatedDynamicalSystem::PositionDimensions() const {
  std::vector<Dimension> dims;
  for (size_t ii = 0; ii < subsystems_.size(); ii++)
    for (Dimension d : subsystems_[ii]->PositionDimensions()) dims.push_back(subsystem_start_dims_[ii] + d);
  return dims;
}

void Problem::OverwriteSolution(const OperatingPoint& operating_point, const std::vector<Strategy>& strategies) {
  CHECK(initialized_);
  *operating_point_ = operating_point;
  *strategies_ = strategies;
}

bool Problem::IsConstrained() const {
  for (const auto& pc : player_costs_)
    if (pc.IsConstrained()) return true;
  return false;
}

// ------------------------------------------------------------------------------------------
// Descriptor packing
// ------------------------------------------------------------------------------------------
namespace host {

DeviceOptions& Options() {
  static DeviceOptions options;
  return options;
}

namespace {

class Packer {
 public:
  explicit Packer(ProblemDescription* out) : out_(out) { out_->polyline_offsets.assign(1, 0); }

  bool Add(const Cost& cost, int role, int player, int arg, bool is_constraint, std::string* why) {
    TermDescription td;
    if (!cost.Describe(&td)) {
      *why = "cost '" + cost.Name() + "' of player " + std::to_string(player) + " has no device kernel";
      return false;
    }
    if (td.term.kind == ILQG_COST_EXTREME_VALUE) {
      // children sit contiguously in front of their parent
      const int begin = static_cast<int>(out_->terms.size());
      for (const auto& child : td.children)
        if (!Add(*child, ILQG_ROLE_CHILD, player, -1, false, why)) return false;
      td.term.child_begin = begin;
      td.term.child_count = static_cast<int>(out_->terms.size()) - begin;
    }
    if (td.polyline != nullptr) td.term.polyline = InternPolyline(*td.polyline);
    if (!td.dense.empty()) {  // an affine constraint's coefficient block: `polyline` is its offset in the dense table
      td.term.polyline = static_cast<int>(out_->dense_params.size());
      out_->dense_params.insert(out_->dense_params.end(), td.dense.begin(), td.dense.end());
    }
    td.term.role = role;
    td.term.player = player;
    td.term.arg = arg;
    td.term.constraint_slot = is_constraint ? out_->num_constraints++ : -1;
    out_->terms.push_back(td.term);
    return true;
  }

 private:
  // Costs hold their polyline by value; identical vertex lists share one device table.
  int InternPolyline(const Polyline2& polyline) {
    std::vector<float> flat;
    for (const auto& p : polyline.Points()) {
      flat.push_back(p.x());
      flat.push_back(p.y());
    }
    for (size_t q = 0; q + 1 < out_->polyline_offsets.size(); q++) {
      const int b = out_->polyline_offsets[q], e = out_->polyline_offsets[q + 1];
      if (static_cast<size_t>(2 * (e - b)) == flat.size() &&
          std::equal(flat.begin(), flat.end(), out_->polyline_points.begin() + 2 * b))
        return static_cast<int>(q);
    }
    out_->polyline_points.insert(out_->polyline_points.end(), flat.begin(), flat.end());
    out_->polyline_offsets.push_back(out_->polyline_offsets.back() + static_cast<int>(flat.size() / 2));
    return static_cast<int>(out_->polyline_offsets.size()) - 2;
  }

  ProblemDescription* out_;
};

}  // namespace

bool DescribeDynamics(const MultiPlayerIntegrableSystem& dynamics, ilqg_problem_desc* d, std::string* why) {
  std::string scratch;
  if (why == nullptr) why = &scratch;
  if (dynamic_cast<const TwoPlayerUnicycle4D*>(&dynamics) != nullptr) {
    d->num_players = 2;
    d->subsystems[0] = ilqg_subsystem{ILQG_DYN_UNICYCLE_4D_DISTURBED, 4, 2, 0.0f};
    d->subsystems[1] = ilqg_subsystem{ILQG_DYN_PLANAR_DISTURBANCE, 0, 2, 0.0f};
    return true;
  }
  if (const auto* air = dynamic_cast<const Air3D*>(&dynamics)) {
    d->num_players = 2;
    d->subsystems[0] = ilqg_subsystem{ILQG_DYN_AIR_3D_EVADER, 3, 1, air->evader_speed_};
    d->subsystems[1] = ilqg_subsystem{ILQG_DYN_AIR_3D_PURSUER, 0, 1, air->pursuer_speed_};
    return true;
  }
  const auto* dyn = dynamic_cast<const ConcatenatedDynamicalSystem*>(&dynamics);
  if (dyn == nullptr) {
    *why = "dynamics are neither a ConcatenatedDynamicalSystem nor TwoPlayerUnicycle4D / Air3D";
    return false;
  }
  const int N = dyn->NumPlayers();
  if (N > ILQG_MAX_PLAYERS) {
    *why = "too many players";
    return false;
  }
  d->num_players = N;
  for (int i = 0; i < N; i++) {
    d->subsystems[i] = dyn->Subsystems()[i]->Describe();
    if (d->subsystems[i].kind == 0) {
      *why = "subsystem " + std::to_string(i) + " has no device model";
      return false;
    }
  }
  return true;
}

bool DescribeProblem(const Problem& problem, const SolverParams& params, ilqg_dtype dtype,
                     ProblemDescription* out, std::string* why) {
  std::string scratch;
  if (why == nullptr) why = &scratch;
  *out = ProblemDescription();
  ilqg_problem_desc& d = out->desc;
  if (!DescribeDynamics(*problem.Dynamics(), &d, why)) return false;
  const int N = d.num_players;
  if (static_cast<size_t>(N) != problem.PlayerCosts().size()) {
    *why = "player count mismatch between dynamics and player costs";
    return false;
  }
  for (int i = 0; i < N; i++) {
    const PlayerCost& pc = problem.PlayerCosts()[i];
    d.player_costs[i].state_regularization = pc.StateRegularization();
    d.player_costs[i].control_regularization = pc.ControlRegularization();
    d.player_costs[i].structure = pc.IsTimeAdditive() ? ILQG_SUM : (pc.IsMaxOverTime() ? ILQG_MAX : ILQG_MIN);
  }
  // PlayerCost::Quadraticize visits state costs, control costs, state constraints, control
  // constraints in that order (src/player_cost.cpp:194-215); the (i, j) block list follows the
  // first touch of each j among player i's control costs then control constraints (:59-86).
  Packer packer(out);
  for (int i = 0; i < N; i++) {
    const PlayerCost& pc = problem.PlayerCosts()[i];
    for (const auto& c : pc.StateCosts())
      if (!packer.Add(*c, ILQG_ROLE_STATE_COST, i, -1, false, why)) return false;
    for (const auto& e : pc.ControlCosts())
      if (!packer.Add(*e.second, ILQG_ROLE_CONTROL_COST, i, e.first, false, why)) return false;
    for (const auto& c : pc.StateConstraints())
      if (!packer.Add(*c, ILQG_ROLE_STATE_CONSTRAINT, i, -1, true, why)) return false;
    for (const auto& e : pc.ControlConstraints())
      if (!packer.Add(*e.second, ILQG_ROLE_CONTROL_CONSTRAINT, i, e.first, true, why)) return false;
    auto touch = [&](int j) {
      for (const auto& p : out->pairs)
        if (p.i == i && p.j == j) return;
      out->pairs.push_back(ilqg_pair{i, j});
    };
    for (const auto& e : pc.ControlCosts()) touch(e.first);
    for (const auto& e : pc.ControlConstraints()) touch(e.first);
  }
  d.num_terms = static_cast<int>(out->terms.size());
  d.terms = out->terms.data();
  d.num_polylines = static_cast<int>(out->polyline_offsets.size()) - 1;
  d.polyline_offsets = out->polyline_offsets.data();
  d.polyline_points = out->polyline_points.data();
  d.num_dense_params = static_cast<int>(out->dense_params.size());
  d.dense_params = out->dense_params.empty() ? nullptr : out->dense_params.data();
  d.T = static_cast<int>(time::kNumTimeSteps);
  d.dt = time::kTimeStep;
  d.dtype = dtype;
  ilqg_solver_params& sp = d.params;
  sp.convergence_tolerance = params.convergence_tolerance;
  sp.max_solver_iters = static_cast<int>(params.max_solver_iters);
  sp.linesearch = params.linesearch ? 1 : 0;
  sp.initial_alpha_scaling = params.initial_alpha_scaling;
  sp.geometric_alpha_scaling = params.geometric_alpha_scaling;
  sp.max_backtracking_steps = static_cast<int>(params.max_backtracking_steps);
  sp.expected_decrease_fraction = params.expected_decrease_fraction;
  sp.open_loop = params.open_loop ? 1 : 0;
  sp.unconstrained_solver_max_iters = static_cast<int>(params.unconstrained_solver_max_iters);
  sp.geometric_mu_scaling = params.geometric_mu_scaling;
  sp.geometric_mu_downscaling = params.geometric_mu_downscaling;
  sp.geometric_lambda_downscaling = params.geometric_lambda_downscaling;
  sp.constraint_error_tolerance = params.constraint_error_tolerance;
  return true;
}

std::string DumpDescription(const ProblemDescription& description) {
  const ilqg_problem_desc& d = description.desc;
  std::ostringstream os;
  os.precision(9);
  os << "problem " << d.num_players << " " << d.T << " " << d.dt << "\n";
  for (int i = 0; i < d.num_players; i++) {
    os << "subsystem " << d.subsystems[i].kind << " " << d.subsystems[i].xdim << " " << d.subsystems[i].udim << " "
       << d.subsystems[i].param0 << "\n";
    os << "player_cost " << d.player_costs[i].state_regularization << " " << d.player_costs[i].control_regularization
       << " " << d.player_costs[i].structure << "\n";
  }
  for (const auto& t : description.terms)
    os << "term " << t.kind << " " << t.role << " " << t.player << " " << t.arg << " " << t.idx[0] << " " << t.idx[1]
       << " " << t.idx[2] << " " << t.idx[3] << " " << t.weight << " " << t.value << " " << t.flags << " "
       << t.polyline << " " << t.child_begin << " " << t.child_count << " " << t.constraint_slot << " " << t.first_step
       << " " << t.value2 << " " << t.idx_extra[0] << " " << t.idx_extra[1] << "\n";
  for (int q = 0; q < d.num_polylines; q++) {
    os << "polyline";
    for (int p = description.polyline_offsets[q]; p < description.polyline_offsets[q + 1]; p++)
      os << " " << description.polyline_points[2 * p] << " " << description.polyline_points[2 * p + 1];
    os << "\n";
  }
  if (!description.dense_params.empty()) {
    os << "dense";
    for (float v : description.dense_params) os << " " << v;
    os << "\n";
  }
  const ilqg_solver_params& sp = d.params;
  os << "params " << sp.convergence_tolerance << " " << sp.max_solver_iters << " " << sp.linesearch << " "
     << sp.initial_alpha_scaling << " " << sp.geometric_alpha_scaling << " " << sp.max_backtracking_steps << " "
     << sp.expected_decrease_fraction << " " << sp.open_loop << " " << sp.unconstrained_solver_max_iters << " "
     << sp.geometric_mu_scaling << " " << sp.geometric_mu_downscaling << " " << sp.geometric_lambda_downscaling
     << " " << sp.constraint_error_tolerance << "\n";
  return os.str();
}

// ------------------------------------------------------------------------------------------
// Device staging
// ------------------------------------------------------------------------------------------
namespace {

void HipCheck(hipError_t e, const char* what) {
  CHECK(e == hipSuccess) << what << ": " << hipGetErrorString(e);
}

// One HBM allocation, grown on demand.
class DeviceBuffer {
 public:
  ~DeviceBuffer() {
    if (ptr_ != nullptr) (void)hipFree(ptr_);
  }
  void* Reserve(size_t bytes) {
    if (bytes > cap_) {
      if (ptr_ != nullptr) HipCheck(hipFree(ptr_), "hipFree");
      HipCheck(hipMalloc(&ptr_, bytes), "hipMalloc");
      cap_ = bytes;
    }
    return ptr_;
  }
  void* get() const { return ptr_; }

 private:
  void* ptr_ = nullptr;
  size_t cap_ = 0;
};

// float host values <-> device scalars of `dtype`
void Upload(DeviceBuffer* buf, const std::vector<float>& host, ilqg_dtype dtype) {
  if (dtype == ILQG_F32) {
    HipCheck(hipMemcpy(buf->Reserve(host.size() * 4), host.data(), host.size() * 4, hipMemcpyHostToDevice), "H2D");
  } else {
    std::vector<double> wide(host.begin(), host.end());
    HipCheck(hipMemcpy(buf->Reserve(wide.size() * 8), wide.data(), wide.size() * 8, hipMemcpyHostToDevice), "H2D");
  }
}
std::vector<float> Download(const DeviceBuffer& buf, size_t count, ilqg_dtype dtype) {
  std::vector<float> out(count);
  if (dtype == ILQG_F32) {
    HipCheck(hipMemcpy(out.data(), buf.get(), count * 4, hipMemcpyDeviceToHost), "D2H");
  } else {
    std::vector<double> wide(count);
    HipCheck(hipMemcpy(wide.data(), buf.get(), count * 8, hipMemcpyDeviceToHost), "D2H");
    for (size_t i = 0; i < count; i++) out[i] = static_cast<float>(wide[i]);
  }
  return out;
}
std::vector<int32_t> DownloadInts(const DeviceBuffer& buf, size_t count) {
  std::vector<int32_t> out(count);
  HipCheck(hipMemcpy(out.data(), buf.get(), count * 4, hipMemcpyDeviceToHost), "D2H");
  return out;
}
size_t ElemBytes(ilqg_dtype dtype) { return dtype == ILQG_F32 ? 4 : 8; }

}  // namespace

// The bytes a device handle was built from: the POD part of the descriptor (pointers cleared) and the arrays it
// points into.  Two descriptions with equal fingerprints build identical device tables.
static std::string Fingerprint(const ProblemDescription& d) {
  // Field by field into zeroed storage: the struct has padding (after num_terms, num_polylines, T, dtype) whose bytes
  // a plain copy leaves unspecified — a spurious mismatch would rebuild the device handle and drop the solver state
  // carried between Solve() calls.
  ilqg_problem_desc pod;
  std::memset(&pod, 0, sizeof(pod));
  pod.num_players = d.desc.num_players;
  for (int i = 0; i < ILQG_MAX_PLAYERS; i++) {
    pod.subsystems[i].kind = d.desc.subsystems[i].kind;
    pod.subsystems[i].xdim = d.desc.subsystems[i].xdim;
    pod.subsystems[i].udim = d.desc.subsystems[i].udim;
    pod.subsystems[i].param0 = d.desc.subsystems[i].param0;
    pod.player_costs[i].state_regularization = d.desc.player_costs[i].state_regularization;
    pod.player_costs[i].control_regularization = d.desc.player_costs[i].control_regularization;
    pod.player_costs[i].structure = d.desc.player_costs[i].structure;
  }
  pod.num_terms = d.desc.num_terms;
  pod.num_polylines = d.desc.num_polylines;
  pod.T = d.desc.T;
  pod.dt = d.desc.dt;
  pod.dtype = d.desc.dtype;
  std::memcpy(&pod.params, &d.desc.params, sizeof(pod.params));  // 4-byte fields only: no padding
  static_assert(sizeof(ilqg_solver_params) % 4 == 0 && sizeof(ilqg_cost_term) == 19 * 4, "packed 4-byte fields");
  std::string f(reinterpret_cast<const char*>(&pod), sizeof(pod));
  f.append(reinterpret_cast<const char*>(d.terms.data()), d.terms.size() * sizeof(ilqg_cost_term));
  f.append(reinterpret_cast<const char*>(d.polyline_offsets.data()), d.polyline_offsets.size() * sizeof(int32_t));
  f.append(reinterpret_cast<const char*>(d.polyline_points.data()), d.polyline_points.size() * sizeof(float));
  f.append(reinterpret_cast<const char*>(d.dense_params.data()), d.dense_params.size() * sizeof(float));
  return f;
}

static void CheckSupportedParams(const SolverParams& params) {
  // The device solve always starts a Solve() the way the reference does with these flags at their defaults
  // (solver_params.h:81-83): multipliers, mu and the problem's stored solution handling are not switchable.
  CHECK(params.reset_problem && params.reset_lambdas && params.reset_mu)
      << "SolverParams::reset_problem / reset_lambdas / reset_mu = false are not supported by the device solve";
}

// ---------------------------------------------------------------------------------------------
// Several GPUs: sharding rule, rendezvous, RCCL communicator
// ---------------------------------------------------------------------------------------------
void InstanceRange(size_t total, int rank, int world, size_t* lo, size_t* hi) {
  CHECK_GT(world, 0);
  CHECK(rank >= 0 && rank < world);
  const size_t base = total / world, extra = total % world;
  *lo = size_t(rank) * base + std::min<size_t>(rank, extra);
  *hi = *lo + base + (size_t(rank) < extra ? 1 : 0);
}

ShardInfo ShardFromEnvironment() {
  ShardInfo s;
  auto env_int = [](const char* name, int fallback) {
    const char* v = std::getenv(name);
    return (v != nullptr && *v != 0) ? std::atoi(v) : fallback;
  };
  s.rank = env_int("RANK", 0);
  s.world = env_int("WORLD_SIZE", 1);
  s.local_rank = env_int("LOCAL_RANK", s.rank);
  if (const char* a = std::getenv("MASTER_ADDR")) s.master_addr = a;
  const int master_port = env_int("MASTER_PORT", 0);
  s.port = env_int("ILQG_RENDEZVOUS_PORT", master_port > 0 ? master_port + 1 : 29517);
  CHECK(s.world >= 1 && s.rank >= 0 && s.rank < s.world) << "RANK / WORLD_SIZE: " << s.rank << " / " << s.world;
  return s;
}

namespace {
// false = the peer went away (a stray connection must not take the process down: no SIGPIPE, no CHECK)
bool SendAll(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n > 0) {
    const ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL);
    if (w <= 0) return false;
    c += w;
    n -= size_t(w);
  }
  return true;
}
bool RecvAll(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  while (n > 0) {
    const ssize_t r = ::recv(fd, c, n, 0);  // SO_RCVTIMEO bounds it
    if (r <= 0) return false;
    c += r;
    n -= size_t(r);
  }
  return true;
}
void SetTimeout(int fd, int seconds) {
  timeval tv{};
  tv.tv_sec = seconds;
  ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  ::setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}
// What a rank says before rank 0 hands it the token: who it is and a word this job's ranks form (the job's launcher
// exports the same MASTER_PORT / WORLD_SIZE — and, if set, ILQG_RENDEZVOUS_SECRET — to every rank).  Without the secret
// the word is a function of port and world size only: it keeps ranks of DIFFERENT jobs apart (a stray connection is
// turned away), it does not authenticate — anyone who can reach the port can compute it.  Set ILQG_RENDEZVOUS_SECRET
// (any string, the same for all ranks) where the rendezvous port is reachable by others.
struct Hello {
  uint32_t magic, rank;
  uint64_t nonce;
};
constexpr uint32_t kHelloMagic = 0x494c5147u;  // "ILQG"
uint64_t JobNonce(const ShardInfo& info) {
  std::string word = std::to_string(info.port) + ":" + std::to_string(info.world) + ":";
  if (const char* secret = std::getenv("ILQG_RENDEZVOUS_SECRET")) word += secret;
  uint64_t h = 1469598103934665603ull;  // FNV-1a
  for (unsigned char ch : word) h = (h ^ ch) * 1099511628211ull;
  return h;
}
}  // namespace

// Rank 0 listens on MASTER_ADDR (not on every interface), answers only connections that identify themselves as a rank
// of this job it has not served yet, and gives up after two minutes; a stray or half-open connection costs it a
// ten-second read timeout, not the run.
void RendezvousBroadcast(const ShardInfo& info, void* token, size_t bytes) {
  if (info.world <= 1) return;
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  addr.sin_port = htons(uint16_t(info.port));
  CHECK_EQ(::inet_pton(AF_INET, info.master_addr.c_str(), &addr.sin_addr), 1)
      << "MASTER_ADDR must be a dotted IPv4 address, got " << info.master_addr;
  const uint64_t nonce = JobNonce(info);
  if (info.rank == 0) {
    const int srv = ::socket(AF_INET, SOCK_STREAM, 0);
    CHECK_GE(srv, 0) << std::strerror(errno);
    const int one = 1;
    ::setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    CHECK_EQ(::bind(srv, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)), 0)
        << "rendezvous bind to " << info.master_addr << ":" << info.port << ": " << std::strerror(errno);
    CHECK_EQ(::listen(srv, info.world), 0) << std::strerror(errno);
    SetTimeout(srv, 5);  // accept() wakes up to look at the clock
    std::vector<bool> served(size_t(info.world), false);
    int remaining = info.world - 1;
    const auto give_up = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    while (remaining > 0) {
      CHECK(std::chrono::steady_clock::now() < give_up) << "rendezvous: " << remaining << " rank(s) never connected";
      const int fd = ::accept(srv, nullptr, nullptr);
      if (fd < 0) continue;  // timeout (or a connection that was reset before it was accepted)
      SetTimeout(fd, 10);
      Hello hello{};
      const bool ok = RecvAll(fd, &hello, sizeof(hello)) && hello.magic == kHelloMagic && hello.nonce == nonce &&
                      hello.rank >= 1 && hello.rank < uint32_t(info.world) && !served[hello.rank];
      if (ok && SendAll(fd, token, bytes)) {
        served[hello.rank] = true;
        remaining--;
      }
      ::close(fd);
    }
    ::close(srv);
  } else {
    int fd = -1;
    for (int attempt = 0; attempt < 600; attempt++) {  // rank 0 may not be listening yet: retry for ~60 s
      fd = ::socket(AF_INET, SOCK_STREAM, 0);
      CHECK_GE(fd, 0) << std::strerror(errno);
      if (::connect(fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) == 0) break;
      ::close(fd);
      fd = -1;
      ::usleep(100000);
    }
    CHECK_GE(fd, 0) << "rendezvous: no answer from " << info.master_addr << ":" << info.port;
    SetTimeout(fd, 120);
    const Hello hello{kHelloMagic, uint32_t(info.rank), nonce};
    CHECK(SendAll(fd, &hello, sizeof(hello))) << "rendezvous send: " << std::strerror(errno);
    CHECK(RecvAll(fd, token, bytes)) << "rendezvous: rank 0 did not answer (wrong job, or it gave up): "
                                     << std::strerror(errno);
    ::close(fd);
  }
}

#define NCCL_CHECK(expr)                                                        \
  do {                                                                          \
    const ncclResult_t r_ = (expr);                                             \
    CHECK(r_ == ncclSuccess) << #expr << ": " << ncclGetErrorString(r_);        \
  } while (0)

ShardContext::ShardContext(const ShardInfo& info) : info_(info) {
  int count = 0;
  HipCheck(hipGetDeviceCount(&count), "hipGetDeviceCount");
  CHECK_GT(count, 0) << "no HIP device visible";
  HipCheck(hipSetDevice(info.local_rank % count), "hipSetDevice");
  hipStream_t st = nullptr;
  HipCheck(hipStreamCreate(&st), "hipStreamCreate");
  stream_ = st;
  if (info.world > 1) {
    ncclUniqueId id;
    if (info.rank == 0) NCCL_CHECK(ncclGetUniqueId(&id));
    RendezvousBroadcast(info, &id, sizeof(id));
    ncclComm_t comm = nullptr;
    NCCL_CHECK(ncclCommInitRank(&comm, info.world, id, info.rank));
    comm_ = comm;
  }
}

ShardContext::~ShardContext() {
  if (comm_ != nullptr) (void)ncclCommDestroy(static_cast<ncclComm_t>(comm_));
  if (stream_ != nullptr) (void)hipStreamDestroy(static_cast<hipStream_t>(stream_));
}

void ShardContext::AllGather(const void* send, void* recv, size_t bytes_per_rank) const {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (info_.world <= 1) {
    HipCheck(hipMemcpyAsync(recv, send, bytes_per_rank, hipMemcpyDeviceToDevice, st), "D2D");
  } else {
    NCCL_CHECK(ncclAllGather(send, recv, bytes_per_rank, ncclChar, static_cast<ncclComm_t>(comm_), st));
  }
  HipCheck(hipStreamSynchronize(st), "all-gather");
}

class DeviceSolve {
 public:
  DeviceSolve(const Problem& problem, const SolverParams& params) : dtype_(Options().dtype) {
    CheckSupportedParams(params);
    std::string why;
    CHECK(DescribeProblem(problem, params, dtype_, &description_, &why)) << why;
    fingerprint_ = Fingerprint(description_);
    const ilqg_status s = ilqg_problem_create(&description_.desc, &handle_);
    CHECK_EQ(s, ILQG_OK) << ilqg_last_error();
    n_ = problem.Dynamics()->XDim();
    m_ = problem.Dynamics()->TotalUDim();
    N_ = problem.Dynamics()->NumPlayers();
    T_ = description_.desc.T;
    for (int i = 0; i < N_; i++) udims_.push_back(problem.Dynamics()->UDim(i));
  }
  ~DeviceSolve() {
    if (handle_ != nullptr) ilqg_problem_destroy(handle_);
  }
  // Does this handle still describe `problem` under `params`?  (Costs, weights, polylines or solver parameters
  // changed on the objects after the first Solve() must not be ignored: GameSolver rebuilds the handle then.)
  bool Matches(const Problem& problem, const SolverParams& params) const {
    ProblemDescription now;
    std::string why;
    CHECK(DescribeProblem(problem, params, dtype_, &now, &why)) << why;
    return Options().dtype == dtype_ && Fingerprint(now) == fingerprint_;
  }

  // `shard` (with `total` > 0): x0s is this rank's block of a sharded batch of `total` instances; after the solve the
  // per-instance results of every rank are all-gathered on the device and the BatchResult covers the whole batch.
  // `max_runtime`: the reference's anytime budget (ilqg_solve_options::max_runtime), infinity = none.
  BatchResult Run(const std::vector<VectorXf>& x0s, const OperatingPoint& warm_op,
                  const std::vector<Strategy>& warm_strategies, bool augmented_lagrangian,
                  bool repeat_single = false, const ShardContext* shard = nullptr, size_t total = 0,
                  Time max_runtime = std::numeric_limits<Time>::infinity()) {
    size_t B = x0s.size();
    CHECK_GT(B, 0);
    const auto start = Clock::now();
    // pack the warm start once, replicate per instance
    std::vector<float> x0(B * n_), xs(B * T_ * n_), us(B * T_ * m_), P(B * T_ * m_ * n_), alpha(B * T_ * m_);
    for (size_t b = 0; b < B; b++) {
      CHECK_EQ(x0s[b].size(), n_);
      std::memcpy(&x0[b * n_], x0s[b].data(), n_ * sizeof(float));
      for (int k = 0; k < T_; k++) {
        std::memcpy(&xs[(b * T_ + k) * n_], warm_op.xs[k].data(), n_ * sizeof(float));
        int row = 0;
        for (int i = 0; i < N_; i++) {
          std::memcpy(&us[(b * T_ + k) * m_ + row], warm_op.us[k][i].data(), udims_[i] * sizeof(float));
          std::memcpy(&alpha[(b * T_ + k) * m_ + row], warm_strategies[i].alphas[k].data(),
                      udims_[i] * sizeof(float));
          // stacked (m x n) gain, column-major: player i owns rows [row, row + m_i)
          for (int c = 0; c < n_; c++)
            for (int r = 0; r < udims_[i]; r++)
              P[((b * T_ + k) * n_ + c) * m_ + row + r] = warm_strategies[i].Ps[k](r, c);
          row += udims_[i];
        }
      }
    }
    Upload(&d_x0_, x0, dtype_);
    Upload(&d_xs_, xs, dtype_);
    Upload(&d_us_, us, dtype_);
    Upload(&d_P_, P, dtype_);
    Upload(&d_alpha_, alpha, dtype_);
    d_costs_.Reserve(B * N_ * ElemBytes(dtype_));
    d_iters_.Reserve(B * 4);
    d_status_.Reserve(B * 4);
    d_conv_.Reserve(B * 4);
    uint64_t ws_bytes = 0;
    CHECK_EQ(ilqg_workspace_bytes(handle_, static_cast<int32_t>(B), &ws_bytes), ILQG_OK) << ilqg_last_error();
    d_workspace_.Reserve(ws_bytes);

    // Solve() called again on this solver object: ILQSolver::last_merit_function_value_ (ilq_solver.h:189) is
    // a member in the reference and outlives the call; the device keeps it in the workspace.  SolveBatch is
    // this library's own entry and always behaves like freshly constructed solvers.
    const bool again = repeat_single && B == 1 && solved_single_ && last_kind_ == augmented_lagrangian;
    solved_single_ = repeat_single && B == 1;
    last_kind_ = augmented_lagrangian;
    ilqg_solve_options so;
    ilqg_default_solve_options(&so);
    so.augmented_lagrangian = augmented_lagrangian ? 1 : 0;
    so.resume = again ? 1 : 0;
    if (std::isfinite(max_runtime)) so.max_runtime = max_runtime > 0.0 ? max_runtime : 1e-9;
    // GameSolver::Solve: every iterate of the solve goes into the log (the batched entries keep the final one)
    const int log_cap = (repeat_single && B == 1 && shard == nullptr) ? Options().logged_iterates : 0;
    ilqg_iterate_log il{};
    if (log_cap > 0) {
      const size_t es = ElemBytes(dtype_);
      il.xs = d_log_xs_.Reserve(size_t(log_cap) * T_ * n_ * es);
      il.us = d_log_us_.Reserve(size_t(log_cap) * T_ * m_ * es);
      il.costs = d_log_costs_.Reserve(size_t(log_cap) * N_ * es);
      il.P = d_log_P_.Reserve(size_t(log_cap) * T_ * m_ * n_ * es);
      il.alpha = d_log_alpha_.Reserve(size_t(log_cap) * T_ * m_ * es);
      il.count = static_cast<int32_t*>(d_log_count_.Reserve(4));
      il.capacity = log_cap;
      so.iterate_log = &il;
    }
    const ilqg_status s = ilqg_solve_batch_ex(
        handle_, static_cast<int32_t>(B), d_x0_.get(), d_xs_.get(), d_us_.get(), d_P_.get(), d_alpha_.get(),
        d_costs_.get(), static_cast<int32_t*>(d_iters_.get()), static_cast<int32_t*>(d_status_.get()),
        static_cast<int32_t*>(d_conv_.get()), d_workspace_.get(), &so, nullptr);
    CHECK_EQ(s, ILQG_OK) << ilqg_last_error();
    HipCheck(hipDeviceSynchronize(), "solve");

    std::vector<float> costs;
    std::vector<int32_t> iters, status, conv;
    if (shard != nullptr && total > 0) {
      // The one exchange step of a sharded solve: every rank's block of every result array, all-gathered on the
      // device (blocks differ by at most one instance: padded to the largest for the collective, trimmed here).
      const int world = shard->Info().world;
      size_t mx = 0;
      for (int r = 0; r < world; r++) {
        size_t lo, hi;
        InstanceRange(total, r, world, &lo, &hi);
        mx = std::max(mx, hi - lo);
      }
      const size_t es = ElemBytes(dtype_);
      auto gather_f = [&](DeviceBuffer& local, size_t per_instance) {
        const size_t blk = mx * per_instance * es;
        DeviceBuffer padded;  // the local block, padded to the common size (the solve's own buffer holds B instances)
        HipCheck(hipMemset(padded.Reserve(blk), 0, blk), "memset");
        HipCheck(hipMemcpy(padded.get(), local.get(), B * per_instance * es, hipMemcpyDeviceToDevice), "D2D");
        DeviceBuffer all;
        shard->AllGather(padded.get(), all.Reserve(blk * world), blk);
        std::vector<float> out;
        out.reserve(total * per_instance);
        for (int r = 0; r < world; r++) {
          size_t lo, hi;
          InstanceRange(total, r, world, &lo, &hi);
          DeviceBuffer view;  // Download() reads from the start of a buffer: copy the rank's valid rows to one
          HipCheck(hipMemcpy(view.Reserve((hi - lo) * per_instance * es), static_cast<char*>(all.get()) + size_t(r) * blk,
                             (hi - lo) * per_instance * es, hipMemcpyDeviceToDevice), "D2D");
          const std::vector<float> part = Download(view, (hi - lo) * per_instance, dtype_);
          out.insert(out.end(), part.begin(), part.end());
        }
        return out;
      };
      auto gather_i = [&](DeviceBuffer& local) {
        const size_t blk = mx * 4;
        DeviceBuffer padded, all;
        HipCheck(hipMemset(padded.Reserve(blk), 0, blk), "memset");
        HipCheck(hipMemcpy(padded.get(), local.get(), B * 4, hipMemcpyDeviceToDevice), "D2D");
        shard->AllGather(padded.get(), all.Reserve(blk * world), blk);
        std::vector<int32_t> host(mx * world), out;
        HipCheck(hipMemcpy(host.data(), all.get(), blk * world, hipMemcpyDeviceToHost), "D2H");
        for (int r = 0; r < world; r++) {
          size_t lo, hi;
          InstanceRange(total, r, world, &lo, &hi);
          out.insert(out.end(), host.begin() + r * mx, host.begin() + r * mx + (hi - lo));
        }
        return out;
      };
      xs = gather_f(d_xs_, size_t(T_) * n_);
      us = gather_f(d_us_, size_t(T_) * m_);
      P = gather_f(d_P_, size_t(T_) * m_ * n_);
      alpha = gather_f(d_alpha_, size_t(T_) * m_);
      costs = gather_f(d_costs_, N_);
      iters = gather_i(d_iters_);
      status = gather_i(d_status_);
      conv = gather_i(d_conv_);
      B = total;
    } else {
      xs = Download(d_xs_, xs.size(), dtype_);
      us = Download(d_us_, us.size(), dtype_);
      P = Download(d_P_, P.size(), dtype_);
      alpha = Download(d_alpha_, alpha.size(), dtype_);
      costs = Download(d_costs_, B * N_, dtype_);
      iters = DownloadInts(d_iters_, B);
      status = DownloadInts(d_status_, B);
      conv = DownloadInts(d_conv_, B);
    }
    const Time elapsed = std::chrono::duration<Time>(Clock::now() - start).count();

    // one logged iterate out of flat (instance-major) host arrays
    auto unpack = [&](const std::vector<float>& fxs, const std::vector<float>& fus, const std::vector<float>& fP,
                      const std::vector<float>& fal, size_t b, OperatingPoint* op, std::vector<Strategy>* strategies) {
      for (int k = 0; k < T_; k++) {
        std::memcpy(op->xs[k].data(), &fxs[(b * T_ + k) * n_], n_ * sizeof(float));
        int row = 0;
        for (int i = 0; i < N_; i++) {
          std::memcpy(op->us[k][i].data(), &fus[(b * T_ + k) * m_ + row], udims_[i] * sizeof(float));
          std::memcpy((*strategies)[i].alphas[k].data(), &fal[(b * T_ + k) * m_ + row], udims_[i] * sizeof(float));
          for (int c = 0; c < n_; c++)
            for (int r = 0; r < udims_[i]; r++)
              (*strategies)[i].Ps[k](r, c) = fP[((b * T_ + k) * n_ + c) * m_ + row + r];
          row += udims_[i];
        }
      }
    };
    BatchResult result;
    for (size_t b = 0; b < B; b++) {
      auto log = std::make_shared<SolverLog>();
      int logged = 0, produced = 0;
      if (log_cap > 0) {
        // the iterates the device copied (slot q = the q-th AddSolverIterate of the reference's log); cumulative
        // run times are the call's wall time spread evenly — the device loop has no per-iteration host clock
        produced = DownloadInts(d_log_count_, 1)[0];
        logged = std::min(produced, log_cap);
        const std::vector<float> lxs = Download(d_log_xs_, size_t(logged) * T_ * n_, dtype_);
        const std::vector<float> lus = Download(d_log_us_, size_t(logged) * T_ * m_, dtype_);
        const std::vector<float> lP = Download(d_log_P_, size_t(logged) * T_ * m_ * n_, dtype_);
        const std::vector<float> lal = Download(d_log_alpha_, size_t(logged) * T_ * m_, dtype_);
        const std::vector<float> lc = Download(d_log_costs_, size_t(logged) * N_, dtype_);
        for (int q = 0; q < logged; q++) {
          OperatingPoint op(warm_op);
          std::vector<Strategy> strategies(warm_strategies);
          unpack(lxs, lus, lP, lal, size_t(q), &op, &strategies);
          const bool last = q + 1 == produced;
          log->AddSolverIterate(op, strategies, std::vector<float>(lc.begin() + q * N_, lc.begin() + (q + 1) * N_),
                                produced > 1 ? elapsed * q / (produced - 1) : elapsed, last && conv[b] != 0);
        }
      }
      if (logged == 0 || logged < produced) {  // no log asked for, or it overflowed: the final iterate from the solve's outputs
        OperatingPoint op(warm_op);
        std::vector<Strategy> strategies(warm_strategies);
        unpack(xs, us, P, alpha, b, &op, &strategies);
        log->AddSolverIterate(op, strategies, std::vector<float>(costs.begin() + b * N_, costs.begin() + (b + 1) * N_),
                              elapsed, conv[b] != 0);
      }
      log->SetDeviceIterations(iters[b]);
      result.logs.push_back(log);
      result.success.push_back(status[b] != 0);
    }
    return result;
  }

 private:
  const ilqg_dtype dtype_;
  ProblemDescription description_;
  std::string fingerprint_;
  ilqg_problem* handle_ = nullptr;
  int n_ = 0, m_ = 0, N_ = 0, T_ = 0;
  std::vector<int> udims_;
  bool solved_single_ = false, last_kind_ = false;
  DeviceBuffer d_x0_, d_xs_, d_us_, d_P_, d_alpha_, d_costs_, d_iters_, d_status_, d_conv_, d_workspace_;
  DeviceBuffer d_log_xs_, d_log_us_, d_log_costs_, d_log_P_, d_log_alpha_, d_log_count_;
};

}  // namespace host

// ------------------------------------------------------------------------------------------
// SolverLog on disk (src/solver_log.cpp:113-171, :208-240)
// ------------------------------------------------------------------------------------------
namespace {

std::string LogRoot() {
  const char* env = std::getenv("ILQGAMES_LOG_DIR");
  return env != nullptr ? std::string(env) : std::string("./logs");
}

// One row the way Eigen streams `x.transpose()`: every coefficient at the stream's default precision,
// right-aligned to the widest one, separated by a single space.
void WriteRow(std::ostream& os, const float* v, std::ptrdiff_t count) {
  std::vector<std::string> cells;
  size_t width = 0;
  for (std::ptrdiff_t i = 0; i < count; i++) {
    std::ostringstream cell;
    cell << v[i];
    cells.push_back(cell.str());
    width = std::max(width, cells.back().size());
  }
  for (size_t i = 0; i < cells.size(); i++) {
    if (i > 0) os << " ";
    os << std::string(width - cells[i].size(), ' ') << cells[i];
  }
  os << std::endl;
}

}  // namespace

bool MakeDirectory(const std::string& directory_name) {
  if (mkdir(directory_name.c_str(), 0777) == -1) {
    LOG(ERROR) << "Could not create directory " << directory_name << ". Error msg: " << std::strerror(errno);
    return false;
  }
  return true;
}

std::string SolverLog::DefaultExperimentName() {
  char buf[64];
  const std::time_t now = std::time(nullptr);
  std::strftime(buf, sizeof(buf), "%Y-%m-%d-%H-%M-%S", std::localtime(&now));
  return std::string("experiment_") + buf;
}

bool SolverLog::Save(bool only_last_trajectory, const std::string& experiment_name) const {
  const std::string dir_name = LogRoot() + "/" + experiment_name;
  if (!MakeDirectory(dir_name)) return false;
  size_t start = 0;
  if (only_last_trajectory) start = operating_points_.size() - 1;
  for (size_t ii = start; ii < operating_points_.size(); ii++) {
    const OperatingPoint& op = operating_points_[ii];
    const std::string sub = dir_name + "/" + std::to_string(ii);
    if (!MakeDirectory(sub)) return false;
    std::ofstream(sub + "/t0.txt") << op.t0 << std::endl;
    {
      std::ofstream file(sub + "/xs.txt");
      for (const auto& x : op.xs) WriteRow(file, x.data(), x.size());
    }
    {
      std::ofstream file(sub + "/costs.txt");
      for (float c : total_player_costs_[ii]) file << c << std::endl;
    }
    std::ofstream(sub + "/cumulative_runtimes.txt") << cumulative_runtimes_[ii] << std::endl;
    const size_t players = NumPlayers();
    for (size_t jj = 0; jj < players; jj++) {
      std::ofstream file(sub + "/u" + std::to_string(jj) + ".txt");
      for (const auto& us : op.us) {
        CHECK_EQ(players, us.size());
        WriteRow(file, us[jj].data(), us[jj].size());
      }
    }
  }
  return true;
}

bool SaveLogs(const std::vector<SolverLog>& logs, bool only_last_trajectory, const std::string& experiment_name) {
  if (!MakeDirectory(LogRoot() + "/" + experiment_name)) return false;
  for (size_t ii = 0; ii < logs.size(); ii++)
    if (!logs[ii].Save(only_last_trajectory, experiment_name + "/" + std::to_string(ii))) return false;
  return true;
}

bool SaveLogs(const std::vector<std::shared_ptr<const SolverLog>>& logs, bool only_last_trajectory,
              const std::string& experiment_name) {
  if (!MakeDirectory(LogRoot() + "/" + experiment_name)) return false;
  for (size_t ii = 0; ii < logs.size(); ii++)
    if (!logs[ii]->Save(only_last_trajectory, experiment_name + "/" + std::to_string(ii))) return false;
  return true;
}

// ------------------------------------------------------------------------------------------
// Receding horizon
// ------------------------------------------------------------------------------------------
namespace host {

// A stored plan (operating point + strategies of any length) flattened the way the C ABI takes it.
struct FlatPlan {
  int rows = 0, n = 0, m = 0;
  std::vector<float> xs, us, P, alpha;
};

FlatPlan FlattenPlan(const MultiPlayerIntegrableSystem& dyn, const OperatingPoint& op,
                     const std::vector<Strategy>& strategies, int capacity) {
  FlatPlan f;
  f.rows = static_cast<int>(op.xs.size());
  f.n = dyn.XDim();
  f.m = dyn.TotalUDim();
  const int N = dyn.NumPlayers(), n = f.n, m = f.m;
  CHECK_LE(f.rows, capacity);
  CHECK_EQ(static_cast<int>(strategies.size()), N);
  f.xs.assign(size_t(capacity) * n, 0.0f);
  f.us.assign(size_t(capacity) * m, 0.0f);
  f.P.assign(size_t(capacity) * m * n, 0.0f);
  f.alpha.assign(size_t(capacity) * m, 0.0f);
  for (int k = 0; k < f.rows; k++) {
    std::memcpy(&f.xs[size_t(k) * n], op.xs[k].data(), n * sizeof(float));
    int row = 0;
    for (int i = 0; i < N; i++) {
      const int mi = dyn.UDim(i);
      std::memcpy(&f.us[size_t(k) * m + row], op.us[k][i].data(), mi * sizeof(float));
      std::memcpy(&f.alpha[size_t(k) * m + row], strategies[i].alphas[k].data(), mi * sizeof(float));
      for (int c = 0; c < n; c++)
        for (int r = 0; r < mi; r++) f.P[(size_t(k) * n + c) * m + row + r] = strategies[i].Ps[k](r, c);
      row += mi;
    }
  }
  return f;
}

// Device buffers of one stored plan, plus its length / start time as the harness kernels read them.
struct DevicePlan {
  DeviceBuffer xs, us, P, alpha, len, t0;
  int capacity = 0;
  void Upload(const FlatPlan& f, int cap, Time plan_t0, ilqg_dtype dtype) {
    capacity = cap;
    host::Upload(&xs, f.xs, dtype);
    host::Upload(&us, f.us, dtype);
    host::Upload(&P, f.P, dtype);
    host::Upload(&alpha, f.alpha, dtype);
    const int32_t rows = f.rows;
    const double start = plan_t0;
    len.Reserve(sizeof(int32_t));
    t0.Reserve(sizeof(double));
    HipCheck(hipMemcpy(len.get(), &rows, sizeof(rows), hipMemcpyHostToDevice), "plan length");
    HipCheck(hipMemcpy(t0.get(), &start, sizeof(start), hipMemcpyHostToDevice), "plan start time");
  }
};

// ilqg_problem handle that only needs the dynamics (the plan kernels never look at cost terms).
struct DynamicsHandle {
  ilqg_problem* handle = nullptr;
  explicit DynamicsHandle(const MultiPlayerIntegrableSystem& dynamics) {
    ilqg_problem_desc d{};
    std::string why;
    CHECK(DescribeDynamics(dynamics, &d, &why)) << why;
    // the handle's tables want every player to own a control Hessian; the plan kernels never evaluate it
    ilqg_cost_term own_control[ILQG_MAX_PLAYERS] = {};
    for (int i = 0; i < d.num_players; i++) {
      ilqg_cost_term& c = own_control[i];
      c.kind = ILQG_COST_QUADRATIC;
      c.role = ILQG_ROLE_CONTROL_COST;
      c.player = c.arg = i;
      c.idx[0] = -1;
      c.weight = 1.0f;
      c.polyline = c.constraint_slot = -1;
    }
    d.terms = own_control;
    d.num_terms = d.num_players;
    const int32_t no_polylines[1] = {0};
    d.polyline_offsets = no_polylines;
    d.T = static_cast<int>(time::kNumTimeSteps);
    d.dt = time::kTimeStep;
    d.dtype = Options().dtype;
    CHECK_EQ(ilqg_problem_create(&d, &handle), ILQG_OK) << ilqg_last_error();
  }
  ~DynamicsHandle() {
    if (handle != nullptr) ilqg_problem_destroy(handle);
  }
};

}  // namespace host

// include/ilqgames/dynamics/multi_player_integrable_system.h:75-79, src/multi_player_integrable_system.cpp:54-74
VectorXf MultiPlayerIntegrableSystem::Integrate(Time t0, Time t, const VectorXf& x0,
                                                const OperatingPoint& operating_point,
                                                const std::vector<Strategy>& strategies) const {
  using namespace host;
  CHECK_GE(t, t0);
  CHECK_GE(t0, operating_point.t0);
  const ilqg_dtype dtype = Options().dtype;
  const int n = XDim();
  const int cap = std::max<int>(static_cast<int>(operating_point.xs.size()), static_cast<int>(time::kNumTimeSteps));
  DynamicsHandle dyn(*this);
  DevicePlan plan;
  plan.Upload(FlattenPlan(*this, operating_point, strategies, cap), cap, operating_point.t0, dtype);
  DeviceBuffer dx, dactive;
  Upload(&dx, std::vector<float>(x0.data(), x0.data() + n), dtype);
  const int32_t one = 1;
  dactive.Reserve(sizeof(int32_t));
  HipCheck(hipMemcpy(dactive.get(), &one, sizeof(one), hipMemcpyHostToDevice), "active flag");
  // must_contain = t0: always inside the plan here (CHECKed above); rows the integration would read past the
  // plan clear the flag, which is where the reference CHECK-aborts (:111-112, :141-142)
  const ilqg_status s = ilqg_plan_integrate_batch(
      dyn.handle, 1, cap, plan.xs.get(), plan.us.get(), plan.P.get(), plan.alpha.get(),
      static_cast<const int32_t*>(plan.len.get()), static_cast<const double*>(plan.t0.get()), t0, t, t0, dx.get(),
      static_cast<int32_t*>(dactive.get()), nullptr);
  CHECK_EQ(s, ILQG_OK) << ilqg_last_error();
  HipCheck(hipDeviceSynchronize(), "plan integrate");
  CHECK_EQ(DownloadInts(dactive, 1)[0], 1) << "Integrate(" << t0 << ", " << t << "): times outside the stored plan";
  const std::vector<float> xn = Download(dx, n, dtype);
  VectorXf out = VectorXf::Zero(n);
  std::memcpy(out.data(), xn.data(), n * sizeof(float));
  return out;
}

void Problem::SetUpNextRecedingHorizon(const VectorXf& x0, Time t0, Time planner_runtime) {
  using namespace host;
  CHECK(initialized_);
  // RouteProgressCost subtracts the window's start time from the time it is handed (src/route_progress_cost.cpp:58,
  // reset by src/problem.cpp:120); the device tabulates its per-step nominals once, for initial time 0
  // (ilqg_problem_create), so a re-anchored problem would diverge from the reference: refused here as the C ABI's
  // receding-horizon entry points refuse it.
  for (const PlayerCost& pc : player_costs_)
    for (const auto& cost : pc.StateCosts()) {
      TermDescription td;
      const bool known = cost->Describe(&td);
      CHECK(!(known && td.term.kind == ILQG_COST_ROUTE_PROGRESS))
          << "SetUpNextRecedingHorizon: RouteProgressCost is only supported in a first solve (initial time 0)";
    }
  const ilqg_dtype dtype = Options().dtype;
  const int n = dynamics_->XDim(), m = dynamics_->TotalUDim(), N = dynamics_->NumPlayers();
  const int T = static_cast<int>(time::kNumTimeSteps);
  const int rows = static_cast<int>(operating_point_->xs.size());
  CHECK_GE(rows, T);  // src/problem.cpp:159
  // the stored plan may be what a SolutionSplicer handed over: up to five rows longer than the horizon
  const int cap = rows;
  DynamicsHandle dyn(*dynamics_);
  DevicePlan plan;
  plan.Upload(FlattenPlan(*dynamics_, *operating_point_, *strategies_, cap), cap, operating_point_->t0, dtype);
  DeviceBuffer dx0, dxs, dus, dP, dal, dxn, dfirst, dt0, dactive;
  Upload(&dx0, std::vector<float>(x0.data(), x0.data() + n), dtype);
  dxs.Reserve(size_t(T) * n * ElemBytes(dtype));
  dus.Reserve(size_t(T) * m * ElemBytes(dtype));
  dP.Reserve(size_t(T) * m * n * ElemBytes(dtype));
  dal.Reserve(size_t(T) * m * ElemBytes(dtype));
  dxn.Reserve(n * ElemBytes(dtype));
  dfirst.Reserve(sizeof(int32_t));
  dt0.Reserve(sizeof(double));
  const int32_t one = 1;
  dactive.Reserve(sizeof(int32_t));
  HipCheck(hipMemcpy(dactive.get(), &one, sizeof(one), hipMemcpyHostToDevice), "active flag");
  const ilqg_status s = ilqg_receding_horizon_sync_batch(
      dyn.handle, 1, cap, plan.xs.get(), plan.us.get(), plan.P.get(), plan.alpha.get(),
      static_cast<const int32_t*>(plan.len.get()), static_cast<const double*>(plan.t0.get()), dx0.get(), t0,
      planner_runtime, dxs.get(), dus.get(), dP.get(), dal.get(), dxn.get(), static_cast<double*>(dt0.get()),
      static_cast<int32_t*>(dfirst.get()), static_cast<int32_t*>(dactive.get()), nullptr);
  CHECK_EQ(s, ILQG_OK) << ilqg_last_error();
  HipCheck(hipDeviceSynchronize(), "receding horizon");
  // times outside the stored plan abort in the reference (CHECKs at src/problem.cpp:68-70); same here
  CHECK_EQ(DownloadInts(dactive, 1)[0], 1) << "SetUpNextRecedingHorizon: t0 / planner_runtime outside the stored plan";
  const std::vector<float> xs = Download(dxs, size_t(T) * n, dtype), us = Download(dus, size_t(T) * m, dtype),
                           P = Download(dP, size_t(T) * m * n, dtype), alpha = Download(dal, size_t(T) * m, dtype),
                           xn = Download(dxn, n, dtype);
  double new_t0 = 0.0;
  HipCheck(hipMemcpy(&new_t0, dt0.get(), sizeof(new_t0), hipMemcpyDeviceToHost), "new plan start time");
  operating_point_->xs.resize(T);
  operating_point_->us.resize(T);
  for (int i = 0; i < N; i++) {
    (*strategies_)[i].Ps.resize(T);
    (*strategies_)[i].alphas.resize(T);
  }
  for (int k = 0; k < T; k++) {
    std::memcpy(operating_point_->xs[k].data(), &xs[size_t(k) * n], n * sizeof(float));
    int row = 0;
    for (int i = 0; i < N; i++) {
      const int mi = dynamics_->UDim(i);
      std::memcpy(operating_point_->us[k][i].data(), &us[size_t(k) * m + row], mi * sizeof(float));
      std::memcpy((*strategies_)[i].alphas[k].data(), &alpha[size_t(k) * m + row], mi * sizeof(float));
      for (int c = 0; c < n; c++)
        for (int r = 0; r < mi; r++) (*strategies_)[i].Ps[k](r, c) = P[(size_t(k) * n + c) * m + row + r];
      row += mi;
    }
  }
  x0_ = VectorXf::Zero(n);
  std::memcpy(x0_.data(), xn.data(), n * sizeof(float));
  operating_point_->t0 = new_t0;
}

namespace host {

std::vector<std::vector<std::shared_ptr<const SolverLog>>> RecedingHorizonSimulatorBatch(
    Time final_time, Time planner_runtime, GameSolver* solver, const std::vector<VectorXf>& x0s,
    Time simulated_solve_time) {
  CHECK_NOTNULL(solver);
  CHECK_GE(simulated_solve_time, 0.0);
  CHECK_LE(simulated_solve_time, planner_runtime);  // receding_horizon_simulator.cpp:119
  const Problem& problem = solver->GetProblem();
  const MultiPlayerIntegrableSystem& dyn = *problem.Dynamics();
  const ilqg_dtype dtype = Options().dtype;
  const bool al = solver->IsAugmentedLagrangian();
  ProblemDescription description;
  std::string why;
  CHECK(DescribeProblem(problem, solver->Params(), dtype, &description, &why)) << why;
  ilqg_problem* handle = nullptr;
  CHECK_EQ(ilqg_problem_create(&description.desc, &handle), ILQG_OK) << ilqg_last_error();
  const int B = static_cast<int>(x0s.size()), n = dyn.XDim(), m = dyn.TotalUDim(), N = dyn.NumPlayers();
  const int T = description.desc.T, cap = T + 5;
  CHECK_GT(B, 0);
  const size_t eb = ElemBytes(dtype);
  // true states, solve buffers (zero warm start: Problem::Initialize), stored plans
  std::vector<float> x(size_t(B) * n);
  for (int b = 0; b < B; b++) {
    CHECK_EQ(x0s[b].size(), n);
    std::memcpy(&x[size_t(b) * n], x0s[b].data(), n * sizeof(float));
  }
  DeviceBuffer dx, dx0, dxs, dus, dP, dal, dcosts, diters, dstatus, dconv, dws, dfirst, dt0, dactive;
  DeviceBuffer pxs, pus, pP, pal, plen, pt0;
  Upload(&dx, x, dtype);
  Upload(&dx0, x, dtype);
  auto zeros = [&](DeviceBuffer* buf, size_t bytes) { HipCheck(hipMemset(buf->Reserve(bytes), 0, bytes), "memset"); };
  zeros(&dxs, size_t(B) * T * n * eb);
  zeros(&dus, size_t(B) * T * m * eb);
  zeros(&dP, size_t(B) * T * m * n * eb);
  zeros(&dal, size_t(B) * T * m * eb);
  zeros(&pxs, size_t(B) * cap * n * eb);
  zeros(&pus, size_t(B) * cap * m * eb);
  zeros(&pP, size_t(B) * cap * m * n * eb);
  zeros(&pal, size_t(B) * cap * m * eb);
  zeros(&plen, size_t(B) * sizeof(int32_t));
  zeros(&pt0, size_t(B) * sizeof(double));
  zeros(&dt0, size_t(B) * sizeof(double));
  zeros(&dfirst, size_t(B) * sizeof(int32_t));
  dcosts.Reserve(size_t(B) * N * eb);
  diters.Reserve(size_t(B) * 4);
  dstatus.Reserve(size_t(B) * 4);
  dconv.Reserve(size_t(B) * 4);
  uint64_t ws_bytes = 0;
  CHECK_EQ(ilqg_workspace_bytes(handle, B, &ws_bytes), ILQG_OK) << ilqg_last_error();
  dws.Reserve(ws_bytes);
  std::vector<int32_t> active(B, 1);
  dactive.Reserve(size_t(B) * 4);
  auto push_active = [&] { HipCheck(hipMemcpy(dactive.get(), active.data(), size_t(B) * 4, hipMemcpyHostToDevice), "active"); };
  auto pull_active = [&] { active = DownloadInts(dactive, B); };
  push_active();
  auto* act = static_cast<int32_t*>(dactive.get());
  auto* it = static_cast<int32_t*>(diters.get());
  auto* st = static_cast<int32_t*>(dstatus.get());
  auto* cv = static_cast<int32_t*>(dconv.get());
  std::vector<std::vector<std::shared_ptr<const SolverLog>>> logs(B);
  std::vector<double> solve_t0(B, 0.0);
  // one log entry per active instance from what the solve left in the buffers
  auto record = [&](const std::vector<int32_t>& who, Time elapsed) {
    const std::vector<float> xs = Download(dxs, size_t(B) * T * n, dtype), us = Download(dus, size_t(B) * T * m, dtype),
                             P = Download(dP, size_t(B) * T * m * n, dtype), alpha = Download(dal, size_t(B) * T * m, dtype),
                             costs = Download(dcosts, size_t(B) * N, dtype);
    const std::vector<int32_t> iters = DownloadInts(diters, B), conv = DownloadInts(dconv, B);
    for (int b = 0; b < B; b++) {
      if (!who[b]) continue;
      OperatingPoint op(T, N, solve_t0[b]);
      std::vector<Strategy> strategies;
      for (int i = 0; i < N; i++) strategies.emplace_back(T, n, dyn.UDim(i));
      for (int k = 0; k < T; k++) {
        op.xs[k] = VectorXf::Zero(n);
        std::memcpy(op.xs[k].data(), &xs[(size_t(b) * T + k) * n], n * sizeof(float));
        int row = 0;
        for (int i = 0; i < N; i++) {
          const int mi = dyn.UDim(i);
          op.us[k][i] = VectorXf::Zero(mi);
          std::memcpy(op.us[k][i].data(), &us[(size_t(b) * T + k) * m + row], mi * sizeof(float));
          std::memcpy(strategies[i].alphas[k].data(), &alpha[(size_t(b) * T + k) * m + row], mi * sizeof(float));
          for (int c = 0; c < n; c++)
            for (int r = 0; r < mi; r++) strategies[i].Ps[k](r, c) = P[((size_t(b) * T + k) * n + c) * m + row + r];
          row += mi;
        }
      }
      auto log = std::make_shared<SolverLog>();
      log->AddSolverIterate(op, strategies,
                            std::vector<float>(costs.begin() + size_t(b) * N, costs.begin() + size_t(b + 1) * N), elapsed,
                            conv[b] != 0);
      log->SetDeviceIterations(iters[b]);
      logs[b].push_back(log);
    }
  };
  auto check = [&](ilqg_status s) { CHECK_EQ(s, ILQG_OK) << ilqg_last_error(); };
  // ---- first call (receding_horizon_simulator.cpp:72-82) ----
  check(al ? ilqg_al_solve_batch(handle, B, dx0.get(), dxs.get(), dus.get(), dP.get(), dal.get(), dcosts.get(), it, st, cv,
                                 dws.get(), nullptr)
           : ilqg_ilq_solve_batch(handle, B, dx0.get(), dxs.get(), dus.get(), dP.get(), dal.get(), dcosts.get(), it, st,
                                  cv, dws.get(), 0, nullptr));
  HipCheck(hipDeviceSynchronize(), "first solve");
  record(active, 0.0);
  check(ilqg_solution_splice_batch(handle, B, cap, pxs.get(), pus.get(), pP.get(), pal.get(),
                                   static_cast<int32_t*>(plen.get()), static_cast<double*>(pt0.get()), dxs.get(),
                                   dus.get(), dP.get(), dal.get(), static_cast<double*>(dt0.get()), nullptr, nullptr,
                                   nullptr));
  {
    const std::vector<int32_t> ok = DownloadInts(dstatus, B);  // CHECK(success) there: a failed instance stops here
    for (int b = 0; b < B; b++) active[b] = ok[b] ? 1 : 0;
    push_active();
  }
  const Time kExtraTime = 0.25;
  Time t = 0.0;
  auto any_active = [&] {
    for (int v : active)
      if (v) return true;
    return false;
  };
  while (any_active()) {
    t += kExtraTime;
    if (t >= final_time) break;
    check(ilqg_plan_integrate_batch(handle, B, cap, pxs.get(), pus.get(), pP.get(), pal.get(),
                                    static_cast<int32_t*>(plen.get()), static_cast<double*>(pt0.get()), t - kExtraTime,
                                    t, t + planner_runtime + time::kTimeStep, dx.get(), act, nullptr));
    check(ilqg_receding_horizon_sync_batch(handle, B, cap, pxs.get(), pus.get(), pP.get(), pal.get(),
                                           static_cast<int32_t*>(plen.get()), static_cast<double*>(pt0.get()), dx.get(),
                                           t, planner_runtime, dxs.get(), dus.get(), dP.get(), dal.get(), dx0.get(),
                                           static_cast<double*>(dt0.get()), static_cast<int32_t*>(dfirst.get()), act,
                                           nullptr));
    HipCheck(hipDeviceSynchronize(), "receding horizon sync");
    pull_active();
    if (!any_active()) break;
    HipCheck(hipMemcpy(solve_t0.data(), dt0.get(), size_t(B) * sizeof(double), hipMemcpyDeviceToHost), "window starts");
    check(ilqg_solve_again_batch(handle, B, dx0.get(), dxs.get(), dus.get(), dP.get(), dal.get(), dcosts.get(), it, st,
                                 cv, dws.get(), al ? 1 : 0, act, nullptr));
    HipCheck(hipDeviceSynchronize(), "solve");
    record(active, simulated_solve_time);
    t += simulated_solve_time;
    if (t >= final_time) break;
    check(ilqg_plan_integrate_batch(handle, B, cap, pxs.get(), pus.get(), pP.get(), pal.get(),
                                    static_cast<int32_t*>(plen.get()), static_cast<double*>(pt0.get()),
                                    t - simulated_solve_time, t, t, dx.get(), act, nullptr));
    check(ilqg_solution_splice_batch(handle, B, cap, pxs.get(), pus.get(), pP.get(), pal.get(),
                                     static_cast<int32_t*>(plen.get()), static_cast<double*>(pt0.get()), dxs.get(),
                                     dus.get(), dP.get(), dal.get(), static_cast<double*>(dt0.get()), cv, act, nullptr));
    HipCheck(hipDeviceSynchronize(), "splice");
    pull_active();
  }
  ilqg_problem_destroy(handle);
  return logs;
}

}  // namespace host

// ------------------------------------------------------------------------------------------
// Equilibrium checks (src/compute_strategy_costs.cpp:108-114, src/check_local_nash_equilibrium.cpp:135-142)
// ------------------------------------------------------------------------------------------
namespace {

// Runs one of the two device checks on the problem's stored solution.
void RunEquilibriumCheck(const Problem& problem, bool open_loop, const float* max_perturbation,
                         std::vector<float>* costs, bool* is_nash) {
  using namespace host;
  const ilqg_dtype dtype = Options().dtype;
  ProblemDescription description;
  std::string why;
  CHECK(DescribeProblem(problem, SolverParams(), dtype, &description, &why)) << why;
  ilqg_problem* handle = nullptr;
  CHECK_EQ(ilqg_problem_create(&description.desc, &handle), ILQG_OK) << ilqg_last_error();
  const MultiPlayerIntegrableSystem& dyn = *problem.Dynamics();
  const int n = dyn.XDim(), N = dyn.NumPlayers(), T = description.desc.T;
  CHECK_EQ(static_cast<int>(problem.CurrentOperatingPoint().xs.size()), T);
  DevicePlan plan;
  plan.Upload(FlattenPlan(dyn, problem.CurrentOperatingPoint(), problem.CurrentStrategies(), T), T,
              problem.CurrentOperatingPoint().t0, dtype);
  DeviceBuffer dx0, dout, dflag;
  const VectorXf& x0 = problem.InitialState();
  Upload(&dx0, std::vector<float>(x0.data(), x0.data() + n), dtype);
  ilqg_status s;
  if (costs != nullptr) {
    dout.Reserve(N * ElemBytes(dtype));
    s = ilqg_strategy_costs_batch(handle, 1, dx0.get(), plan.xs.get(), plan.us.get(), plan.P.get(), plan.alpha.get(),
                                  open_loop ? 1 : 0, /*euler=*/0, dout.get(), nullptr);
  } else {
    dflag.Reserve(sizeof(int32_t));
    s = ilqg_check_local_nash_batch(handle, 1, dx0.get(), plan.xs.get(), plan.us.get(), plan.P.get(),
                                    plan.alpha.get(), *max_perturbation, open_loop ? 1 : 0,
                                    static_cast<int32_t*>(dflag.get()), nullptr, nullptr);
  }
  CHECK_EQ(s, ILQG_OK) << ilqg_last_error();
  HipCheck(hipDeviceSynchronize(), "equilibrium check");
  if (costs != nullptr) *costs = Download(dout, N, dtype);
  if (is_nash != nullptr) *is_nash = DownloadInts(dflag, 1)[0] != 0;
  ilqg_problem_destroy(handle);
}

}  // namespace

std::vector<float> ComputeStrategyCosts(const Problem& problem, bool open_loop) {
  std::vector<float> costs;
  RunEquilibriumCheck(problem, open_loop, nullptr, &costs, nullptr);
  return costs;
}

bool CheckSufficientLocalNashEquilibrium(const Problem& problem) {
  using namespace host;
  const ilqg_dtype dtype = Options().dtype;
  ProblemDescription description;
  std::string why;
  CHECK(DescribeProblem(problem, SolverParams(), dtype, &description, &why)) << why;
  ilqg_problem* handle = nullptr;
  CHECK_EQ(ilqg_problem_create(&description.desc, &handle), ILQG_OK) << ilqg_last_error();
  const int T = description.desc.T;
  CHECK_EQ(static_cast<int>(problem.CurrentOperatingPoint().xs.size()), T);
  DevicePlan plan;
  plan.Upload(FlattenPlan(*problem.Dynamics(), problem.CurrentOperatingPoint(), problem.CurrentStrategies(), T), T,
              problem.CurrentOperatingPoint().t0, dtype);
  DeviceBuffer dflag;
  dflag.Reserve(sizeof(int32_t));
  const ilqg_status s = ilqg_check_sufficient_nash_batch(handle, 1, plan.xs.get(), plan.us.get(),
                                                         static_cast<int32_t*>(dflag.get()), nullptr);
  CHECK_EQ(s, ILQG_OK) << ilqg_last_error();
  HipCheck(hipDeviceSynchronize(), "sufficient Nash check");
  const bool ok = DownloadInts(dflag, 1)[0] != 0;
  ilqg_problem_destroy(handle);
  return ok;
}

bool NumericalCheckLocalNashEquilibrium(const Problem& problem, float max_perturbation, bool open_loop) {
  bool is_nash = false;
  RunEquilibriumCheck(problem, open_loop, &max_perturbation, nullptr, &is_nash);
  return is_nash;
}

// ------------------------------------------------------------------------------------------
// SolutionSplicer (src/solution_splicer.cpp:56-129) — bookkeeping on the host-side containers; the batched
// form that keeps plans on the device is ilqg_solution_splice_batch.
// ------------------------------------------------------------------------------------------
SolutionSplicer::SolutionSplicer(const SolverLog& log)
    : strategies_(log.FinalStrategies()), operating_point_(log.FinalOperatingPoint()) {}

void SolutionSplicer::Splice(const SolverLog& log) {
  const OperatingPoint& fresh = log.FinalOperatingPoint();
  const std::vector<Strategy>& fresh_strategies = log.FinalStrategies();
  const size_t T = time::kNumTimeSteps;
  CHECK_GE(fresh.t0, operating_point_.t0);
  CHECK_GE(operating_point_.xs.size(), T);
  CHECK_EQ(fresh.xs.size(), T);
  // where the new solution begins inside the stored one, and how much of the past to keep in front of it
  const size_t begins_at = static_cast<size_t>(1e-4 + (fresh.t0 - operating_point_.t0) / time::kTimeStep);
  const size_t kept = std::min<size_t>(begins_at, 5);
  const size_t dropped = begins_at - kept;
  const size_t total = kept + T;
  OperatingPoint spliced(total, static_cast<PlayerIndex>(strategies_.size()),
                         operating_point_.t0 + dropped * time::kTimeStep);
  std::vector<Strategy> spliced_strategies;
  for (const Strategy& st : strategies_)
    spliced_strategies.emplace_back(total, static_cast<Dimension>(st.Ps[0].cols()),
                                    static_cast<Dimension>(st.Ps[0].rows()));
  for (size_t kk = 0; kk < total; kk++) {
    const bool old_part = kk < kept;
    const OperatingPoint& src = old_part ? operating_point_ : fresh;
    const std::vector<Strategy>& src_strategies = old_part ? strategies_ : fresh_strategies;
    const size_t at = old_part ? dropped + kk : kk - kept;
    spliced.xs[kk] = src.xs[at];
    spliced.us[kk] = src.us[at];
    for (size_t ii = 0; ii < strategies_.size(); ii++) {
      spliced_strategies[ii].Ps[kk] = src_strategies[ii].Ps[at];
      spliced_strategies[ii].alphas[kk] = src_strategies[ii].alphas[at];
    }
  }
  operating_point_.swap(spliced);
  strategies_.swap(spliced_strategies);
}

// ------------------------------------------------------------------------------------------
// RecedingHorizonSimulator (src/receding_horizon_simulator.cpp:64-137)
// ------------------------------------------------------------------------------------------
std::vector<std::shared_ptr<const SolverLog>> RecedingHorizonSimulator(Time final_time, Time planner_runtime,
                                                                       GameSolver* solver) {
  CHECK_NOTNULL(solver);
  using Clock = std::chrono::system_clock;
  const Time fixed_solve_time = host::Options().simulated_solve_time;
  std::vector<std::shared_ptr<const SolverLog>> logs;
  bool success = false;
  logs.push_back(solver->Solve(&success));
  CHECK(success);
  Problem& problem = solver->GetProblem();
  SolutionSplicer splicer(*logs.front());
  VectorXf x(problem.InitialState());
  Time t = splicer.CurrentOperatingPoint().t0;
  const Time kExtraTime = 0.25;
  while (true) {
    t += kExtraTime;
    if (t >= final_time || !splicer.ContainsTime(t + planner_runtime + time::kTimeStep)) break;
    x = problem.Dynamics()->Integrate(t - kExtraTime, t, x, splicer.CurrentOperatingPoint(),
                                      splicer.CurrentStrategies());
    problem.OverwriteSolution(splicer.CurrentOperatingPoint(), splicer.CurrentStrategies());
    problem.SetUpNextRecedingHorizon(x, t, planner_runtime);
    const auto call_time = Clock::now();
    // a fixed simulated solve time asks for reproducible runs: no wall-clock budget inside the solve then either
    logs.push_back(solver->Solve(&success, fixed_solve_time >= 0.0 ? std::numeric_limits<Time>::infinity()
                                                                    : planner_runtime));
    Time elapsed = std::chrono::duration<Time>(Clock::now() - call_time).count();
    if (fixed_solve_time >= 0.0) elapsed = fixed_solve_time;  // deterministic runs (host::DeviceOptions)
    // receding_horizon_simulator.cpp:119 CHECKs elapsed <= planner_runtime, which the reference's anytime exit
    // guarantees; a device solve has no such exit and is accounted as returning at its deadline (see ILQSolver::Solve)
    if (elapsed > planner_runtime) {
      LOG(WARNING) << "solve took " << elapsed << " s of a " << planner_runtime << " s budget: accounted as the budget";
      elapsed = planner_runtime;
    }
    t += elapsed;
    if (t >= final_time || !splicer.ContainsTime(t)) break;
    x = problem.Dynamics()->Integrate(t - elapsed, t, x, splicer.CurrentOperatingPoint(),
                                      splicer.CurrentStrategies());
    if (logs.back()->WasConverged()) splicer.Splice(*logs.back());
  }
  return logs;
}

// ------------------------------------------------------------------------------------------
// MinimallyInvasiveRecedingHorizonSimulator (src/minimally_invasive_receding_horizon_simulator.cpp:68-218):
// two planners over one plant.  Both are re-solved at every replanning instant from the spliced plan; the
// safety planner's answer is adopted when its P1 value is above the threshold (:201-204) or when it alone
// converged (:205-206), otherwise the original planner's (if that one converged).
// ------------------------------------------------------------------------------------------
std::vector<ActiveProblem> MinimallyInvasiveRecedingHorizonSimulator(
    Time final_time, Time planner_runtime, GameSolver* original, GameSolver* safety,
    std::vector<std::shared_ptr<const SolverLog>>* original_logs,
    std::vector<std::shared_ptr<const SolverLog>>* safety_logs) {
  CHECK_NOTNULL(original);
  CHECK_NOTNULL(safety);
  CHECK_NOTNULL(original_logs);
  CHECK_NOTNULL(safety_logs);
  using Clock = std::chrono::system_clock;
  const Time fixed_solve_time = host::Options().simulated_solve_time;
  Problem& plan_a = original->GetProblem();
  Problem& plan_b = safety->GetProblem();
  CHECK(plan_a.InitialState().isApprox(plan_b.InitialState(), constants::kSmallNumber));
  CHECK_NEAR(plan_a.InitialTime(), plan_b.InitialTime(), constants::kSmallNumber);
  const MultiPlayerIntegrableSystem& plant = *plan_a.Dynamics();
  {
    const MultiPlayerIntegrableSystem& other = *plan_b.Dynamics();
    CHECK(typeid(plant) == typeid(other));
  }
  original_logs->clear();
  safety_logs->clear();

  // one solver call, charged its wall-clock time (or the fixed simulated time of host::DeviceOptions)
  const auto timed_solve = [&](GameSolver* solver, std::vector<std::shared_ptr<const SolverLog>>* logs, bool first) {
    const auto call_time = Clock::now();
    bool success = false;
    logs->push_back((first || fixed_solve_time >= 0.0) ? solver->Solve(&success) : solver->Solve(&success, planner_runtime));
    if (first) CHECK(success);
    const Time elapsed = std::chrono::duration<Time>(Clock::now() - call_time).count();
    return fixed_solve_time >= 0.0 ? fixed_solve_time : elapsed;
  };
  timed_solve(original, original_logs, true);
  timed_solve(safety, safety_logs, true);

  SolutionSplicer splicer(*original_logs->front());  // the original controller is the one that starts (:121-123)
  std::vector<ActiveProblem> active = {ActiveProblem::ORIGINAL};
  VectorXf x(plan_a.InitialState());
  Time t = plan_a.InitialTime();
  const Time kExtraTime = 0.25;
  constexpr float kSafetyThreshold = -1.0;
  while (true) {
    t += kExtraTime;
    if (t >= final_time || !splicer.ContainsTime(t + planner_runtime + time::kTimeStep)) break;
    x = plant.Integrate(t - kExtraTime, t, x, splicer.CurrentOperatingPoint(), splicer.CurrentStrategies());

    // both planners restart from the plan in force and from the active planner's initial state (:141-160)
    const VectorXf anchor = (active.back() == ActiveProblem::ORIGINAL ? plan_a : plan_b).InitialState();
    for (Problem* plan : {&plan_a, &plan_b}) {
      plan->OverwriteSolution(splicer.CurrentOperatingPoint(), splicer.CurrentStrategies());
      plan->ResetInitialState(anchor);
    }
    for (Problem* plan : {&plan_a, &plan_b}) plan->SetUpNextRecedingHorizon(x, t, planner_runtime);

    // (minimally_invasive_receding_horizon_simulator.cpp CHECKs both against planner_runtime: see ILQSolver::Solve)
    const Time spent_a = std::min(timed_solve(original, original_logs, false), planner_runtime);
    const Time spent_b = std::min(timed_solve(safety, safety_logs, false), planner_runtime);
    const Time elapsed = std::max(spent_a, spent_b);
    t += elapsed;
    if (t >= final_time || !splicer.ContainsTime(t)) break;
    x = plant.Integrate(t - elapsed, t, x, splicer.CurrentOperatingPoint(), splicer.CurrentStrategies());

    const SolverLog& log_a = *original_logs->back();
    const SolverLog& log_b = *safety_logs->back();
    const float safety_value = log_b.TotalCosts().front();
    if (safety_value > kSafetyThreshold || (log_b.WasConverged() && !log_a.WasConverged())) {
      active.push_back(ActiveProblem::SAFETY);
      splicer.Splice(log_b);
    } else {
      active.push_back(ActiveProblem::ORIGINAL);
      if (log_a.WasConverged()) splicer.Splice(log_a);
    }
  }
  return active;
}

// ------------------------------------------------------------------------------------------
// Solvers
// ------------------------------------------------------------------------------------------
GameSolver::GameSolver(const std::shared_ptr<Problem>& problem, const SolverParams& params,
                       bool augmented_lagrangian)
    : problem_(problem), params_(params), augmented_lagrangian_(augmented_lagrangian) {
  CHECK_NOTNULL(problem_.get());
  CHECK_NOTNULL(problem_->Dynamics().get());
}

GameSolver::~GameSolver() {}

// The device tables are built on the first Solve() and rebuilt when the Problem's costs / weights / polylines or the
// SolverParams no longer flatten to what they were built from (a rebuilt handle starts like a new solver object).
void GameSolver::RefreshDevice() {
  if (device_ && !device_->Matches(*problem_, params_)) {
    LOG(INFO) << "GameSolver: the problem or its SolverParams changed since the device tables were built; rebuilding "
                 "them (the solver state carried between Solve() calls starts afresh).";
    device_.reset();
  }
  if (!device_) device_.reset(new host::DeviceSolve(*problem_, params_));
}

host::BatchResult GameSolver::SolveBatch(const std::vector<VectorXf>& x0s) {
  RefreshDevice();
  return device_->Run(x0s, problem_->CurrentOperatingPoint(), problem_->CurrentStrategies(), augmented_lagrangian_);
}

host::BatchResult GameSolver::SolveBatchSharded(const std::vector<VectorXf>& x0s, const host::ShardContext& shard) {
  RefreshDevice();
  size_t lo, hi;
  host::InstanceRange(x0s.size(), shard.Info().rank, shard.Info().world, &lo, &hi);
  CHECK_GT(hi, lo) << "a sharded batch needs at least one instance per rank";
  const std::vector<VectorXf> mine(x0s.begin() + lo, x0s.begin() + hi);
  return device_->Run(mine, problem_->CurrentOperatingPoint(), problem_->CurrentStrategies(), augmented_lagrangian_,
                      /*repeat_single=*/false, &shard, x0s.size());
}

host::BatchResult GameSolver::SolveOne(Time max_runtime) {
  RefreshDevice();
  return device_->Run({problem_->InitialState()}, problem_->CurrentOperatingPoint(), problem_->CurrentStrategies(),
                      augmented_lagrangian_, /*repeat_single=*/true, nullptr, 0, max_runtime);
}

// `max_runtime` is the reference's wall-clock anytime exit (src/ilq_solver.cpp:123-124; per inner solve of a
// constrained problem max_runtime / max_solver_iters, src/augmented_lagrangian_solver.cpp:85-88): the device loop
// leaves at the first iteration boundary at which the host's clock says the next iteration no longer fits
// (ilqg_solve_options::max_runtime).  Infinity (the default) runs to the iteration bounds.  The receding-horizon
// simulators below still treat a solve that overran its budget as one that returned AT its deadline instead of
// CHECK-failing on the clock: a round of the device loop cannot be interrupted.
std::shared_ptr<SolverLog> ILQSolver::Solve(bool* success, Time max_runtime) {
  host::BatchResult r = SolveOne(max_runtime);
  if (success != nullptr) *success = r.success[0];
  return r.logs[0];
}

std::shared_ptr<SolverLog> AugmentedLagrangianSolver::Solve(bool* success, Time max_runtime) {
  host::BatchResult r = SolveOne(max_runtime);
  if (success != nullptr) *success = r.success[0];
  return r.logs[0];
}

std::vector<Strategy> LQSolver::SolveOnDevice(
    bool open_loop, const std::vector<LinearDynamicsApproximation>& linearization,
    const std::vector<std::vector<QuadraticCostApproximation>>& quadraticization, const VectorXf& x0,
    std::vector<VectorXf>* delta_xs, std::vector<std::vector<VectorXf>>* costates) {
  using namespace host;
  const int T = static_cast<int>(num_time_steps_), n = dynamics_->XDim(), N = dynamics_->NumPlayers();
  const int m = dynamics_->TotalUDim();
  CHECK_EQ(linearization.size(), num_time_steps_);
  CHECK_EQ(quadraticization.size(), num_time_steps_);
  const ilqg_dtype dtype = Options().dtype;

  ilqg_dims d{};
  d.n = n;
  d.num_players = N;
  d.T = T;
  d.batch = 1;
  d.dtype = dtype;
  d.adaptive_regularization = 1;
  std::vector<int> uoff(N + 1, 0);
  for (int i = 0; i < N; i++) {
    d.udim[i] = dynamics_->UDim(i);
    uoff[i + 1] = uoff[i] + d.udim[i];
  }
  // (i, j) blocks = keys of quad[0][i].control, in stored order; offsets into the R / r slabs
  std::vector<ilqg_pair> pairs;
  std::vector<int> Roff(1, 0), roff(1, 0);
  for (int i = 0; i < N; i++) {
    CHECK_EQ(quadraticization[0].size(), static_cast<size_t>(N));
    for (const auto& e : quadraticization[0][i].control) {
      pairs.push_back(ilqg_pair{i, static_cast<int>(e.first)});
      Roff.push_back(Roff.back() + d.udim[e.first] * d.udim[e.first]);
      roff.push_back(roff.back() + d.udim[e.first]);
    }
  }
  const int Rsz = Roff.back(), rsz = roff.back();
  std::vector<float> A(T * n * n), Bm(T * n * m), Q(T * N * n * n), l(T * N * n), R(T * Rsz), r(T * rsz);
  for (int k = 0; k < T; k++) {
    const auto& lin = linearization[k];
    CHECK_EQ(lin.A.rows(), n);
    std::memcpy(&A[k * n * n], lin.A.data(), n * n * sizeof(float));
    for (int i = 0; i < N; i++)
      std::memcpy(&Bm[k * n * m + n * uoff[i]], lin.Bs[i].data(), n * d.udim[i] * sizeof(float));
    int q = 0;
    for (int i = 0; i < N; i++) {
      const auto& quad = quadraticization[k][i];
      std::memcpy(&Q[(k * N + i) * n * n], quad.state.hess.data(), n * n * sizeof(float));
      std::memcpy(&l[(k * N + i) * n], quad.state.grad.data(), n * sizeof(float));
      for (const auto& e : quad.control) {
        CHECK_EQ(pairs[q].j, static_cast<int>(e.first)) << "control blocks must not change along the horizon";
        const int mj = d.udim[e.first];
        std::memcpy(&R[k * Rsz + Roff[q]], e.second.hess.data(), mj * mj * sizeof(float));
        std::memcpy(&r[k * rsz + roff[q]], e.second.grad.data(), mj * sizeof(float));
        q++;
      }
    }
  }
  // the reference CHECKs that delta_xs and costates come together (lq_feedback_solver.cpp:77-78,
  // lq_open_loop_solver.cpp:83-84)
  if (delta_xs) CHECK_NOTNULL(costates);
  if (costates) CHECK_NOTNULL(delta_xs);
  DeviceBuffer dA, dB, dQ, dl, dR, dr, dx0, dP, dalpha, ddx, dco;
  Upload(&dA, A, dtype);
  Upload(&dB, Bm, dtype);
  Upload(&dQ, Q, dtype);
  Upload(&dl, l, dtype);
  Upload(&dR, R, dtype);
  Upload(&dr, r, dtype);
  Upload(&dx0, std::vector<float>(x0.data(), x0.data() + n), dtype);
  dP.Reserve(T * m * n * ElemBytes(dtype));
  dalpha.Reserve(T * m * ElemBytes(dtype));
  ddx.Reserve(T * n * ElemBytes(dtype));
  if (costates != nullptr) dco.Reserve(T * N * n * ElemBytes(dtype));

  auto fn = open_loop ? ilqg_lq_openloop_batch : ilqg_lq_feedback_batch;
  const ilqg_status s = fn(&d, dA.get(), dB.get(), dQ.get(), dl.get(), dR.get(), dr.get(), pairs.data(),
                           static_cast<int32_t>(pairs.size()), dx0.get(), dP.get(), dalpha.get(), ddx.get(),
                           costates != nullptr ? dco.get() : nullptr, nullptr);
  // dimension mismatches / a missing R_ii abort in the reference (glog CHECK); same here
  CHECK_EQ(s, ILQG_OK) << ilqg_last_error();
  HipCheck(hipDeviceSynchronize(), "lq solve");

  const std::vector<float> P = Download(dP, T * m * n, dtype), alpha = Download(dalpha, T * m, dtype),
                           dxs = Download(ddx, T * n, dtype);
  std::vector<Strategy> strategies;
  for (int i = 0; i < N; i++) {
    strategies.emplace_back(num_time_steps_, n, d.udim[i]);
    for (int k = 0; k < T; k++) {
      for (int c = 0; c < n; c++)
        for (int rr = 0; rr < d.udim[i]; rr++) strategies[i].Ps[k](rr, c) = P[(k * n + c) * m + uoff[i] + rr];
      for (int rr = 0; rr < d.udim[i]; rr++) strategies[i].alphas[k](rr) = alpha[k * m + uoff[i] + rr];
    }
  }
  if (delta_xs != nullptr) {
    delta_xs->assign(T, VectorXf::Zero(n));
    for (int k = 0; k < T; k++) std::memcpy((*delta_xs)[k].data(), &dxs[k * n], n * sizeof(float));
  }
  if (costates != nullptr) {  // [T][N][n] (lq_feedback_solver.cpp:83-92)
    const std::vector<float> co = Download(dco, T * N * n, dtype);
    costates->assign(T, std::vector<VectorXf>(N, VectorXf::Zero(n)));
    for (int k = 0; k < T; k++)
      for (int i = 0; i < N; i++) std::memcpy((*costates)[k][i].data(), &co[(k * N + i) * n], n * sizeof(float));
  }
  return strategies;
}

}  // namespace ilqgames
