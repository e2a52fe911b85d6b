// Host-side C++ mirror of the reference's problem-definition and solver API, bound to the
// MI355X kernels through the C ABI of include/ilqg.h.
//
// Same class names, constructor arguments and call sequence as the reference (cited per class,
// paths relative to the reference repo) so a `Problem` subclass written for the reference compiles
// unchanged — but the objects here are *descriptors*: they record what the user asked for and are
// flattened into an `ilqg_problem_desc` (host::DescribeProblem).  All arithmetic of the hot path
// (rollout, linearisation, quadraticisation, LQ sweep, line search, AL loop) runs on the GPU inside
// libilqg_hip.so; there is no CPU evaluation path behind `Cost`, `Constraint` or
// `SinglePlayerDynamicalSystem`, so user-defined subclasses with their own Evaluate() have no device
// kernel and make Solve() report ILQG_ERR_UNSUPPORTED through a CHECK, the reference's error style.
#ifndef ILQGAMES_HOST_API_HPP_
#define ILQGAMES_HOST_API_HPP_

#include <ilqg.h>
#include <ilqgames/host/linalg.hpp>
#include <ilqgames/host/logging.hpp>

#include <math.h>
#include <algorithm>
#include <chrono>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

namespace ilqgames {

// ---------------------------------------------------------------------------------------------
// Types and constants (include/ilqgames/utils/types.h:60-147)
// ---------------------------------------------------------------------------------------------
using VectorXf = host::Vector<float>;
using MatrixXf = host::Matrix<float>;
using Point2 = host::Point2f;
using PointList2 = std::vector<Point2>;
using PlayerIndex = unsigned short;
using Dimension = int;
using Time = double;
using Clock = std::chrono::system_clock;

template <typename T>
using PtrVector = std::vector<std::shared_ptr<T>>;
template <typename T>
using PlayerMap = std::unordered_map<PlayerIndex, T>;
// Insertion-ordered (player, object) list.  The reference keeps control costs in an
// unordered_multimap (types.h:84-86); an ordered list makes the per-player summation order —
// and therefore the floating-point result — a defined property of the problem definition.
template <typename T>
class PlayerPtrMultiMap : public std::vector<std::pair<PlayerIndex, std::shared_ptr<T>>> {
 public:
  void emplace(PlayerIndex idx, const std::shared_ptr<T>& ptr) { this->emplace_back(idx, ptr); }
};

namespace constants {
static constexpr float kSmallNumber = 1e-4;
static constexpr float kInfinity = std::numeric_limits<float>::infinity();
static constexpr float kInvalidValue = std::numeric_limits<float>::quiet_NaN();
static constexpr float kDefaultLambda = 0.0;
static constexpr float kDefaultMu = 10.0;
}  // namespace constants

namespace time {
static constexpr Time kTimeStep = 0.1;
static constexpr Time kTimeHorizon = 10.0;
static constexpr size_t kNumTimeSteps =
    static_cast<size_t>((kTimeHorizon + constants::kSmallNumber) / kTimeStep);
}  // namespace time

template <typename T, typename... Args>
std::unique_ptr<T> make_unique(Args&&... args) {
  return std::unique_ptr<T>(new T(std::forward<Args>(args)...));
}
template <typename T>
inline constexpr T sgn(T x) {
  return static_cast<T>((T(0) < x) - (x < T(0)));
}

class Cost;
class Constraint;
class Polyline2;

namespace host {
// What one Cost / Constraint contributes to the device descriptor.
struct TermDescription {
  ilqg_cost_term term{};                            // kind, idx, weight, value, flags
  const Polyline2* polyline = nullptr;              // for *_POLYLINE2 kinds
  std::vector<std::shared_ptr<const Cost>> children;  // EXTREME_VALUE
  std::vector<float> dense;                         // affine constraints: their coefficient block (ilqg.h)
};
}  // namespace host

// ---------------------------------------------------------------------------------------------
// Geometry (include/ilqgames/geometry/line_segment2.h:52-92, polyline2.h:54-96)
// ---------------------------------------------------------------------------------------------
class LineSegment2 {
 public:
  LineSegment2(const Point2& point1, const Point2& point2);
  float Length() const { return length_; }
  const Point2& FirstPoint() const { return p1_; }
  const Point2& SecondPoint() const { return p2_; }
  const Point2& UnitDirection() const { return unit_direction_; }
  float Heading() const { return std::atan2(unit_direction_.y(), unit_direction_.x()); }

 private:
  Point2 p1_, p2_;
  float length_;
  Point2 unit_direction_;
};

class Polyline2 {
 public:
  Polyline2() : length_(0.0f) {}
  Polyline2(const PointList2& points);
  void AddPoint(const Point2& point);
  float Length() const { return length_; }
  const std::vector<LineSegment2>& Segments() const { return segments_; }
  // Vertices in order (what the device tables are built from).
  PointList2 Points() const;

 private:
  std::vector<LineSegment2> segments_;
  std::vector<float> cumulative_lengths_;
  float length_;
};

// ---------------------------------------------------------------------------------------------
// Costs (include/ilqgames/cost/cost.h:55-96 and the concrete classes cited below)
// ---------------------------------------------------------------------------------------------
class Cost {
 public:
  virtual ~Cost() {}
  void SetWeight(float weight) { weight_ = weight; }
  void ScaleWeight(float scale) { weight_ *= scale; }
  float Weight() const { return weight_; }
  const std::string& Name() const { return name_; }

  // Device description; false = this cost has no gfx950 kernel.
  virtual bool Describe(host::TermDescription* out) const { (void)out; return false; }

 protected:
  explicit Cost(float weight, const std::string& name) : weight_(weight), name_(name) {}
  float weight_;
  std::string name_;
};

class TimeInvariantCost : public Cost {
 protected:
  explicit TimeInvariantCost(float weight, const std::string& name) : Cost(weight, name) {}
};

// include/ilqgames/cost/quadratic_cost.h:55-84 — dim < 0 penalises every dimension.
class QuadraticCost : public TimeInvariantCost {
 public:
  QuadraticCost(float weight, Dimension dim, float nominal = 0.0, const std::string& name = "")
      : TimeInvariantCost(weight, name), dimension_(dim), nominal_(nominal) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  Dimension dimension_;
  float nominal_;
};

// include/ilqgames/cost/semiquadratic_cost.h:54-85
class SemiquadraticCost : public TimeInvariantCost {
 public:
  SemiquadraticCost(float weight, Dimension dim, float threshold, bool oriented_right,
                    const std::string& name = "")
      : TimeInvariantCost(weight, name), dimension_(dim), threshold_(threshold), oriented_right_(oriented_right) {
    CHECK_GE(dimension_, 0);
  }
  bool Describe(host::TermDescription* out) const override;

 private:
  Dimension dimension_;
  float threshold_;
  bool oriented_right_;
};

// include/ilqgames/cost/quadratic_polyline2_cost.h:56-85
class QuadraticPolyline2Cost : public TimeInvariantCost {
 public:
  QuadraticPolyline2Cost(float weight, const Polyline2& polyline,
                         const std::pair<Dimension, Dimension>& position_idxs, const std::string& name = "")
      : TimeInvariantCost(weight, name), polyline_(polyline), xidx_(position_idxs.first), yidx_(position_idxs.second) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  Polyline2 polyline_;
  Dimension xidx_, yidx_;
};

// include/ilqgames/cost/semiquadratic_polyline2_cost.h:57-103
class SemiquadraticPolyline2Cost : public TimeInvariantCost {
 public:
  SemiquadraticPolyline2Cost(float weight, const Polyline2& polyline,
                             const std::pair<Dimension, Dimension>& position_idxs, float threshold,
                             bool oriented_right, const std::string& name = "")
      : TimeInvariantCost(weight, name), polyline_(polyline), xidx_(position_idxs.first),
        yidx_(position_idxs.second), threshold_(threshold), oriented_right_(oriented_right) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  Polyline2 polyline_;
  Dimension xidx_, yidx_;
  float threshold_;
  bool oriented_right_;
};

// include/ilqgames/cost/proximity_cost.h:55-90
class ProximityCost : public TimeInvariantCost {
 public:
  ProximityCost(float weight, const std::pair<Dimension, Dimension>& position_idxs1,
                const std::pair<Dimension, Dimension>& position_idxs2, float threshold,
                const std::string& name = "")
      : TimeInvariantCost(weight, name), threshold_(threshold), xidx1_(position_idxs1.first),
        yidx1_(position_idxs1.second), xidx2_(position_idxs2.first), yidx2_(position_idxs2.second) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  float threshold_;
  Dimension xidx1_, yidx1_, xidx2_, yidx2_;
};

// include/ilqgames/cost/signed_distance_cost.h:56-92
class SignedDistanceCost : public TimeInvariantCost {
 public:
  SignedDistanceCost(const std::pair<Dimension, Dimension>& dims1, const std::pair<Dimension, Dimension>& dims2,
                     float nominal = 0.0, bool less_is_positive = true, const std::string& name = "")
      : TimeInvariantCost(1.0, name), xdim1_(dims1.first), ydim1_(dims1.second), xdim2_(dims2.first),
        ydim2_(dims2.second), nominal_(nominal), less_is_positive_(less_is_positive) {
    CHECK_GE(nominal_, 0.0);
  }
  bool Describe(host::TermDescription* out) const override;

 private:
  Dimension xdim1_, ydim1_, xdim2_, ydim2_;
  float nominal_;
  bool less_is_positive_;
};

// include/ilqgames/cost/polyline2_signed_distance_cost.h:55-93 — signed distance to a polyline minus a nominal
// value (positive on the polyline's right, or on its left when oriented_same_as_polyline is false).
class Polyline2SignedDistanceCost : public TimeInvariantCost {
 public:
  Polyline2SignedDistanceCost(const Polyline2& polyline, const std::pair<Dimension, Dimension>& position_idxs,
                              const float nominal = 0.0, bool oriented_same_as_polyline = true,
                              const std::string& name = "")
      : TimeInvariantCost(1.0, name), polyline_(polyline), xidx_(position_idxs.first), yidx_(position_idxs.second),
        nominal_(nominal), oriented_same_as_polyline_(oriented_same_as_polyline) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  const Polyline2 polyline_;
  const Dimension xidx_, yidx_;
  const float nominal_;
  const bool oriented_same_as_polyline_;
};

// include/ilqgames/cost/quadratic_difference_cost.h:55-75 — 0.5 w |x[dims1] - x[dims2]|^2; the device carries the
// two-pair form the reference's examples use (a position against a position).
class QuadraticDifferenceCost : public TimeInvariantCost {
 public:
  QuadraticDifferenceCost(float weight, const std::vector<Dimension>& dims1, const std::vector<Dimension>& dims2,
                          const std::string& name = "")
      : TimeInvariantCost(weight, name), dims1_(dims1), dims2_(dims2) {
    CHECK_EQ(dims1_.size(), dims2_.size());
  }
  bool Describe(host::TermDescription* out) const override;

 private:
  const std::vector<Dimension> dims1_, dims2_;
};

// include/ilqgames/cost/orientation_cost.h:55-82 — 0.5 w d^2 with d the heading error wrapped into [-pi, pi).
class OrientationCost : public TimeInvariantCost {
 public:
  OrientationCost(float weight, Dimension dim, float nominal = 0.0, const std::string& name = "")
      : TimeInvariantCost(weight, name), dim_(dim), nominal_(nominal) {
    CHECK_GE(dim_, 0);
  }
  bool Describe(host::TermDescription* out) const override;

 private:
  const Dimension dim_;
  const float nominal_;
};

// include/ilqgames/cost/quadratic_norm_cost.h:56-88 — 0.5 w (|(x[d1], x[d2])| - nominal)^2
class QuadraticNormCost : public TimeInvariantCost {
 public:
  QuadraticNormCost(float weight, const std::pair<Dimension, Dimension>& dims, float nominal = 0.0,
                    const std::string& name = "")
      : TimeInvariantCost(weight, name), dim1_(dims.first), dim2_(dims.second), nominal_(nominal) {
    CHECK_GE(dim1_, 0);
    CHECK_GE(dim2_, 0);
  }
  bool Describe(host::TermDescription* out) const override;

 private:
  const Dimension dim1_, dim2_;
  const float nominal_;
};

// include/ilqgames/cost/semiquadratic_norm_cost.h:57-93 — the one-sided version of QuadraticNormCost.
class SemiquadraticNormCost : public TimeInvariantCost {
 public:
  SemiquadraticNormCost(float weight, const std::pair<Dimension, Dimension>& dims, float threshold,
                        bool oriented_right, const std::string& name = "")
      : TimeInvariantCost(weight, name), dim1_(dims.first), dim2_(dims.second), threshold_(threshold),
        oriented_right_(oriented_right) {
    CHECK_GE(dim1_, 0);
    CHECK_GE(dim2_, 0);
  }
  bool Describe(host::TermDescription* out) const override;

 private:
  const Dimension dim1_, dim2_;
  const float threshold_;
  const bool oriented_right_;
};

// include/ilqgames/cost/relative_distance_cost.h:55-78 — w |p1 - p2| (not squared).
class RelativeDistanceCost : public TimeInvariantCost {
 public:
  RelativeDistanceCost(float weight, const std::pair<Dimension, Dimension>& dims1,
                       const std::pair<Dimension, Dimension>& dims2, const std::string& name = "")
      : TimeInvariantCost(weight, name), dims1_(dims1), dims2_(dims2) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  const std::pair<Dimension, Dimension> dims1_, dims2_;
};

// include/ilqgames/cost/locally_convex_proximity_cost.h:54-86 — 0.5 w min((thr - |dx|)^2, (thr - |dy|)^2) inside the
// square |dx|, |dy| < thr around the other player.
class LocallyConvexProximityCost : public TimeInvariantCost {
 public:
  LocallyConvexProximityCost(float weight, const std::pair<Dimension, Dimension>& position_idxs1,
                             const std::pair<Dimension, Dimension>& position_idxs2, float threshold,
                             const std::string& name = "")
      : TimeInvariantCost(weight, name), threshold_(threshold), xidx1_(position_idxs1.first),
        yidx1_(position_idxs1.second), xidx2_(position_idxs2.first), yidx2_(position_idxs2.second) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  const float threshold_;
  const Dimension xidx1_, yidx1_, xidx2_, yidx2_;
};

// include/ilqgames/cost/weighted_convex_proximity_cost.h:54-92 — LocallyConvexProximityCost scaled by v1^2 + v2^2.
class WeightedConvexProximityCost : public TimeInvariantCost {
 public:
  WeightedConvexProximityCost(float weight, const std::pair<Dimension, Dimension>& position_idxs1,
                              const std::pair<Dimension, Dimension>& position_idxs2, Dimension vidx1, Dimension vidx2,
                              float threshold, const std::string& name = "")
      : TimeInvariantCost(weight, name), threshold_(threshold), xidx1_(position_idxs1.first),
        yidx1_(position_idxs1.second), vidx1_(vidx1), xidx2_(position_idxs2.first), yidx2_(position_idxs2.second),
        vidx2_(vidx2) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  const float threshold_;
  const Dimension xidx1_, yidx1_, vidx1_;
  const Dimension xidx2_, yidx2_, vidx2_;
};

// include/ilqgames/cost/curvature_cost.h:53-74 — 0.5 w (omega / v)^2
class CurvatureCost : public TimeInvariantCost {
 public:
  CurvatureCost(float weight, Dimension omega_idx, Dimension v_idx, const std::string& name = "")
      : TimeInvariantCost(weight, name), omega_idx_(omega_idx), v_idx_(v_idx) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  const Dimension omega_idx_, v_idx_;
};

// include/ilqgames/cost/nominal_path_length_cost.h:54-77 — 0.5 w (x[dim] - t nominal_speed)^2, t the time the solver
// hands to the cost (relative to the start of the window, src/ilq_solver.cpp:186,236).
class NominalPathLengthCost : public Cost {
 public:
  NominalPathLengthCost(float weight, Dimension dim, float nominal_speed, const std::string& name = "")
      : Cost(weight, name), dimension_(dim), nominal_speed_(nominal_speed) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  const Dimension dimension_;
  const float nominal_speed_;
};

// include/ilqgames/cost/route_progress_cost.h:55-96 — 0.5 w |p - route(initial_route_pos + t nominal_speed)|^2
class RouteProgressCost : public Cost {
 public:
  RouteProgressCost(float weight, float nominal_speed, const Polyline2& polyline,
                    const std::pair<Dimension, Dimension>& position_idxs, const std::string& name = "",
                    float initial_route_pos = 0.0)
      : Cost(weight, name), nominal_speed_(nominal_speed), polyline_(polyline), xidx_(position_idxs.first),
        yidx_(position_idxs.second), initial_route_pos_(initial_route_pos) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  const float nominal_speed_;
  const Polyline2 polyline_;
  const Dimension xidx_, yidx_;
  const float initial_route_pos_;
};

// include/ilqgames/cost/final_time_cost.h:55-88 — another cost, switched on from `threshold_time` (relative to the
// start of the window) onwards; zero value and derivatives before.
class FinalTimeCost : public Cost {
 public:
  FinalTimeCost(const std::shared_ptr<const Cost>& cost, Time threshold_time, const std::string& name = "")
      : Cost(0.0, name), cost_(cost), threshold_time_(threshold_time) {
    CHECK_NOTNULL(cost.get());
  }
  bool Describe(host::TermDescription* out) const override;

 private:
  const std::shared_ptr<const Cost> cost_;
  const Time threshold_time_;
};

// include/ilqgames/cost/extreme_value_cost.h:56-88
class ExtremeValueCost : public Cost {
 public:
  ExtremeValueCost(const std::vector<std::shared_ptr<const Cost>>& costs, bool is_min,
                   const std::string& name = "")
      : Cost(1.0, name), costs_(costs), is_min_(is_min) {
    CHECK_GT(costs_.size(), 0);
  }
  bool Describe(host::TermDescription* out) const override;

 private:
  std::vector<std::shared_ptr<const Cost>> costs_;
  bool is_min_;
};

// ---------------------------------------------------------------------------------------------
// Constraints (include/ilqgames/constraint/constraint.h:62-155).  Multipliers live per instance
// on the device (one lambda per constraint and time step, one mu per instance), not in these
// objects and not in a process-global as the reference's Constraint::mu_.
// ---------------------------------------------------------------------------------------------
class Constraint : public Cost {
 public:
  bool IsEquality() const { return is_equality_; }

 protected:
  explicit Constraint(bool is_equality, const std::string& name) : Cost(1.0, name), is_equality_(is_equality) {}
  bool is_equality_;
};

class TimeInvariantConstraint : public Constraint {
 protected:
  explicit TimeInvariantConstraint(bool is_equality, const std::string& name) : Constraint(is_equality, name) {}
};

// include/ilqgames/constraint/proximity_constraint.h:56-92
class ProximityConstraint : public TimeInvariantConstraint {
 public:
  ProximityConstraint(const std::pair<Dimension, Dimension>& dims1, const std::pair<Dimension, Dimension>& dims2,
                      float threshold, bool keep_within, const std::string& name = "")
      : TimeInvariantConstraint(false, name), xidx1_(dims1.first), yidx1_(dims1.second), xidx2_(dims2.first),
        yidx2_(dims2.second), threshold_(threshold), keep_within_(keep_within) {
    CHECK_GT(threshold_, 0.0);
  }
  bool Describe(host::TermDescription* out) const override;

 private:
  Dimension xidx1_, yidx1_, xidx2_, yidx2_;
  float threshold_;
  bool keep_within_;
};

// include/ilqgames/constraint/single_dimension_constraint.h:54-104
class SingleDimensionConstraint : public TimeInvariantConstraint {
 public:
  SingleDimensionConstraint(Dimension dim, float threshold, bool keep_below, const std::string& name = "")
      : TimeInvariantConstraint(false, name), dim_(dim), threshold_(threshold), keep_below_(keep_below) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  Dimension dim_;
  float threshold_;
  bool keep_below_;
};

// include/ilqgames/constraint/affine_scalar_constraint.h:54-100 — g(x) = a^T x - b on the whole input vector.
class AffineScalarConstraint : public TimeInvariantConstraint {
 public:
  AffineScalarConstraint(const VectorXf& a, float b, bool is_equality, const std::string& name = "")
      : TimeInvariantConstraint(is_equality, name), a_(a), b_(b) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  const VectorXf a_;
  const float b_;
};

// include/ilqgames/constraint/affine_vector_constraint.h:52-112 — g(x) = |A x - b|, A square (its Quadraticize CHECKs
// input.size() == b.size()).
class AffineVectorConstraint : public TimeInvariantConstraint {
 public:
  AffineVectorConstraint(const MatrixXf& A, const VectorXf& b, bool is_equality, const std::string& name = "")
      : TimeInvariantConstraint(is_equality, name), A_(A), b_(b) {
    CHECK_EQ(A_.rows(), b_.size());
  }
  bool Describe(host::TermDescription* out) const override;

 private:
  const MatrixXf A_;
  const VectorXf b_;
};

// include/ilqgames/constraint/polyline2_signed_distance_constraint.h:57-91 — g = signed distance to the polyline
// minus a threshold, <= 0 when keep_left and >= 0 otherwise.
class Polyline2SignedDistanceConstraint : public TimeInvariantConstraint {
 public:
  Polyline2SignedDistanceConstraint(const Polyline2& polyline, const std::pair<Dimension, Dimension>& dims,
                                    float threshold, bool keep_left, const std::string& name = "")
      : TimeInvariantConstraint(false, name), polyline_(polyline), xidx_(dims.first), yidx_(dims.second),
        threshold_(threshold), keep_left_(keep_left) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  Polyline2 polyline_;
  Dimension xidx_, yidx_;
  float threshold_;
  bool keep_left_;
};

// include/ilqgames/constraint/final_time_constraint.h:55-90 — another constraint, switched on from `threshold_time`
// (relative to the start of the window) onwards: g = 0 and no derivatives before.
class FinalTimeConstraint : public Constraint {
 public:
  FinalTimeConstraint(const std::shared_ptr<Constraint>& constraint, Time threshold_time)
      : Constraint(*CHECK_NOTNULL(constraint.get())), constraint_(constraint), threshold_time_(threshold_time) {}
  bool Describe(host::TermDescription* out) const override;

 private:
  const std::shared_ptr<Constraint> constraint_;
  const Time threshold_time_;
};

// ---------------------------------------------------------------------------------------------
// PlayerCost (include/ilqgames/cost/player_cost.h:60-170)
// ---------------------------------------------------------------------------------------------
class PlayerCost {
 public:
  explicit PlayerCost(const std::string& name = "", float state_regularization = 0.0,
                      float control_regularization = 0.0)
      : name_(name), state_regularization_(state_regularization),
        control_regularization_(control_regularization), cost_structure_(SUM), time_of_extreme_cost_(0) {}

  void AddStateCost(const std::shared_ptr<Cost>& cost) { state_costs_.push_back(cost); }
  void AddControlCost(PlayerIndex idx, const std::shared_ptr<Cost>& cost) { control_costs_.emplace(idx, cost); }
  void AddStateConstraint(const std::shared_ptr<Constraint>& constraint) { state_constraints_.push_back(constraint); }
  void AddControlConstraint(PlayerIndex idx, const std::shared_ptr<Constraint>& constraint) {
    control_constraints_.emplace(idx, constraint);
  }

  enum CostStructure { SUM, MAX, MIN };
  void SetTimeAdditive() { cost_structure_ = SUM; }
  void SetMaxOverTime() { cost_structure_ = MAX; }
  void SetMinOverTime() { cost_structure_ = MIN; }
  bool IsTimeAdditive() const { return cost_structure_ == SUM; }
  bool IsMaxOverTime() const { return cost_structure_ == MAX; }
  bool IsMinOverTime() const { return cost_structure_ == MIN; }
  size_t TimeOfExtremeCost() { return time_of_extreme_cost_; }
  void SetTimeOfExtremeCost(size_t kk) { time_of_extreme_cost_ = kk; }

  const PtrVector<Cost>& StateCosts() const { return state_costs_; }
  const PlayerPtrMultiMap<Cost>& ControlCosts() const { return control_costs_; }
  const PtrVector<Constraint>& StateConstraints() const { return state_constraints_; }
  const PlayerPtrMultiMap<Constraint>& ControlConstraints() const { return control_constraints_; }
  bool IsConstrained() const { return !state_constraints_.empty() || !control_constraints_.empty(); }
  float StateRegularization() const { return state_regularization_; }
  float ControlRegularization() const { return control_regularization_; }
  CostStructure Structure() const { return cost_structure_; }
  const std::string& Name() const { return name_; }

 private:
  std::string name_;
  PtrVector<Cost> state_costs_;
  PlayerPtrMultiMap<Cost> control_costs_;
  PtrVector<Constraint> state_constraints_;
  PlayerPtrMultiMap<Constraint> control_constraints_;
  float state_regularization_;
  float control_regularization_;
  CostStructure cost_structure_;
  size_t time_of_extreme_cost_;
};

// ---------------------------------------------------------------------------------------------
// Dynamics (include/ilqgames/dynamics/*.h).  Subsystems name a device model (ilqg_dyn_kind).
// ---------------------------------------------------------------------------------------------
class SinglePlayerDynamicalSystem {
 public:
  virtual ~SinglePlayerDynamicalSystem() {}
  Dimension XDim() const { return xdim_; }
  Dimension UDim() const { return udim_; }
  virtual std::vector<Dimension> PositionDimensions() const = 0;
  // Device model of this subsystem; kind 0 = none.
  virtual ilqg_subsystem Describe() const { return ilqg_subsystem{0, xdim_, udim_, 0.0f}; }

 protected:
  SinglePlayerDynamicalSystem(Dimension xdim, Dimension udim) : xdim_(xdim), udim_(udim) {}
  const Dimension xdim_;
  const Dimension udim_;
};

// include/ilqgames/dynamics/single_player_unicycle_4d.h:55-88
class SinglePlayerUnicycle4D : public SinglePlayerDynamicalSystem {
 public:
  SinglePlayerUnicycle4D() : SinglePlayerDynamicalSystem(kNumXDims, kNumUDims) {}
  std::vector<Dimension> PositionDimensions() const override { return {kPxIdx, kPyIdx}; }
  ilqg_subsystem Describe() const override { return ilqg_subsystem{ILQG_DYN_UNICYCLE_4D, xdim_, udim_, 0.0f}; }
  static const Dimension kNumXDims, kPxIdx, kPyIdx, kThetaIdx, kVIdx;
  static const Dimension kNumUDims, kOmegaIdx, kAIdx;
};

// include/ilqgames/dynamics/single_player_dubins_car.h:57-91 — planar car at constant speed, one control (turn rate)
class SinglePlayerDubinsCar : public SinglePlayerDynamicalSystem {
 public:
  SinglePlayerDubinsCar(float v) : SinglePlayerDynamicalSystem(kNumXDims, kNumUDims), v_(v) { CHECK_GT(v_, 0.0); }
  std::vector<Dimension> PositionDimensions() const override { return {kPxIdx, kPyIdx}; }
  ilqg_subsystem Describe() const override { return ilqg_subsystem{ILQG_DYN_DUBINS_CAR, xdim_, udim_, v_}; }
  static const Dimension kNumXDims, kPxIdx, kPyIdx, kThetaIdx;
  static const Dimension kNumUDims, kOmegaIdx;

 private:
  const float v_;
};

// include/ilqgames/dynamics/single_player_point_mass_2d.h:55-86 — planar double integrator
class SinglePlayerPointMass2D : public SinglePlayerDynamicalSystem {
 public:
  SinglePlayerPointMass2D() : SinglePlayerDynamicalSystem(kNumXDims, kNumUDims) {}
  std::vector<Dimension> PositionDimensions() const override { return {kPxIdx, kPyIdx}; }
  ilqg_subsystem Describe() const override { return ilqg_subsystem{ILQG_DYN_POINT_MASS_2D, xdim_, udim_, 0.0f}; }
  static const Dimension kNumXDims, kPxIdx, kPyIdx, kVxIdx, kVyIdx;
  static const Dimension kNumUDims, kAxIdx, kAyIdx;
};

// include/ilqgames/dynamics/single_player_car_5d.h:59-98
class SinglePlayerCar5D : public SinglePlayerDynamicalSystem {
 public:
  SinglePlayerCar5D(float inter_axle_distance)
      : SinglePlayerDynamicalSystem(kNumXDims, kNumUDims), inter_axle_distance_(inter_axle_distance) {}
  std::vector<Dimension> PositionDimensions() const override { return {kPxIdx, kPyIdx}; }
  ilqg_subsystem Describe() const override {
    return ilqg_subsystem{ILQG_DYN_CAR_5D, xdim_, udim_, inter_axle_distance_};
  }
  static const Dimension kNumXDims, kPxIdx, kPyIdx, kThetaIdx, kPhiIdx, kVIdx;
  static const Dimension kNumUDims, kOmegaIdx, kAIdx;

 private:
  const float inter_axle_distance_;
};

// include/ilqgames/dynamics/single_player_car_6d.h:60-100
class SinglePlayerCar6D : public SinglePlayerDynamicalSystem {
 public:
  SinglePlayerCar6D(float inter_axle_distance)
      : SinglePlayerDynamicalSystem(kNumXDims, kNumUDims), inter_axle_distance_(inter_axle_distance) {}
  std::vector<Dimension> PositionDimensions() const override { return {kPxIdx, kPyIdx}; }
  ilqg_subsystem Describe() const override {
    return ilqg_subsystem{ILQG_DYN_CAR_6D, xdim_, udim_, inter_axle_distance_};
  }
  static const Dimension kNumXDims, kPxIdx, kPyIdx, kThetaIdx, kPhiIdx, kVIdx, kAIdx;
  static const Dimension kNumUDims, kOmegaIdx, kJerkIdx;

 private:
  const float inter_axle_distance_;
};

// include/ilqgames/dynamics/single_player_unicycle_5d.h:55-88 — the unicycle with a path-length state.
class SinglePlayerUnicycle5D : public SinglePlayerDynamicalSystem {
 public:
  SinglePlayerUnicycle5D() : SinglePlayerDynamicalSystem(kNumXDims, kNumUDims) {}
  std::vector<Dimension> PositionDimensions() const override { return {kPxIdx, kPyIdx}; }
  ilqg_subsystem Describe() const override { return ilqg_subsystem{ILQG_DYN_UNICYCLE_5D, xdim_, udim_, 0.0f}; }
  static const Dimension kNumXDims, kPxIdx, kPyIdx, kThetaIdx, kVIdx, kSIdx;
  static const Dimension kNumUDims, kOmegaIdx, kAIdx;
};

// include/ilqgames/dynamics/single_player_car_7d.h:61-100 — Car5D with curvature and path-length states.
class SinglePlayerCar7D : public SinglePlayerDynamicalSystem {
 public:
  SinglePlayerCar7D(float inter_axle_distance)
      : SinglePlayerDynamicalSystem(kNumXDims, kNumUDims), inter_axle_distance_(inter_axle_distance) {}
  std::vector<Dimension> PositionDimensions() const override { return {kPxIdx, kPyIdx}; }
  ilqg_subsystem Describe() const override {
    return ilqg_subsystem{ILQG_DYN_CAR_7D, xdim_, udim_, inter_axle_distance_};
  }
  static const Dimension kNumXDims, kPxIdx, kPyIdx, kThetaIdx, kPhiIdx, kVIdx, kKappaIdx, kSIdx;
  static const Dimension kNumUDims, kOmegaIdx, kAIdx;

 private:
  const float inter_axle_distance_;
};

// include/ilqgames/dynamics/single_player_delayed_dubins_car.h:57-99 — constant speed, the turn rate is a state.
class SinglePlayerDelayedDubinsCar : public SinglePlayerDynamicalSystem {
 public:
  SinglePlayerDelayedDubinsCar(float v) : SinglePlayerDynamicalSystem(kNumXDims, kNumUDims), v_(v) { CHECK_GT(v_, 0.0); }
  std::vector<Dimension> PositionDimensions() const override { return {kPxIdx, kPyIdx}; }
  ilqg_subsystem Describe() const override { return ilqg_subsystem{ILQG_DYN_DELAYED_DUBINS_CAR, xdim_, udim_, v_}; }
  static const Dimension kNumXDims, kPxIdx, kPyIdx, kThetaIdx, kOmegaIdx;
  static const Dimension kNumUDims, kAlphaIdx;

 private:
  const float v_;
};

// include/ilqgames/dynamics/multi_player_integrable_system.h:57-140 (shape queries only; the
// RK4 integrator is ilqg_rollout_batch on the device).
struct OperatingPoint;
struct Strategy;
class MultiPlayerIntegrableSystem {
 public:
  virtual ~MultiPlayerIntegrableSystem() {}
  virtual bool TreatAsLinear() const { return false; }
  Dimension XDim() const { return xdim_; }
  Dimension TotalUDim() const {
    Dimension total = 0;
    for (PlayerIndex ii = 0; ii < NumPlayers(); ii++) total += UDim(ii);
    return total;
  }
  virtual Dimension UDim(PlayerIndex player_idx) const = 0;
  virtual PlayerIndex NumPlayers() const = 0;
  virtual std::vector<Dimension> PositionDimensions() const = 0;
  // The reference toggles Euler / RK4 globally (multi_player_integrable_system.h:118-120); the device
  // integrator is RK4 with two sub-steps, the reference default.
  static bool IntegrationUsesEuler() { return false; }
  // Integrate x0 from time t0 to t under the given plan's strategies (multi_player_integrable_system.h:75-79,
  // src/multi_player_integrable_system.cpp:54-74): partial step to the next plan time, whole steps, partial last
  // step.  Runs on the device (ilqg_plan_integrate_batch); concatenated systems only.
  VectorXf Integrate(Time t0, Time t, const VectorXf& x0, const OperatingPoint& operating_point,
                     const std::vector<Strategy>& strategies) const;

 protected:
  MultiPlayerIntegrableSystem(Dimension xdim) : xdim_(xdim) {}
  const Dimension xdim_;
};

class MultiPlayerDynamicalSystem : public MultiPlayerIntegrableSystem {
 protected:
  MultiPlayerDynamicalSystem(Dimension xdim) : MultiPlayerIntegrableSystem(xdim) {}
};

// include/ilqgames/dynamics/two_player_unicycle_4d.h:57-100 — one unicycle, two players: player 1 steers and
// accelerates, player 2 adds a planar velocity disturbance.  On the device: the row pair
// (ILQG_DYN_UNICYCLE_4D_DISTURBED, ILQG_DYN_PLANAR_DISTURBANCE).
class TwoPlayerUnicycle4D : public MultiPlayerDynamicalSystem {
 public:
  TwoPlayerUnicycle4D() : MultiPlayerDynamicalSystem(kNumXDims) {}
  std::vector<Dimension> PositionDimensions() const override { return {kPxIdx, kPyIdx}; }
  Dimension UDim(PlayerIndex player_idx) const override { return (player_idx == 0) ? kNumU1Dims : kNumU2Dims; }
  PlayerIndex NumPlayers() const override { return kNumPlayers; }
  static const Dimension kNumXDims, kPxIdx, kPyIdx, kThetaIdx, kVIdx;
  static const PlayerIndex kNumPlayers;
  static const Dimension kNumU1Dims, kOmegaIdx, kAIdx;
  static const Dimension kNumU2Dims, kDxIdx, kDyIdx;
};

// include/ilqgames/dynamics/air_3d.h:64-110 — the classic two-aircraft pursuit-evasion model in the evader's frame:
// x = (rx, ry, rtheta), player 1 turns the evader, player 2 the pursuer.  On the device: the row pair
// (ILQG_DYN_AIR_3D_EVADER, ILQG_DYN_AIR_3D_PURSUER).
class Air3D : public MultiPlayerDynamicalSystem {
 public:
  Air3D(float evader_speed, float pursuer_speed)
      : MultiPlayerDynamicalSystem(kNumXDims), evader_speed_(evader_speed), pursuer_speed_(pursuer_speed) {}
  Dimension UDim(PlayerIndex player_idx) const override { return (player_idx == 0) ? kNumU1Dims : kNumU2Dims; }
  PlayerIndex NumPlayers() const override { return kNumPlayers; }
  std::vector<Dimension> PositionDimensions() const override { return {kRxIdx, kRyIdx}; }
  const float evader_speed_;
  const float pursuer_speed_;
  static const Dimension kNumXDims, kRxIdx, kRyIdx, kRThetaIdx;
  static const PlayerIndex kNumPlayers;
  static const Dimension kNumU1Dims, kOmega1Idx;
  static const Dimension kNumU2Dims, kOmega2Idx;
};

using SubsystemList = std::vector<std::shared_ptr<SinglePlayerDynamicalSystem>>;

// include/ilqgames/dynamics/concatenated_dynamical_system.h:57-104
class ConcatenatedDynamicalSystem : public MultiPlayerDynamicalSystem {
 public:
  ConcatenatedDynamicalSystem(const SubsystemList& subsystems);
  const SubsystemList& Subsystems() const { return subsystems_; }
  PlayerIndex NumPlayers() const override { return static_cast<PlayerIndex>(subsystems_.size()); }
  Dimension SubsystemStartDim(PlayerIndex player_idx) const { return subsystem_start_dims_[player_idx]; }
  Dimension SubsystemXDim(PlayerIndex player_idx) const { return subsystems_[player_idx]->XDim(); }
  Dimension UDim(PlayerIndex player_idx) const override { return subsystems_[player_idx]->UDim(); }
  std::vector<Dimension> PositionDimensions() const override;

 private:
  const SubsystemList subsystems_;
  std::vector<Dimension> subsystem_start_dims_;
};

// ---------------------------------------------------------------------------------------------
// Trajectory containers (include/ilqgames/utils/{strategy,operating_point,
// linear_dynamics_approximation,quadratic_cost_approximation,solver_log}.h)
// ---------------------------------------------------------------------------------------------
struct Strategy {
  std::vector<MatrixXf> Ps;
  std::vector<VectorXf> alphas;
  Strategy(size_t horizon, Dimension xdim, Dimension udim) : Ps(horizon), alphas(horizon) {
    for (size_t kk = 0; kk < horizon; kk++) {
      Ps[kk] = MatrixXf::Zero(udim, xdim);
      alphas[kk] = VectorXf::Zero(udim);
    }
  }
  VectorXf operator()(size_t time_index, const VectorXf& delta_x, const VectorXf& u_ref) const {
    return u_ref - Ps[time_index] * delta_x - alphas[time_index];
  }
  size_t NumVariables() const { return Ps.size() * (Ps.front().size() + alphas.front().size()); }
};

struct OperatingPoint {
  std::vector<VectorXf> xs;
  std::vector<std::vector<VectorXf>> us;
  Time t0;
  OperatingPoint(size_t num_time_steps, PlayerIndex num_players, Time initial_time)
      : xs(num_time_steps), us(num_time_steps, std::vector<VectorXf>(num_players)), t0(initial_time) {}
  template <typename MultiPlayerSystemType>
  OperatingPoint(size_t num_time_steps, Time initial_time,
                 const std::shared_ptr<const MultiPlayerSystemType>& dynamics)
      : OperatingPoint(num_time_steps, dynamics->NumPlayers(), initial_time) {
    for (size_t kk = 0; kk < num_time_steps; kk++) {
      xs[kk] = VectorXf::Zero(dynamics->XDim());
      for (PlayerIndex ii = 0; ii < dynamics->NumPlayers(); ii++) us[kk][ii] = VectorXf::Zero(dynamics->UDim(ii));
    }
  }
  void swap(OperatingPoint& other) {
    xs.swap(other.xs);
    us.swap(other.us);
    std::swap(t0, other.t0);
  }
};

struct LinearDynamicsApproximation {
  MatrixXf A;
  std::vector<MatrixXf> Bs;
  LinearDynamicsApproximation() {}
  template <typename MultiPlayerSystemType>
  explicit LinearDynamicsApproximation(const MultiPlayerSystemType& system)
      : A(MatrixXf::Identity(system.XDim(), system.XDim())), Bs(system.NumPlayers()) {
    for (size_t ii = 0; ii < system.NumPlayers(); ii++) Bs[ii] = MatrixXf::Zero(system.XDim(), system.UDim(ii));
  }
};

struct SingleCostApproximation {
  MatrixXf hess;
  VectorXf grad;
  SingleCostApproximation(const MatrixXf& hessian, const VectorXf& gradient) : hess(hessian), grad(gradient) {
    CHECK_EQ(hess.rows(), hess.cols());
    CHECK_EQ(hess.rows(), grad.size());
  }
  SingleCostApproximation(Dimension dim, float regularization = 0.0)
      : hess(regularization * MatrixXf::Identity(dim, dim)), grad(VectorXf::Zero(dim)) {}
};

struct QuadraticCostApproximation {
  SingleCostApproximation state;
  // Ordered so the (i, j) block list handed to the device is deterministic.
  std::vector<std::pair<PlayerIndex, SingleCostApproximation>> control;
  explicit QuadraticCostApproximation(Dimension xdim, float regularization = 0.0) : state(xdim, regularization) {}
  SingleCostApproximation& Control(PlayerIndex jj, Dimension udim) {
    for (auto& e : control)
      if (e.first == jj) return e.second;
    control.emplace_back(jj, SingleCostApproximation(udim));
    return control.back().second;
  }
};

// include/ilqgames/utils/solver_log.h:62-160 — the device solve keeps no per-iteration deep copies
// (that is the point of running it on the GPU), so a log holds the final iterate of each solve.
class SolverLog {
 public:
  SolverLog() {}
  void AddSolverIterate(const OperatingPoint& operating_point, const std::vector<Strategy>& strategies,
                        const std::vector<float>& total_costs, Time cumulative_runtime, bool was_converged) {
    operating_points_.push_back(operating_point);
    strategies_.push_back(strategies);
    total_player_costs_.push_back(total_costs);
    cumulative_runtimes_.push_back(cumulative_runtime);
    was_converged_.push_back(was_converged);
  }
  bool WasConverged() const { return was_converged_.back(); }
  bool WasConverged(size_t idx) const { return was_converged_[idx]; }
  PlayerIndex NumPlayers() const { return static_cast<PlayerIndex>(strategies_[0].size()); }
  size_t NumIterates() const { return operating_points_.size(); }
  std::vector<float> TotalCosts() const { return total_player_costs_.back(); }
  const std::vector<Strategy>& InitialStrategies() const { return strategies_.front(); }
  const OperatingPoint& InitialOperatingPoint() const { return operating_points_.front(); }
  const std::vector<Strategy>& FinalStrategies() const { return strategies_.back(); }
  const OperatingPoint& FinalOperatingPoint() const { return operating_points_.back(); }
  Time CumulativeRuntime() const { return cumulative_runtimes_.back(); }
  // On-disk layout of the reference (src/solver_log.cpp:113-171), read by its matlab/ scripts and GUI:
  //   <log dir>/<experiment>/<iterate>/{t0,xs,u<player>,costs,cumulative_runtimes}.txt
  // one trajectory row per line, Eigen's default stream formatting (6 significant digits, columns right-
  // aligned to a common width, single-space separator).  <log dir> = $ILQGAMES_LOG_DIR, default "./logs"
  // (the reference bakes its source tree's logs/ in at configure time).  Directories that already exist make
  // Save() fail, as MakeDirectory does there.
  bool Save(bool only_last_trajectory = false, const std::string& experiment_name = DefaultExperimentName()) const;
  static std::string DefaultExperimentName();
  // Outer iterations the device performed for the solve this log belongs to.
  int DeviceIterations() const { return device_iterations_; }
  void SetDeviceIterations(int iters) { device_iterations_ = iters; }

 private:
  std::vector<OperatingPoint> operating_points_;
  std::vector<std::vector<Strategy>> strategies_;
  std::vector<std::vector<float>> total_player_costs_;
  std::vector<Time> cumulative_runtimes_;
  std::vector<bool> was_converged_;
  int device_iterations_ = 0;
};

bool SaveLogs(const std::vector<SolverLog>& logs, bool only_last_trajectory = true,
              const std::string& experiment_name = SolverLog::DefaultExperimentName());
bool SaveLogs(const std::vector<std::shared_ptr<const SolverLog>>& logs, bool only_last_trajectory = true,
              const std::string& experiment_name = SolverLog::DefaultExperimentName());
bool MakeDirectory(const std::string& directory_name);  // include/ilqgames/utils/make_directory.h

// ---------------------------------------------------------------------------------------------
// SolverParams (include/ilqgames/solver/solver_params.h:50-106)
// ---------------------------------------------------------------------------------------------
struct SolverParams {
  float convergence_tolerance = 1e-1;
  size_t max_solver_iters = 1000;
  bool linesearch = true;
  float initial_alpha_scaling = 0.5;
  float geometric_alpha_scaling = 0.5;
  size_t max_backtracking_steps = 10;
  float expected_decrease_fraction = 0.1;
  bool open_loop = false;
  float state_regularization = 0.0;
  float control_regularization = 0.0;
  size_t unconstrained_solver_max_iters = 10;
  float geometric_mu_scaling = 1.1;
  float geometric_mu_downscaling = 0.5;
  float geometric_lambda_downscaling = 0.5;
  float constraint_error_tolerance = 1e-1;
  bool reset_problem = true;
  bool reset_lambdas = true;
  bool reset_mu = true;
};

// ---------------------------------------------------------------------------------------------
// Problem (include/ilqgames/solver/problem.h:61-185)
// ---------------------------------------------------------------------------------------------
class Problem {
 public:
  virtual ~Problem() {}
  virtual void Initialize() {
    ConstructDynamics();
    ConstructPlayerCosts();
    ConstructInitialState();
    ConstructInitialOperatingPoint();
    ConstructInitialStrategies();
    initialized_ = true;
  }
  void ResetInitialTime(Time t0) {
    CHECK(initialized_);
    operating_point_->t0 = t0;
  }
  void ResetInitialState(const VectorXf& x0) {
    CHECK(initialized_);
    x0_ = x0;
  }
  // Receding-horizon re-sync (src/problem.cpp:127-186): integrates x0 forward under the stored strategies by about
  // planner_runtime, re-anchors the stored plan at the nearest state, zero-extends and re-propagates its tail, and
  // sets InitialState() / InitialTime() for the next solve.  Runs on the device (ilqg_receding_horizon_shift_batch).
  virtual void SetUpNextRecedingHorizon(const VectorXf& x0, Time t0, Time planner_runtime = 0.1);
  virtual void OverwriteSolution(const OperatingPoint& operating_point, const std::vector<Strategy>& strategies);
  bool IsConstrained() const;
  virtual Time InitialTime() const { return operating_point_->t0; }
  const VectorXf& InitialState() const { return x0_; }
  std::vector<PlayerCost>& PlayerCosts() { return player_costs_; }
  const std::vector<PlayerCost>& PlayerCosts() const { return player_costs_; }
  const std::shared_ptr<const MultiPlayerIntegrableSystem>& Dynamics() const { return dynamics_; }
  virtual const OperatingPoint& CurrentOperatingPoint() const { return *operating_point_; }
  virtual const std::vector<Strategy>& CurrentStrategies() const { return *strategies_; }

 protected:
  Problem() : initialized_(false) {}
  virtual void ConstructDynamics() = 0;
  virtual void ConstructPlayerCosts() = 0;
  virtual void ConstructInitialState() = 0;
  virtual void ConstructInitialOperatingPoint() {
    operating_point_.reset(new OperatingPoint(time::kNumTimeSteps, 0.0, dynamics_));
  }
  virtual void ConstructInitialStrategies() {
    strategies_.reset(new std::vector<Strategy>());
    for (PlayerIndex ii = 0; ii < dynamics_->NumPlayers(); ii++)
      strategies_->emplace_back(time::kNumTimeSteps, dynamics_->XDim(), dynamics_->UDim(ii));
  }

  std::shared_ptr<const MultiPlayerIntegrableSystem> dynamics_;
  std::vector<PlayerCost> player_costs_;
  VectorXf x0_;
  std::unique_ptr<OperatingPoint> operating_point_;
  std::unique_ptr<std::vector<Strategy>> strategies_;
  bool initialized_;
};

// include/ilqgames/solver/top_down_renderable_problem.h:52-65
class TopDownRenderableProblem : public Problem {
 public:
  virtual ~TopDownRenderableProblem() {}
  virtual std::vector<float> Xs(const VectorXf& x) const = 0;
  virtual std::vector<float> Ys(const VectorXf& x) const = 0;
  virtual std::vector<float> Thetas(const VectorXf& x) const = 0;

 protected:
  TopDownRenderableProblem() : Problem() {}
};

// ---------------------------------------------------------------------------------------------
// Device binding helpers
// ---------------------------------------------------------------------------------------------
namespace host {

// Arithmetic type of the device solve (the containers above stay float, as in the reference).
struct DeviceOptions {
  ilqg_dtype dtype = ILQG_F64;
  // RecedingHorizonSimulator: time charged per solver call.  Negative = the wall clock, as the reference does
  // (src/receding_horizon_simulator.cpp:110-114); a fixed value makes runs reproducible.
  Time simulated_solve_time = -1.0;
  // GameSolver::Solve keeps the iterates of the solve in its SolverLog as the reference does (src/ilq_solver.cpp:111,
  // 164: one AddSolverIterate per accepted step) — copied on the device into an ilqg_iterate_log of this many
  // slots per solve; iterates past it are dropped and the final one is appended.  0 = the final iterate only.
  // SolveBatch logs the final iterate of each instance.
  int logged_iterates = 64;
};
DeviceOptions& Options();

// Flattened, self-owning form of a Problem: the arrays `desc` points into live in this object.
struct ProblemDescription {
  ilqg_problem_desc desc{};
  std::vector<ilqg_cost_term> terms;
  std::vector<int32_t> polyline_offsets;
  std::vector<float> polyline_points;
  std::vector<ilqg_pair> pairs;  // (i, j) control blocks in PlayerCost first-touch order
  int num_constraints = 0;
  std::vector<float> dense_params;  // coefficient blocks of the affine constraints
};

// Walks Problem::Dynamics() and Problem::PlayerCosts() (after Initialize()) and fills the POD
// descriptor of include/ilqg.h.  Returns false and sets *why when some object has no device kernel.
bool DescribeProblem(const Problem& problem, const SolverParams& params, ilqg_dtype dtype,
                     ProblemDescription* out, std::string* why);

// The dynamics rows of a descriptor (ConcatenatedDynamicalSystem of built-in subsystems, or TwoPlayerUnicycle4D).
bool DescribeDynamics(const MultiPlayerIntegrableSystem& dynamics, ilqg_problem_desc* desc, std::string* why);

// One line per subsystem / player cost / term / polyline; read back by the test harness
// (ilqgames_amd/abi.py: ProblemSpec.from_dump).
std::string DumpDescription(const ProblemDescription& description);

class DeviceSolve;  // device buffers + ilqg_problem handle (src: ilqgames_amd/host/ilqgames_host.cpp)

// Result of one instance of a batched solve.
struct BatchResult {
  std::vector<std::shared_ptr<SolverLog>> logs;
  std::vector<bool> success;
};

// ---- several GPUs: one process per GPU, instances sharded, one RCCL all-gather of the results ----
// Game instances share nothing, so a batch is cut into contiguous blocks, block r to rank r; the first total % world
// ranks take one instance more (the same rule as ilqgames_amd/sharding.py::instance_range).
void InstanceRange(size_t total, int rank, int world, size_t* lo, size_t* hi);

// Where this process stands among the processes of a job, as the launcher's environment says (the convention of
// torch.distributed.run / mpirun wrappers): RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR; the rendezvous port is
// ILQG_RENDEZVOUS_PORT, else MASTER_PORT + 1 (MASTER_PORT itself may be held by the launcher), else 29517.
struct ShardInfo {
  int rank = 0, world = 1, local_rank = 0;
  std::string master_addr = "127.0.0.1";
  int port = 29517;
};
ShardInfo ShardFromEnvironment();

// The communicator of a sharded job.  Construction binds the process to GPU `local_rank` (hipSetDevice — the C ABI
// of include/ilqg.h works on the calling thread's current device, one process per GPU) and builds the RCCL communicator:
// rank 0 creates the ncclUniqueId and hands it to the other ranks over a TCP connection to master_addr:port.
class ShardContext {
 public:
  explicit ShardContext(const ShardInfo& info);
  ~ShardContext();
  ShardContext(const ShardContext&) = delete;
  ShardContext& operator=(const ShardContext&) = delete;
  const ShardInfo& Info() const { return info_; }
  // all-gather of `bytes_per_rank` bytes per rank, device buffers (RCCL over xGMI; world == 1: a device copy)
  void AllGather(const void* send, void* recv, size_t bytes_per_rank) const;

 private:
  ShardInfo info_;
  void* comm_ = nullptr;    // ncclComm_t
  void* stream_ = nullptr;  // hipStream_t the collectives run on
};

// The rendezvous alone (no GPU, no RCCL): rank 0 sends `bytes` of `token` to every other rank, which receives them
// into `token`.  Used by ShardContext for the ncclUniqueId; exported for the CPU test of the bootstrap.
void RendezvousBroadcast(const ShardInfo& info, void* token, size_t bytes);

}  // namespace host

// ---------------------------------------------------------------------------------------------
// Solvers
// ---------------------------------------------------------------------------------------------

// include/ilqgames/solver/lq_solver.h:58-86
class LQSolver {
 public:
  virtual ~LQSolver() {}
  virtual std::vector<Strategy> Solve(const std::vector<LinearDynamicsApproximation>& linearization,
                                      const std::vector<std::vector<QuadraticCostApproximation>>& quadraticization,
                                      const VectorXf& x0, std::vector<VectorXf>* delta_xs = nullptr,
                                      std::vector<std::vector<VectorXf>>* costates = nullptr) = 0;

 protected:
  LQSolver(const std::shared_ptr<const MultiPlayerIntegrableSystem>& dynamics, size_t num_time_steps)
      : dynamics_(dynamics), num_time_steps_(num_time_steps) {
    CHECK_NOTNULL(dynamics.get());
  }
  std::vector<Strategy> SolveOnDevice(bool open_loop, const std::vector<LinearDynamicsApproximation>& linearization,
                                      const std::vector<std::vector<QuadraticCostApproximation>>& quadraticization,
                                      const VectorXf& x0, std::vector<VectorXf>* delta_xs,
                                      std::vector<std::vector<VectorXf>>* costates);
  const std::shared_ptr<const MultiPlayerIntegrableSystem> dynamics_;
  const size_t num_time_steps_;
};

// include/ilqgames/solver/lq_feedback_solver.h:62-111 -> ilqg_lq_feedback_batch
class LQFeedbackSolver : public LQSolver {
 public:
  LQFeedbackSolver(const std::shared_ptr<const MultiPlayerIntegrableSystem>& dynamics, size_t num_time_steps)
      : LQSolver(dynamics, num_time_steps) {}
  std::vector<Strategy> Solve(const std::vector<LinearDynamicsApproximation>& linearization,
                              const std::vector<std::vector<QuadraticCostApproximation>>& quadraticization,
                              const VectorXf& x0, std::vector<VectorXf>* delta_xs = nullptr,
                              std::vector<std::vector<VectorXf>>* costates = nullptr) override {
    return SolveOnDevice(false, linearization, quadraticization, x0, delta_xs, costates);
  }
};

// include/ilqgames/solver/lq_open_loop_solver.h:64-112 -> ilqg_lq_openloop_batch
class LQOpenLoopSolver : public LQSolver {
 public:
  LQOpenLoopSolver(const std::shared_ptr<const MultiPlayerIntegrableSystem>& dynamics, size_t num_time_steps)
      : LQSolver(dynamics, num_time_steps) {}
  std::vector<Strategy> Solve(const std::vector<LinearDynamicsApproximation>& linearization,
                              const std::vector<std::vector<QuadraticCostApproximation>>& quadraticization,
                              const VectorXf& x0, std::vector<VectorXf>* delta_xs = nullptr,
                              std::vector<std::vector<VectorXf>>* costates = nullptr) override {
    return SolveOnDevice(true, linearization, quadraticization, x0, delta_xs, costates);
  }
};

// include/ilqgames/solver/game_solver.h:69-112
class GameSolver {
 public:
  virtual ~GameSolver();
  virtual std::shared_ptr<SolverLog> Solve(bool* success = nullptr, Time max_runtime = constants::kInfinity) = 0;
  Problem& GetProblem() { return *problem_; }

  // Batched form of Solve(): one instance per entry of x0s, all sharing the Problem definition and
  // its current operating point / strategies as warm start.  This is the call that fills the GPU.
  host::BatchResult SolveBatch(const std::vector<VectorXf>& x0s);
  // The same over several GPUs: every rank passes the SAME global list; rank r solves its block
  // (host::InstanceRange) on its GPU, the per-instance results (operating points, strategies, costs, flags) are
  // all-gathered with RCCL, and every rank returns the BatchResult of the whole batch, in the order of x0s.
  host::BatchResult SolveBatchSharded(const std::vector<VectorXf>& x0s, const host::ShardContext& shard);
  const SolverParams& Params() const { return params_; }
  bool IsAugmentedLagrangian() const { return augmented_lagrangian_; }

 protected:
  GameSolver(const std::shared_ptr<Problem>& problem, const SolverParams& params, bool augmented_lagrangian);
  virtual std::shared_ptr<SolverLog> CreateNewLog() const { return std::make_shared<SolverLog>(); }
  host::BatchResult SolveOne(Time max_runtime = std::numeric_limits<Time>::infinity());  // Solve(): one instance; a repeated call continues this solver object's state
  void RefreshDevice();          // (re)builds the device tables when the Problem / SolverParams no longer match them
  const std::shared_ptr<Problem> problem_;
  const SolverParams params_;

 private:
  const bool augmented_lagrangian_;
  std::unique_ptr<host::DeviceSolve> device_;
};

// include/ilqgames/solver/ilq_solver.h:66-190 -> ilqg_ilq_solve_batch
class ILQSolver : public GameSolver {
 public:
  ILQSolver(const std::shared_ptr<Problem>& problem, const SolverParams& params = SolverParams())
      : GameSolver(problem, params, false) {}
  std::shared_ptr<SolverLog> Solve(bool* success = nullptr,
                                   Time max_runtime = std::numeric_limits<Time>::infinity()) override;
};

// include/ilqgames/solver/augmented_lagrangian_solver.h:69-92 -> ilqg_al_solve_batch
class AugmentedLagrangianSolver : public GameSolver {
 public:
  AugmentedLagrangianSolver(const std::shared_ptr<Problem>& problem, const SolverParams& params)
      : GameSolver(problem, params, true) {}
  std::shared_ptr<SolverLog> Solve(bool* success = nullptr, Time max_runtime = 5.0) override;
};

// include/ilqgames/solver/solution_splicer.h:57-86
class SolutionSplicer {
 public:
  ~SolutionSplicer() {}
  explicit SolutionSplicer(const SolverLog& log);
  // Splice in a new solution: up to five steps of the stored plan that precede the new solution's start are
  // kept in front of it (src/solution_splicer.cpp:60-129).
  void Splice(const SolverLog& log);
  bool ContainsTime(Time t) const {
    return (operating_point_.t0 <= t) && (operating_point_.t0 + operating_point_.xs.size() * time::kTimeStep >= t);
  }
  const std::vector<Strategy>& CurrentStrategies() const { return strategies_; }
  const OperatingPoint& CurrentOperatingPoint() const { return operating_point_; }

 private:
  std::vector<Strategy> strategies_;
  OperatingPoint operating_point_;
};

// include/ilqgames/examples/receding_horizon_simulator.h:58-60.  The solver is called repeatedly on problems
// re-anchored at the simulated state; between calls the state follows the spliced plan for 0.25 s plus the
// solve time (wall clock, or host::Options().simulated_solve_time when that is >= 0).
std::vector<std::shared_ptr<const SolverLog>> RecedingHorizonSimulator(Time final_time, Time planner_runtime,
                                                                       GameSolver* solver);

// include/ilqgames/examples/minimally_invasive_receding_horizon_simulator.h:60-72: two planners (the original one
// and a safety one, same dynamics type and initial condition) are re-solved at every replanning instant; the
// returned vector says whose plan was spliced in after each pair of calls.
enum ActiveProblem { ORIGINAL, SAFETY };
std::vector<ActiveProblem> MinimallyInvasiveRecedingHorizonSimulator(
    Time final_time, Time planner_runtime, GameSolver* original, GameSolver* safety,
    std::vector<std::shared_ptr<const SolverLog>>* original_logs,
    std::vector<std::shared_ptr<const SolverLog>>* safety_logs);

namespace host {
// RecedingHorizonSimulator for a batch of initial states: every instance shares the simulator's clock (0.25 s of
// motion before each call, `simulated_solve_time` after it) and keeps its own spliced plan; plans, states and the
// solver's workspace stay on the device between calls (ilqg_plan_integrate_batch, ilqg_receding_horizon_sync_batch,
// ilqg_solve_again_batch, ilqg_solution_splice_batch).  An instance leaves the loop where the reference would
// (ContainsTime false, or a failed first solve); logs[b] holds one log per solver call instance b took part in.
std::vector<std::vector<std::shared_ptr<const SolverLog>>> RecedingHorizonSimulatorBatch(
    Time final_time, Time planner_runtime, GameSolver* solver, const std::vector<VectorXf>& x0s,
    Time simulated_solve_time = 0.25);
}  // namespace host

// include/ilqgames/utils/compute_strategy_costs.h:58-60, include/ilqgames/utils/check_local_nash_equilibrium.h:61-70:
// the Problem overloads (the problem's current operating point, strategies and initial state).  Both run on the
// device: ilqg_strategy_costs_batch (default RK4 integration) and ilqg_check_local_nash_batch (one-step Euler, every
// alpha entry moved down and up by max_perturbation — 2 m (T-1) rollouts in one launch).
std::vector<float> ComputeStrategyCosts(const Problem& problem, bool open_loop = false);
bool NumericalCheckLocalNashEquilibrium(const Problem& problem, float max_perturbation, bool open_loop = false);
// check_local_nash_equilibrium.h:76-77: Q_i, R_ij of every player's quadraticisation along the problem's current
// operating point positive semidefinite to within 1e-4 (ilqg_check_sufficient_nash_batch).
bool CheckSufficientLocalNashEquilibrium(const Problem& problem);

// include/ilqgames/geometry/draw_shapes.h:52-53 (src/draw_shapes.cpp:61-73): a circle as a closed polyline of
// num_segments chords, starting at angle 0 and running counter-clockwise.
Polyline2 DrawCircle(const Point2& center, float radius, size_t num_segments);

// include/ilqgames/examples/roundabout_lane_center.h:55-57
PointList2 RoundaboutLaneCenter(float entrance_angle, float exit_angle, float distance_from_roundabout);

}  // namespace ilqgames

#endif  // ILQGAMES_HOST_API_HPP_
