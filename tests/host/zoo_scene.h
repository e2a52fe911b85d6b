// Test scenes written against the mirrored reference API (include/ilqgames/host/api.hpp): the same games as
// ilqgames_amd/examples.py::cost_zoo_scene / dynamics_zoo_scene / delayed_dubins_scene, term for term and in the same order, so that tests/test_host_mirror.py can
// check that the C++ classes of the cost / constraint kinds no reference example uses flatten to the descriptor the
// Python builders produce.  Not a reference example.
#pragma once
#include <ilqgames/host/api.hpp>

#include <cmath>
#include <memory>
#include <utility>
#include <vector>

namespace ilqgames {

class CostZooScene : public TopDownRenderableProblem {
 public:
  using Car = SinglePlayerCar5D;
  void ConstructDynamics() override {
    dynamics_.reset(new ConcatenatedDynamicalSystem({std::make_shared<Car>(4.0f), std::make_shared<Car>(4.0f)}));
  }
  void ConstructInitialState() override {
    x0_ = VectorXf::Zero(dynamics_->XDim());
    x0_(Car::kPyIdx) = -30.0f;
    x0_(Car::kThetaIdx) = static_cast<float>(M_PI_2);
    x0_(Car::kVIdx) = 4.0f;
    x0_(5 + Car::kPxIdx) = -5.0f;
    x0_(5 + Car::kPyIdx) = 30.0f;
    x0_(5 + Car::kThetaIdx) = static_cast<float>(-M_PI_2);
    x0_(5 + Car::kVIdx) = 3.0f;
  }
  void ConstructPlayerCosts() override {
    player_costs_.emplace_back("car1");
    player_costs_.emplace_back("car2");
    const auto xy = [](PlayerIndex ii) { return std::make_pair(Dimension(5 * ii + Car::kPxIdx), Dimension(5 * ii + Car::kPyIdx)); };
    const std::pair<Dimension, Dimension> controls(Car::kOmegaIdx, Car::kAIdx);
    for (PlayerIndex ii = 0; ii < 2; ii++) {
      PlayerCost& cost = player_costs_[ii];
      cost.AddControlCost(ii, std::make_shared<QuadraticCost>(25.0f, Car::kOmegaIdx, 0.0f, "steer"));
      cost.AddControlCost(ii, std::make_shared<QuadraticCost>(15.0f, Car::kAIdx, 0.0f, "accelerate"));
      cost.AddControlCost(ii, std::make_shared<SemiquadraticNormCost>(40.0f, controls, 1.5f, true, "effort"));
    }
    player_costs_[0].AddStateCost(std::make_shared<QuadraticCost>(10.0f, Car::kVIdx, 8.0f, "cruise"));
    player_costs_[1].AddStateCost(std::make_shared<QuadraticCost>(10.0f, 5 + Car::kVIdx, 8.0f, "cruise"));
    const Polyline2 lane1({Point2(0.0, -1000.0), Point2(0.0, 1000.0)});
    const Polyline2 lane2({Point2(-5.0, 1000.0), Point2(-5.0, 5.0), Point2(0.0, 0.0), Point2(995.0, 0.0)});
    const Polyline2 wall({Point2(3.0, -1000.0), Point2(3.0, -10.0), Point2(0.2, 0.0), Point2(3.0, 10.0), Point2(3.0, 1000.0)});
    player_costs_[0].AddStateCost(std::make_shared<QuadraticPolyline2Cost>(25.0f, lane1, xy(0), "lane"));
    player_costs_[1].AddStateCost(std::make_shared<QuadraticPolyline2Cost>(25.0f, lane2, xy(1), "lane"));
    player_costs_[0].AddStateCost(std::make_shared<OrientationCost>(5.0f, Car::kThetaIdx, static_cast<float>(M_PI_2), "heading"));
    player_costs_[1].AddStateCost(std::make_shared<OrientationCost>(2.0f, 5 + Car::kThetaIdx, -1.0f, "heading"));
    player_costs_[0].AddStateCost(std::make_shared<QuadraticNormCost>(0.5f, xy(0), 40.0f, "ring"));
    player_costs_[1].AddStateCost(std::make_shared<SemiquadraticNormCost>(0.5f, xy(1), 45.0f, true, "inside"));
    player_costs_[1].AddStateCost(std::make_shared<SemiquadraticNormCost>(0.5f, xy(1), 2.0f, false, "not at the origin"));
    player_costs_[1].AddStateCost(std::make_shared<RelativeDistanceCost>(1.0f, xy(1), xy(0), "follow"));
    for (PlayerIndex ii = 0; ii < 2; ii++) {
      player_costs_[ii].AddStateCost(std::make_shared<LocallyConvexProximityCost>(50.0f, xy(ii), xy(1 - ii), 6.0f, "gap"));
      player_costs_[ii].AddStateCost(std::make_shared<CurvatureCost>(20.0f, 5 * ii + Car::kPhiIdx, 5 * ii + Car::kVIdx, "curvature"));
    }
    player_costs_[0].AddStateConstraint(std::make_shared<Polyline2SignedDistanceConstraint>(wall, xy(0), -0.5f, true, "wall"));
    player_costs_[1].AddStateConstraint(std::make_shared<FinalTimeConstraint>(
        std::make_shared<Polyline2SignedDistanceConstraint>(lane1, xy(1), -4.0f, false, "far side"), 4.0f));
  }
  std::vector<float> Xs(const VectorXf& x) const override { return {x(0), x(5)}; }
  std::vector<float> Ys(const VectorXf& x) const override { return {x(1), x(6)}; }
  std::vector<float> Thetas(const VectorXf& x) const override { return {x(2), x(7)}; }
};

// ilqgames_amd/examples.py::affine_constraint_scene — two Car5D with the reference's two dense constraints
// (AffineScalarConstraint as an inequality and as an equality on the state, AffineVectorConstraint on a control vector).
class AffineConstraintScene : public TopDownRenderableProblem {
 public:
  using Car = SinglePlayerCar5D;
  void ConstructDynamics() override {
    dynamics_.reset(new ConcatenatedDynamicalSystem({std::make_shared<Car>(4.0f), std::make_shared<Car>(4.0f)}));
  }
  void ConstructInitialState() override {
    x0_ = VectorXf::Zero(dynamics_->XDim());
    x0_(Car::kPyIdx) = -25.0f;
    x0_(Car::kThetaIdx) = static_cast<float>(M_PI_2);
    x0_(Car::kVIdx) = 5.0f;
    x0_(5 + Car::kPxIdx) = -30.0f;
    x0_(5 + Car::kPyIdx) = 2.0f;
    x0_(5 + Car::kVIdx) = 5.0f;
  }
  void ConstructPlayerCosts() override {
    player_costs_.emplace_back("car1", 10.0f, 10.0f);
    player_costs_.emplace_back("car2", 10.0f, 10.0f);
    const auto xy = [](PlayerIndex ii) { return std::make_pair(Dimension(5 * ii + Car::kPxIdx), Dimension(5 * ii + Car::kPyIdx)); };
    const Polyline2 lane1({Point2(0.0, -1000.0), Point2(0.0, 1000.0)});
    const Polyline2 lane2({Point2(-1000.0, 2.0), Point2(1000.0, 2.0)});
    for (PlayerIndex ii = 0; ii < 2; ii++) {
      PlayerCost& cost = player_costs_[ii];
      cost.AddControlCost(ii, std::make_shared<QuadraticCost>(10.0f, Car::kOmegaIdx, 0.0f, "steer"));
      cost.AddControlCost(ii, std::make_shared<QuadraticCost>(5.0f, Car::kAIdx, 0.0f, "accelerate"));
      cost.AddStateCost(std::make_shared<QuadraticCost>(10.0f, 5 * ii + Car::kVIdx, 6.0f, "cruise"));
    }
    player_costs_[0].AddStateCost(std::make_shared<QuadraticPolyline2Cost>(25.0f, lane1, xy(0), "lane"));
    player_costs_[1].AddStateCost(std::make_shared<QuadraticPolyline2Cost>(25.0f, lane2, xy(1), "lane"));
    VectorXf a1 = VectorXf::Zero(10), a2 = VectorXf::Zero(10);
    a1(Car::kPxIdx) = 1.0f;
    a1(5 + Car::kPxIdx) = -0.25f;
    a2(5 + Car::kThetaIdx) = 1.0f;
    a2(5 + Car::kVIdx) = -0.01f;
    player_costs_[0].AddStateConstraint(std::make_shared<AffineScalarConstraint>(a1, 6.0f, false, "behind the line"));
    player_costs_[1].AddStateConstraint(std::make_shared<AffineScalarConstraint>(a2, -0.06f, true, "heading tied to speed"));
    MatrixXf A = MatrixXf::Zero(2, 2);
    A(0, 0) = 1.0f; A(0, 1) = 0.3f; A(1, 0) = -0.2f; A(1, 1) = 1.0f;
    VectorXf b = VectorXf::Zero(2);
    b(0) = 2.0f; b(1) = 3.0f;
    player_costs_[0].AddControlConstraint(0, std::make_shared<AffineVectorConstraint>(A, b, false, "controls drawn to A u = b"));
  }
  std::vector<float> Xs(const VectorXf& x) const override { return {x(0), x(5)}; }
  std::vector<float> Ys(const VectorXf& x) const override { return {x(1), x(6)}; }
  std::vector<float> Thetas(const VectorXf& x) const override { return {x(2), x(7)}; }
};

// ilqgames_amd/examples.py::weighted_proximity_scene — the skeleton example with WeightedConvexProximityCost.
class WeightedProximityScene : public TopDownRenderableProblem {
 public:
  using Car = SinglePlayerCar5D;
  void ConstructDynamics() override {
    dynamics_.reset(new ConcatenatedDynamicalSystem({std::make_shared<Car>(4.0f), std::make_shared<Car>(4.0f)}));
  }
  void ConstructInitialState() override {
    x0_ = VectorXf::Zero(dynamics_->XDim());
    x0_(Car::kPyIdx) = -30.0f;
    x0_(Car::kThetaIdx) = static_cast<float>(M_PI_2);
    x0_(Car::kVIdx) = 4.0f;
    x0_(5 + Car::kPxIdx) = -5.0f;
    x0_(5 + Car::kPyIdx) = 30.0f;
    x0_(5 + Car::kThetaIdx) = static_cast<float>(-M_PI_2);
    x0_(5 + Car::kVIdx) = 3.0f;
  }
  void ConstructPlayerCosts() override {
    player_costs_.emplace_back("car1");
    player_costs_.emplace_back("car2");
    const auto xy = [](PlayerIndex ii) { return std::make_pair(Dimension(5 * ii + Car::kPxIdx), Dimension(5 * ii + Car::kPyIdx)); };
    for (PlayerIndex ii = 0; ii < 2; ii++) {
      player_costs_[ii].AddControlCost(ii, std::make_shared<QuadraticCost>(25.0f, Car::kOmegaIdx, 0.0f, "steer"));
      player_costs_[ii].AddControlCost(ii, std::make_shared<QuadraticCost>(15.0f, Car::kAIdx, 0.0f, "accelerate"));
    }
    player_costs_[0].AddStateCost(std::make_shared<QuadraticCost>(10.0f, Car::kVIdx, 8.0f, "cruise"));
    player_costs_[1].AddStateCost(std::make_shared<QuadraticCost>(10.0f, 5 + Car::kVIdx, 8.0f, "cruise"));
    const Polyline2 lane1({Point2(0.0, -1000.0), Point2(0.0, 1000.0)});
    const Polyline2 lane2({Point2(-5.0, 1000.0), Point2(-5.0, 5.0), Point2(0.0, 0.0), Point2(995.0, 0.0)});
    player_costs_[0].AddStateCost(std::make_shared<QuadraticPolyline2Cost>(25.0f, lane1, xy(0), "lane"));
    player_costs_[1].AddStateCost(std::make_shared<QuadraticPolyline2Cost>(25.0f, lane2, xy(1), "lane"));
    for (PlayerIndex ii = 0; ii < 2; ii++)
      player_costs_[ii].AddStateCost(std::make_shared<WeightedConvexProximityCost>(
          0.02f, xy(ii), xy(1 - ii), Dimension(5 * ii + Car::kVIdx), Dimension(5 * (1 - ii) + Car::kVIdx), 40.0f, "gap"));
  }
  std::vector<float> Xs(const VectorXf& x) const override { return {x(0), x(5)}; }
  std::vector<float> Ys(const VectorXf& x) const override { return {x(1), x(6)}; }
  std::vector<float> Thetas(const VectorXf& x) const override { return {x(2), x(7)}; }
};

// ilqgames_amd/examples.py::dynamics_zoo_scene — SinglePlayerCar7D and two SinglePlayerUnicycle5D.
class DynamicsZooScene : public TopDownRenderableProblem {
 public:
  using Car = SinglePlayerCar7D;
  using Uni = SinglePlayerUnicycle5D;
  void ConstructDynamics() override {
    dynamics_.reset(new ConcatenatedDynamicalSystem(
        {std::make_shared<Car>(4.0f), std::make_shared<Uni>(), std::make_shared<Uni>()}));
  }
  void ConstructInitialState() override {
    x0_ = VectorXf::Zero(dynamics_->XDim());
    x0_(Car::kPyIdx) = -30.0f;
    x0_(Car::kThetaIdx) = static_cast<float>(M_PI_2);
    x0_(Car::kVIdx) = 4.0f;
    x0_(7 + Uni::kPxIdx) = -5.0f;
    x0_(7 + Uni::kPyIdx) = 30.0f;
    x0_(7 + Uni::kThetaIdx) = static_cast<float>(-M_PI_2);
    x0_(7 + Uni::kVIdx) = 3.0f;
    x0_(12 + Uni::kPxIdx) = -25.0f;
    x0_(12 + Uni::kPyIdx) = 8.0f;
    x0_(12 + Uni::kVIdx) = 5.0f;
  }
  void ConstructPlayerCosts() override {
    for (const char* name : {"car", "unicycle1", "unicycle2"}) player_costs_.emplace_back(name);
    const Dimension base[3] = {0, 7, 12};
    const Dimension v[3] = {Car::kVIdx, 7 + Uni::kVIdx, 12 + Uni::kVIdx};
    const Dimension s[3] = {Car::kSIdx, 7 + Uni::kSIdx, 12 + Uni::kSIdx};
    const auto xy = [&](PlayerIndex ii) { return std::make_pair(Dimension(base[ii]), Dimension(base[ii] + 1)); };
    for (PlayerIndex ii = 0; ii < 3; ii++) {
      PlayerCost& cost = player_costs_[ii];
      cost.AddControlCost(ii, std::make_shared<QuadraticCost>(25.0f, 0, 0.0f, "steer"));
      cost.AddControlCost(ii, std::make_shared<QuadraticCost>(15.0f, 1, 0.0f, "accelerate"));
      cost.AddStateCost(std::make_shared<QuadraticCost>(10.0f, v[ii], 6.0f, "cruise"));
      cost.AddStateCost(std::make_shared<QuadraticCost>(0.02f, s[ii], 50.0f, "path length"));
    }
    player_costs_[0].AddStateCost(std::make_shared<QuadraticCost>(30.0f, Car::kKappaIdx, 0.0f, "curvature"));
    player_costs_[1].AddStateCost(std::make_shared<NominalPathLengthCost>(0.5f, s[1], 5.0f, "on schedule"));
    const Polyline2 lanes[3] = {
        Polyline2({Point2(0.0, -1000.0), Point2(0.0, 1000.0)}),
        Polyline2({Point2(-5.0, 1000.0), Point2(-5.0, 5.0), Point2(0.0, 0.0), Point2(995.0, 0.0)}),
        Polyline2({Point2(-1000.0, 8.0), Point2(1000.0, 8.0)})};
    for (PlayerIndex ii = 0; ii < 3; ii++)
      player_costs_[ii].AddStateCost(std::make_shared<QuadraticPolyline2Cost>(25.0f, lanes[ii], xy(ii), "lane"));
    player_costs_[0].AddStateCost(std::make_shared<RouteProgressCost>(2.0f, 6.0f, lanes[0], xy(0), "progress", 970.0f));
    player_costs_[1].AddStateCost(std::make_shared<RouteProgressCost>(1.0f, 4.0f, lanes[1], xy(1), "progress", 968.0f));
    for (PlayerIndex ii = 0; ii < 3; ii++)
      for (PlayerIndex jj = 0; jj < 3; jj++)
        if (ii != jj) player_costs_[ii].AddStateCost(std::make_shared<ProximityCost>(100.0f, xy(ii), xy(jj), 6.0f, "gap"));
  }
  std::vector<float> Xs(const VectorXf& x) const override { return {x(0), x(7), x(12)}; }
  std::vector<float> Ys(const VectorXf& x) const override { return {x(1), x(8), x(13)}; }
  std::vector<float> Thetas(const VectorXf& x) const override { return {x(2), x(9), x(14)}; }
};

// ilqgames_amd/examples.py::delayed_dubins_scene — two SinglePlayerDelayedDubinsCar.
class DelayedDubinsScene : public TopDownRenderableProblem {
 public:
  using Car = SinglePlayerDelayedDubinsCar;
  void ConstructDynamics() override {
    dynamics_.reset(new ConcatenatedDynamicalSystem({std::make_shared<Car>(1.0f), std::make_shared<Car>(1.0f)}));
  }
  void ConstructInitialState() override {
    x0_ = VectorXf::Zero(dynamics_->XDim());
    const float x0[8] = {2.0f, 1.0f, static_cast<float>(M_PI_2), 0.0f, -1.0f, -2.0f, 0.3f, 0.1f};
    for (int i = 0; i < 8; i++) x0_(i) = x0[i];
  }
  void ConstructPlayerCosts() override {
    player_costs_.emplace_back("p1");
    player_costs_.emplace_back("p2");
    for (PlayerIndex ii = 0; ii < 2; ii++) {
      player_costs_[ii].AddControlCost(ii, std::make_shared<QuadraticCost>(1.0f, Car::kAlphaIdx, 0.0f, "turn"));
      player_costs_[ii].AddStateCost(std::make_shared<QuadraticCost>(2.0f, 4 * ii + Car::kOmegaIdx, 0.0f, "turn rate"));
    }
    player_costs_[0].AddStateCost(std::make_shared<QuadraticCost>(1.0f, Car::kPxIdx, 0.0f, "x"));
    player_costs_[0].AddStateCost(std::make_shared<QuadraticCost>(1.0f, Car::kPyIdx, 0.0f, "y"));
    player_costs_[1].AddStateCost(std::make_shared<QuadraticDifferenceCost>(
        1.0f, std::vector<Dimension>{4 + Car::kPxIdx, 4 + Car::kPyIdx}, std::vector<Dimension>{Car::kPxIdx, Car::kPyIdx}, "chase"));
  }
  std::vector<float> Xs(const VectorXf& x) const override { return {x(0), x(4)}; }
  std::vector<float> Ys(const VectorXf& x) const override { return {x(1), x(5)}; }
  std::vector<float> Thetas(const VectorXf& x) const override { return {x(2), x(6)}; }
};

}  // namespace ilqgames
