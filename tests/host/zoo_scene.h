// Test scene written against the mirrored reference API (include/ilqgames/host/api.hpp): the same game as
// ilqgames_amd/examples.py::cost_zoo_scene, term for term and in the same order, so that tests/test_host_mirror.py can
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
    player_costs_[1].AddStateConstraint(std::make_shared<Polyline2SignedDistanceConstraint>(lane1, xy(1), -60.0f, false, "far side"));
  }
  std::vector<float> Xs(const VectorXf& x) const override { return {x(0), x(5)}; }
  std::vector<float> Ys(const VectorXf& x) const override { return {x(1), x(6)}; }
  std::vector<float> Thetas(const VectorXf& x) const override { return {x(2), x(7)}; }
};

}  // namespace ilqgames
