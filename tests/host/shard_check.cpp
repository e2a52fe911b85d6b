// Driver for the host-side multi-GPU entry (include/ilqgames/host/api.hpp: InstanceRange, ShardFromEnvironment,
// RendezvousBroadcast, ShardContext, GameSolver::SolveBatchSharded).
//   shard_check range            prints the cut of several (total, world) pairs — CPU, compared with
//                                ilqgames_amd/sharding.py by tests/test_host_mirror.py
//   shard_check rendezvous       every rank (RANK / WORLD_SIZE / ILQG_RENDEZVOUS_PORT from the environment) takes part in
//                                one RendezvousBroadcast of a token and prints its block of a 10-instance batch — CPU,
//                                run as two processes by the test
//   shard_check solve            one GPU: GameSolver::SolveBatchSharded on a world of one (and whatever world the
//                                environment names) must return what SolveBatch returns, instance by instance
#include <ilqgames/cost/proximity_cost.h>
#include <ilqgames/cost/quadratic_cost.h>
#include <ilqgames/dynamics/concatenated_dynamical_system.h>
#include <ilqgames/dynamics/single_player_car_5d.h>
#include <ilqgames/solver/ilq_solver.h>
#include <ilqgames/solver/top_down_renderable_problem.h>

#include <cstring>
#include <iostream>

using namespace ilqgames;

namespace {
// Two cars on crossing courses (n = 10): enough of a game for the sharded and the plain batch solve to be compared.
class CrossingCars : public TopDownRenderableProblem {
 public:
  using Car = SinglePlayerCar5D;
  void ConstructDynamics() override {
    dynamics_.reset(new ConcatenatedDynamicalSystem({std::make_shared<Car>(4.0f), std::make_shared<Car>(4.0f)}));
  }
  void ConstructInitialState() override {
    x0_ = VectorXf::Zero(dynamics_->XDim());
    x0_(Car::kPxIdx) = -20.0f;
    x0_(Car::kVIdx) = 5.0f;
    x0_(5 + Car::kPyIdx) = -20.0f;
    x0_(5 + Car::kThetaIdx) = static_cast<float>(M_PI_2);
    x0_(5 + Car::kVIdx) = 5.0f;
  }
  void ConstructPlayerCosts() override {
    player_costs_.emplace_back("car1", 1.0f, 1.0f);
    player_costs_.emplace_back("car2", 1.0f, 1.0f);
    for (PlayerIndex ii = 0; ii < 2; ii++) {
      const Dimension base = 5 * ii;
      PlayerCost& cost = player_costs_[ii];
      cost.AddStateCost(std::make_shared<QuadraticCost>(4.0f, base + Car::kVIdx, 6.0f, "cruise"));
      cost.AddStateCost(std::make_shared<QuadraticCost>(2.0f, base + (ii == 0 ? Car::kPyIdx : Car::kPxIdx), 0.0f, "lane"));
      cost.AddStateCost(std::make_shared<ProximityCost>(20.0f, std::make_pair(base, base + 1),
                                                        std::make_pair(Dimension(5 - base), Dimension(6 - base)), 6.0f, "gap"));
      cost.AddControlCost(ii, std::make_shared<QuadraticCost>(0.5f, 0, 0.0f, "steer"));
      cost.AddControlCost(ii, std::make_shared<QuadraticCost>(0.2f, 1, 0.0f, "accelerate"));
    }
  }
  std::vector<float> Xs(const VectorXf& x) const override { return {x(0), x(5)}; }
  std::vector<float> Ys(const VectorXf& x) const override { return {x(1), x(6)}; }
  std::vector<float> Thetas(const VectorXf& x) const override { return {x(2), x(7)}; }
};
}  // namespace

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "range";
  if (mode == "range") {
    const size_t totals[] = {1, 7, 10, 1024, 65536, 65537};
    const int worlds[] = {1, 2, 3, 8};
    for (size_t total : totals)
      for (int world : worlds)
        for (int r = 0; r < world; r++) {
          size_t lo, hi;
          host::InstanceRange(total, r, world, &lo, &hi);
          std::cout << total << " " << world << " " << r << " " << lo << " " << hi << "\n";
        }
    return 0;
  }
  if (mode == "rendezvous") {
    const host::ShardInfo info = host::ShardFromEnvironment();
    char token[128];
    std::memset(token, 0, sizeof(token));
    if (info.rank == 0) std::strcpy(token, "ncclUniqueId-stand-in:0123456789abcdef");
    host::RendezvousBroadcast(info, token, sizeof(token));
    size_t lo, hi;
    host::InstanceRange(10, info.rank, info.world, &lo, &hi);
    std::cout << "rank " << info.rank << " of " << info.world << " token " << token << " block " << lo << " " << hi << "\n";
    return 0;
  }
  if (mode == "solve") {
    const host::ShardInfo info = host::ShardFromEnvironment();
    host::ShardContext shard(info);
    SolverParams params;
    params.max_solver_iters = 6;
    params.initial_alpha_scaling = 0.1;
    params.expected_decrease_fraction = 0.001;
    auto problem = std::make_shared<CrossingCars>();
    problem->Initialize();
    ILQSolver solver(problem, params);
    std::vector<VectorXf> x0s;
    for (int b = 0; b < 7; b++) {
      VectorXf x = problem->InitialState();
      x(0) += 0.1f * b;
      x(1) -= 0.05f * b;
      x0s.push_back(x);
    }
    const host::BatchResult whole = solver.SolveBatch(x0s);
    const host::BatchResult sharded = solver.SolveBatchSharded(x0s, shard);
    CHECK_EQ(whole.logs.size(), sharded.logs.size());
    double worst = 0.0;
    for (size_t b = 0; b < x0s.size(); b++) {
      CHECK_EQ(whole.success[b], sharded.success[b]);
      const OperatingPoint &oa = whole.logs[b]->FinalOperatingPoint(), &ob = sharded.logs[b]->FinalOperatingPoint();
      for (size_t k = 0; k < oa.xs.size(); k++)
        for (int e = 0; e < oa.xs[k].size(); e++) worst = std::max(worst, double(std::abs(oa.xs[k](e) - ob.xs[k](e))));
      const std::vector<Strategy> &sa = whole.logs[b]->FinalStrategies(), &sb = sharded.logs[b]->FinalStrategies();
      for (size_t i = 0; i < sa.size(); i++)
        for (size_t k = 0; k < sa[i].Ps.size(); k++)
          for (int e = 0; e < sa[i].Ps[k].size(); e++)
            worst = std::max(worst, double(std::abs(sa[i].Ps[k].data()[e] - sb[i].Ps[k].data()[e])));
    }
    std::cout << "rank " << info.rank << " of " << info.world << ": " << x0s.size()
              << " instances, largest difference between SolveBatch and SolveBatchSharded " << worst << "\n";
    return worst == 0.0 ? 0 : 1;
  }
  std::cerr << "usage: shard_check range|rendezvous|solve\n";
  return 2;
}
