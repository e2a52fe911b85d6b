// GPU test driver for the host-side C++ mirror: defines a three-player merge scene with the
// reference's problem-definition API (Problem / PlayerCost / *Cost / *Constraint /
// ConcatenatedDynamicalSystem), solves it with ILQSolver, GameSolver::SolveBatch and
// AugmentedLagrangianSolver, and runs one LQ game through LQFeedbackSolver / LQOpenLoopSolver.
// Everything is written to <outdir>/ as text; tests/test_host_mirror.py replays the same inputs
// through the CPU oracle and compares.
#include <cstdlib>
#include <ilqgames/constraint/proximity_constraint.h>
#include <ilqgames/cost/proximity_cost.h>
#include <ilqgames/cost/quadratic_cost.h>
#include <ilqgames/cost/quadratic_polyline2_cost.h>
#include <ilqgames/cost/semiquadratic_cost.h>
#include <ilqgames/cost/semiquadratic_polyline2_cost.h>
#include <ilqgames/dynamics/concatenated_dynamical_system.h>
#include <ilqgames/dynamics/single_player_car_5d.h>
#include <ilqgames/dynamics/single_player_unicycle_4d.h>
#include <ilqgames/dynamics/two_player_unicycle_4d.h>
#include <ilqgames/solver/augmented_lagrangian_solver.h>
#include <ilqgames/solver/ilq_solver.h>
#include <ilqgames/solver/lq_feedback_solver.h>
#include <ilqgames/solver/lq_open_loop_solver.h>
#include <ilqgames/examples/minimally_invasive_receding_horizon_simulator.h>
#include <ilqgames/examples/receding_horizon_simulator.h>
#include <ilqgames/solver/solution_splicer.h>
#include <ilqgames/solver/top_down_renderable_problem.h>
#include <ilqgames/utils/check_local_nash_equilibrium.h>
#include <ilqgames/utils/compute_strategy_costs.h>

#include <fstream>
#include <iomanip>
#include <iostream>

namespace ilqgames {
namespace {

using Car = SinglePlayerCar5D;
using Walker = SinglePlayerUnicycle4D;
constexpr Dimension kCar1 = 0, kCar2 = 5, kWalker = 10;  // first state index of each player

// Two cars approach a junction at right angles while a slow unicycle crosses the first car's road.
class MergeScene : public TopDownRenderableProblem {
 public:
  explicit MergeScene(bool constrained, bool rewarded = false) : constrained_(constrained), rewarded_(rewarded) {}

  void ConstructDynamics() override {
    dynamics_.reset(new ConcatenatedDynamicalSystem(
        {std::make_shared<Car>(3.5f), std::make_shared<Car>(3.5f), std::make_shared<Walker>()}));
  }

  void ConstructInitialState() override {
    x0_ = VectorXf::Zero(dynamics_->XDim());
    x0_(kCar1 + Car::kPxIdx) = -25.0f;
    x0_(kCar1 + Car::kVIdx) = 6.0f;
    x0_(kCar2 + Car::kPxIdx) = 20.0f;
    x0_(kCar2 + Car::kPyIdx) = -30.0f;
    x0_(kCar2 + Car::kThetaIdx) = static_cast<float>(M_PI_2);
    x0_(kCar2 + Car::kVIdx) = 5.0f;
    x0_(kWalker + Walker::kPxIdx) = 5.0f;
    x0_(kWalker + Walker::kPyIdx) = 9.0f;
    x0_(kWalker + Walker::kThetaIdx) = static_cast<float>(-M_PI_2);
    x0_(kWalker + Walker::kVIdx) = 1.0f;
  }

  void ConstructPlayerCosts() override {
    const char* names[] = {"car1", "car2", "walker"};
    for (const char* name : names) player_costs_.emplace_back(name, 1.0f, 1.0f);

    const Polyline2 road1({Point2(-500.0f, 0.0f), Point2(10.0f, 0.0f), Point2(18.0f, 3.0f), Point2(500.0f, 3.0f)});
    const Polyline2 road2({Point2(20.0f, -500.0f), Point2(20.0f, 500.0f)});
    const Polyline2 crossing({Point2(5.0f, 500.0f), Point2(5.0f, -500.0f)});

    struct Lane {
      const Polyline2* line;
      Dimension base, vidx;
      float nominal_v, max_v;
    };
    const Lane lanes[] = {{&road1, kCar1, kCar1 + Car::kVIdx, 7.0f, 10.0f},
                          {&road2, kCar2, kCar2 + Car::kVIdx, 6.0f, 9.0f},
                          {&crossing, kWalker, kWalker + Walker::kVIdx, 1.2f, 2.0f}};
    for (PlayerIndex ii = 0; ii < 3; ii++) {
      const Lane& lane = lanes[ii];
      const std::pair<Dimension, Dimension> xy(lane.base, lane.base + 1);
      PlayerCost& cost = player_costs_[ii];
      cost.AddStateCost(std::make_shared<QuadraticPolyline2Cost>(20.0f, *lane.line, xy, "center"));
      cost.AddStateCost(std::make_shared<SemiquadraticPolyline2Cost>(80.0f, *lane.line, xy, 2.0f, true, "right"));
      cost.AddStateCost(std::make_shared<SemiquadraticPolyline2Cost>(80.0f, *lane.line, xy, -2.0f, false, "left"));
      cost.AddStateCost(std::make_shared<QuadraticCost>(8.0f, lane.vidx, lane.nominal_v, "cruise"));
      cost.AddStateCost(std::make_shared<SemiquadraticCost>(60.0f, lane.vidx, lane.max_v, true, "too fast"));
      cost.AddStateCost(std::make_shared<SemiquadraticCost>(60.0f, lane.vidx, 0.0f, false, "reversing"));
      cost.AddControlCost(ii, std::make_shared<QuadraticCost>(0.5f, 0, 0.0f, "steer"));
      cost.AddControlCost(ii, std::make_shared<QuadraticCost>(0.2f, 1, 0.0f, "accelerate"));
      for (PlayerIndex jj = 0; jj < 3; jj++) {
        if (jj == ii) continue;
        const std::pair<Dimension, Dimension> other(lanes[jj].base, lanes[jj].base + 1);
        cost.AddStateCost(std::make_shared<ProximityCost>(15.0f, xy, other, 5.0f, "gap"));
      }
    }
    if (rewarded_)  // car 1 is paid for car 2's distance from a far-away line: its total cost goes far below -1
      player_costs_[0].AddStateCost(std::make_shared<QuadraticCost>(-0.01f, kCar2 + Car::kPyIdx, 1000.0f, "reward"));
    if (constrained_) {
      // the two cars must stay at least 3 m apart (hard constraint, handled by the AL loop)
      const std::pair<Dimension, Dimension> a(kCar1, kCar1 + 1), b(kCar2, kCar2 + 1);
      player_costs_[0].AddStateConstraint(std::make_shared<ProximityConstraint>(a, b, 3.0f, false, "keep apart"));
      player_costs_[1].AddStateConstraint(std::make_shared<ProximityConstraint>(b, a, 3.0f, false, "keep apart"));
    }
  }

  std::vector<float> Xs(const VectorXf& x) const override { return {x(kCar1), x(kCar2), x(kWalker)}; }
  std::vector<float> Ys(const VectorXf& x) const override { return {x(kCar1 + 1), x(kCar2 + 1), x(kWalker + 1)}; }
  std::vector<float> Thetas(const VectorXf& x) const override {
    return {x(kCar1 + 2), x(kCar2 + 2), x(kWalker + 2)};
  }

 private:
  const bool constrained_;
  const bool rewarded_;
};

// One unicycle, two players (TwoPlayerUnicycle4D): the scene of ilqgames_amd/examples.py two_player_unicycle_4d_scene.
class PushedUnicycle : public TopDownRenderableProblem {
 public:
  void ConstructDynamics() override { dynamics_.reset(new TwoPlayerUnicycle4D()); }
  void ConstructInitialState() override {
    using Dyn = TwoPlayerUnicycle4D;
    x0_ = VectorXf::Zero(dynamics_->XDim());
    x0_(Dyn::kPyIdx) = -10.0f;
    x0_(Dyn::kThetaIdx) = static_cast<float>(M_PI / 4.0);
    x0_(Dyn::kVIdx) = 5.0f;
  }
  void ConstructPlayerCosts() override {
    using Dyn = TwoPlayerUnicycle4D;
    player_costs_.emplace_back("driver", 1.0f, 1.0f);
    player_costs_.emplace_back("pusher", 1.0f, 1.0f);
    auto& p1 = player_costs_[0];
    auto& p2 = player_costs_[1];
    p1.AddStateCost(std::make_shared<QuadraticCost>(1.0f, Dyn::kPxIdx, 0.0f, "px"));
    p1.AddStateCost(std::make_shared<QuadraticCost>(1.0f, Dyn::kPyIdx, 0.0f, "py"));
    p1.AddStateCost(std::make_shared<QuadraticCost>(2.0f, Dyn::kVIdx, 3.0f, "v"));
    p1.AddStateCost(std::make_shared<SemiquadraticCost>(50.0f, Dyn::kVIdx, 8.0f, true, "vmax"));
    p2.AddStateCost(std::make_shared<QuadraticCost>(0.5f, Dyn::kPxIdx, 6.0f, "px"));
    p2.AddStateCost(std::make_shared<QuadraticCost>(0.5f, Dyn::kPyIdx, -4.0f, "py"));
    p1.AddControlCost(0, std::make_shared<QuadraticCost>(0.1f, -1, 0.0f, "u1"));
    p2.AddControlCost(1, std::make_shared<QuadraticCost>(0.1f, -1, 0.0f, "u2"));
    p2.AddControlCost(1, std::make_shared<QuadraticCost>(5.0f, -1, 0.0f, "u2 bound"));
    p1.AddControlCost(1, std::make_shared<QuadraticCost>(0.05f, -1, 0.0f, "u2 seen by p1"));
  }
  std::vector<float> Xs(const VectorXf& x) const override { return {x(TwoPlayerUnicycle4D::kPxIdx)}; }
  std::vector<float> Ys(const VectorXf& x) const override { return {x(TwoPlayerUnicycle4D::kPyIdx)}; }
  std::vector<float> Thetas(const VectorXf& x) const override { return {x(TwoPlayerUnicycle4D::kThetaIdx)}; }
};

void WriteLog(std::ostream& os, const SolverLog& log, bool success) {
  os << std::setprecision(9);
  os << "success " << (success ? 1 : 0) << " converged " << (log.WasConverged() ? 1 : 0) << " iters "
     << log.DeviceIterations() << "\n";
  os << "costs";
  for (float c : log.TotalCosts()) os << " " << c;
  os << "\n";
  const OperatingPoint& op = log.FinalOperatingPoint();
  for (size_t k = 0; k < op.xs.size(); k++) {
    os << "x " << op.xs[k] << "\n";
    os << "u";
    for (const auto& u : op.us[k]) os << " " << u;
    os << "\n";
  }
  const auto& strategies = log.FinalStrategies();
  for (size_t k = 0; k < op.xs.size(); k++) {
    os << "alpha";
    for (const auto& s : strategies) os << " " << s.alphas[k];
    os << "\n";
  }
}

// Deterministic pseudo-random numbers in [-1, 1) for the LQ game (no <random>: the same stream
// must be reproducible by the Python side, which reads the matrices back from the dump anyway).
struct Lcg {
  uint32_t state;
  float next() {
    state = state * 1664525u + 1013904223u;
    return static_cast<float>(static_cast<int32_t>(state >> 8) % 20001 - 10000) * 1e-4f;
  }
};

// Two planar point masses (n = 4 shared state, two controls each) with random coupled quadratic costs.
void RunLqGame(const std::string& outdir, bool open_loop) {
  class TwoPointMasses : public MultiPlayerDynamicalSystem {
   public:
    TwoPointMasses() : MultiPlayerDynamicalSystem(4) {}
    Dimension UDim(PlayerIndex) const override { return 2; }
    PlayerIndex NumPlayers() const override { return 2; }
    std::vector<Dimension> PositionDimensions() const override { return {0, 1}; }
  };
  const std::shared_ptr<const MultiPlayerIntegrableSystem> dynamics = std::make_shared<TwoPointMasses>();
  const size_t T = 12;
  const int n = 4, N = 2, mi = 2;
  Lcg rng{open_loop ? 77u : 42u};
  std::vector<LinearDynamicsApproximation> lin;
  std::vector<std::vector<QuadraticCostApproximation>> quad;
  for (size_t k = 0; k < T; k++) {
    LinearDynamicsApproximation l(*dynamics);
    for (int c = 0; c < n; c++)
      for (int r = 0; r < n; r++) l.A(r, c) += 0.1f * rng.next();
    for (int i = 0; i < N; i++)
      for (int c = 0; c < mi; c++)
        for (int r = 0; r < n; r++) l.Bs[i](r, c) = 0.5f * rng.next();
    lin.push_back(l);
    std::vector<QuadraticCostApproximation> per_player;
    for (int i = 0; i < N; i++) {
      QuadraticCostApproximation q(n);
      MatrixXf G(n, n);
      for (int c = 0; c < n; c++)
        for (int r = 0; r < n; r++) G(r, c) = rng.next();
      q.state.hess = G.transpose() * G + 0.5f * MatrixXf::Identity(n, n);
      for (int r = 0; r < n; r++) q.state.grad(r) = rng.next();
      for (int j = 0; j < N; j++) {  // own block first is not required; keep natural order
        SingleCostApproximation& blk = q.Control(static_cast<PlayerIndex>(j), mi);
        MatrixXf H(mi, mi);
        for (int c = 0; c < mi; c++)
          for (int r = 0; r < mi; r++) H(r, c) = rng.next();
        blk.hess = H.transpose() * H + (j == i ? 1.0f : 0.1f) * MatrixXf::Identity(mi, mi);
        for (int r = 0; r < mi; r++) blk.grad(r) = rng.next();
      }
      per_player.push_back(q);
    }
    quad.push_back(per_player);
  }
  VectorXf x0(n);
  for (int r = 0; r < n; r++) x0(r) = rng.next();

  std::unique_ptr<LQSolver> solver;
  if (open_loop)
    solver.reset(new LQOpenLoopSolver(dynamics, T));
  else
    solver.reset(new LQFeedbackSolver(dynamics, T));
  std::vector<VectorXf> delta_xs;
  std::vector<std::vector<VectorXf>> costates;  // the reference CHECKs that the two come together
  const std::vector<Strategy> strategies = solver->Solve(lin, quad, x0, &delta_xs, &costates);

  std::ofstream os(outdir + (open_loop ? "/lq_openloop.txt" : "/lq_feedback.txt"));
  os << std::setprecision(9) << "dims " << n << " " << N << " " << mi << " " << T << "\nx0 " << x0 << "\n";
  for (size_t k = 0; k < T; k++) {
    os << "A";
    for (int i = 0; i < n * n; i++) os << " " << lin[k].A.data()[i];
    os << "\nB";
    for (int p = 0; p < N; p++)
      for (int i = 0; i < n * mi; i++) os << " " << lin[k].Bs[p].data()[i];
    os << "\n";
    for (int p = 0; p < N; p++) {
      os << "Q";
      for (int i = 0; i < n * n; i++) os << " " << quad[k][p].state.hess.data()[i];
      os << "\nl " << quad[k][p].state.grad << "\n";
      for (const auto& e : quad[k][p].control) {
        os << "R";
        for (int i = 0; i < mi * mi; i++) os << " " << e.second.hess.data()[i];
        os << "\nr " << e.second.grad << "\n";
      }
    }
    for (int p = 0; p < N; p++) {
      os << "P";
      for (int i = 0; i < mi * n; i++) os << " " << strategies[p].Ps[k].data()[i];
      os << "\nalpha " << strategies[p].alphas[k] << "\n";
    }
    os << "dx " << delta_xs[k] << "\n";
    for (int p = 0; p < N; p++) os << "costate " << costates[k][p] << "\n";
  }
}

}  // namespace
}  // namespace ilqgames

int main(int argc, char** argv) {
  using namespace ilqgames;
  if (argc < 2) {
    std::cerr << "usage: host_solve_demo <outdir>\n";
    return 2;
  }
  const std::string outdir = argv[1];

  SolverParams params;
  params.max_solver_iters = 12;
  params.convergence_tolerance = 1e-3f;
  params.initial_alpha_scaling = 0.25f;
  params.expected_decrease_fraction = 0.01f;

  // 1. unconstrained scene: single solve (the reference call) and a batch of perturbed starts
  {
    auto problem = std::make_shared<MergeScene>(false);
    problem->Initialize();
    host::ProblemDescription description;
    std::string why;
    CHECK(host::DescribeProblem(*problem, params, ILQG_F64, &description, &why)) << why;
    std::ofstream(outdir + "/scene.txt") << host::DumpDescription(description);

    ILQSolver solver(problem, params);
    bool success = false;
    const std::shared_ptr<SolverLog> log = solver.Solve(&success);
    std::ofstream os(outdir + "/ilq_single.txt");
    os << std::setprecision(9) << "x0 " << problem->InitialState() << "\n";
    WriteLog(os, *log, success);

    // SolverLog on disk, from logs the device produced: every iterate of a solve (src/solver_log.cpp:113-171) and the
    // SaveLogs form (last trajectory only, :208-240).  The solve above gives up in its second iteration's line search
    // (as the reference's does with these parameters): its log holds iterates 0 and 1.  A second solve with the
    // intersection main's line-search parameters runs its six iterations.
    setenv("ILQGAMES_LOG_DIR", outdir.c_str(), 1);
    CHECK(log->Save(false, "ilq_single_log"));
    std::ofstream(outdir + "/ilq_single_log_meta.txt") << "iterates " << log->NumIterates() << " device_iterations "
                                                        << log->DeviceIterations() << "\n";
    {
      SolverParams log_params = params;
      log_params.max_solver_iters = 6;
      log_params.initial_alpha_scaling = 0.1f;
      log_params.expected_decrease_fraction = 0.001f;
      auto logged_problem = std::make_shared<MergeScene>(false);
      logged_problem->Initialize();
      host::ProblemDescription logged_description;
      CHECK(host::DescribeProblem(*logged_problem, log_params, ILQG_F64, &logged_description, &why)) << why;
      std::ofstream(outdir + "/scene_logged.txt") << host::DumpDescription(logged_description);
      ILQSolver logged_solver(logged_problem, log_params);
      bool logged_success = false;
      const std::shared_ptr<SolverLog> logged = logged_solver.Solve(&logged_success);
      std::ofstream ol(outdir + "/ilq_logged.txt");
      ol << std::setprecision(9) << "x0 " << logged_problem->InitialState() << "\n";
      WriteLog(ol, *logged, logged_success);
      CHECK(logged->Save(false, "ilq_logged_log"));
      CHECK(SaveLogs(std::vector<std::shared_ptr<const SolverLog>>{logged}, true, "ilq_logged_last"));
      std::ofstream(outdir + "/ilq_logged_log_meta.txt") << "iterates " << logged->NumIterates() << " device_iterations "
                                                          << logged->DeviceIterations() << "\n";
    }

    // the anytime exit (src/ilq_solver.cpp:123-124): a budget no iteration fits in returns iterate 0 with success
    {
      auto timed_problem = std::make_shared<MergeScene>(false);
      timed_problem->Initialize();
      ILQSolver timed(timed_problem, params);
      bool timed_success = false;
      const std::shared_ptr<SolverLog> timed_log = timed.Solve(&timed_success, 1e-6);
      std::ofstream od(outdir + "/ilq_deadline.txt");
      od << std::setprecision(9) << "x0 " << timed_problem->InitialState() << "\n";
      od << "iterates " << timed_log->NumIterates() << "\n";
      WriteLog(od, *timed_log, timed_success);
      // ... and a generous one changes nothing
      auto relaxed_problem = std::make_shared<MergeScene>(false);
      relaxed_problem->Initialize();
      ILQSolver relaxed(relaxed_problem, params);
      bool relaxed_success = false;
      const std::shared_ptr<SolverLog> relaxed_log = relaxed.Solve(&relaxed_success, 3600.0);
      std::ofstream orx(outdir + "/ilq_relaxed.txt");
      orx << std::setprecision(9) << "x0 " << relaxed_problem->InitialState() << "\n";
      WriteLog(orx, *relaxed_log, relaxed_success);
    }

    // receding horizon: adopt the solution, re-sync to a measured state 0.33 s into the plan
    {
      auto rh_problem = std::make_shared<MergeScene>(false);
      rh_problem->Initialize();
      rh_problem->OverwriteSolution(log->FinalOperatingPoint(), log->FinalStrategies());
      VectorXf x_meas = log->FinalOperatingPoint().xs[3];
      for (int e = 0; e < x_meas.size(); e++) x_meas(e) += 0.02f * static_cast<float>((e % 3) - 1);
      rh_problem->SetUpNextRecedingHorizon(x_meas, 0.33, 0.25);
      std::ofstream orh(outdir + "/receding.txt");
      orh << std::setprecision(9) << "x_meas " << x_meas << "\nt0 " << rh_problem->InitialTime() << "\nx0 "
          << rh_problem->InitialState() << "\n";
      const OperatingPoint& op = rh_problem->CurrentOperatingPoint();
      for (size_t k = 0; k < op.xs.size(); k++) {
        orh << "x " << op.xs[k] << "\n";
        orh << "u";
        for (const auto& u : op.us[k]) orh << " " << u;
        orh << "\n";
      }
    }

    std::vector<VectorXf> x0s;
    for (int b = 0; b < 6; b++) {
      VectorXf x0 = problem->InitialState();
      x0(kCar1 + Car::kPxIdx) += 0.7f * b;
      x0(kCar2 + Car::kPyIdx) -= 0.4f * b;
      x0(kWalker + Walker::kVIdx) += 0.05f * b;
      x0s.push_back(x0);
    }
    const host::BatchResult batch = solver.SolveBatch(x0s);
    for (size_t b = 0; b < x0s.size(); b++) {
      std::ofstream ob(outdir + "/ilq_batch_" + std::to_string(b) + ".txt");
      ob << std::setprecision(9) << "x0 " << x0s[b] << "\n";
      WriteLog(ob, *batch.logs[b], batch.success[b]);
    }
  }

  // 2. constrained scene through the augmented-Lagrangian outer loop
  {
    SolverParams al_params(params);
    al_params.max_solver_iters = 30;
    al_params.unconstrained_solver_max_iters = 5;
    auto problem = std::make_shared<MergeScene>(true);
    problem->Initialize();
    CHECK(problem->IsConstrained());
    host::ProblemDescription description;
    std::string why;
    CHECK(host::DescribeProblem(*problem, al_params, ILQG_F64, &description, &why)) << why;
    std::ofstream(outdir + "/scene_constrained.txt") << host::DumpDescription(description);
    AugmentedLagrangianSolver solver(problem, al_params);
    bool success = false;
    const std::shared_ptr<SolverLog> log = solver.Solve(&success);
    std::ofstream os(outdir + "/al_single.txt");
    os << std::setprecision(9) << "x0 " << problem->InitialState() << "\n";
    WriteLog(os, *log, success);
  }

  // 3. the receding-horizon loop of the reference's examples, with a fixed simulated solve time
  {
    SolverParams rh_params(params);
    rh_params.max_solver_iters = 8;
    rh_params.max_backtracking_steps = 100;
    rh_params.initial_alpha_scaling = 0.5f;
    rh_params.convergence_tolerance = 0.5f;
    auto problem = std::make_shared<MergeScene>(false);
    problem->Initialize();
    host::ProblemDescription description;
    std::string why;
    CHECK(host::DescribeProblem(*problem, rh_params, ILQG_F64, &description, &why)) << why;
    std::ofstream(outdir + "/scene_rh.txt") << host::DumpDescription(description);
    ILQSolver solver(problem, rh_params);
    host::Options().simulated_solve_time = 0.25;
    const auto logs = RecedingHorizonSimulator(3.0, 0.25, &solver);
    host::Options().simulated_solve_time = -1.0;
    std::ofstream os(outdir + "/rh_sim.txt");
    os << std::setprecision(9) << "calls " << logs.size() << "\n";
    for (const auto& log : logs) {
      os << "t0 " << log->FinalOperatingPoint().t0 << "\n";
      WriteLog(os, *log, true);
    }
    // the same loop for a batch of initial states, plans resident on the device
    auto batch_problem = std::make_shared<MergeScene>(false);
    batch_problem->Initialize();
    ILQSolver batch_solver(batch_problem, rh_params);
    std::vector<VectorXf> x0s;
    for (int b = 0; b < 3; b++) {
      VectorXf x0 = batch_problem->InitialState();
      x0(kCar1 + Car::kPxIdx) += 0.9f * b;
      x0(kCar2 + Car::kVIdx) += 0.3f * b;
      x0s.push_back(x0);
    }
    const auto batch_logs = host::RecedingHorizonSimulatorBatch(3.0, 0.25, &batch_solver, x0s, 0.25);
    for (size_t b = 0; b < x0s.size(); b++) {
      std::ofstream ob(outdir + "/rh_batch_" + std::to_string(b) + ".txt");
      ob << std::setprecision(9) << "calls " << batch_logs[b].size() << "\n";
      for (const auto& log : batch_logs[b]) {
        ob << "t0 " << log->FinalOperatingPoint().t0 << "\n";
        WriteLog(ob, *log, true);
      }
    }
  }

  // 3b. two planners over one plant (MinimallyInvasiveRecedingHorizonSimulator): once with the same scene on both
  // sides, once with a safety scene whose P1 value is far below the threshold
  for (int variant = 0; variant < 2; variant++) {
    SolverParams mi_params(params);
    mi_params.max_solver_iters = 8;
    mi_params.max_backtracking_steps = 100;
    mi_params.initial_alpha_scaling = 0.5f;
    mi_params.convergence_tolerance = 0.5f;
    auto plan_a = std::make_shared<MergeScene>(false);
    auto plan_b = std::make_shared<MergeScene>(false, variant == 1);
    plan_a->Initialize();
    plan_b->Initialize();
    ILQSolver original(plan_a, mi_params), safety(plan_b, mi_params);
    std::vector<std::shared_ptr<const SolverLog>> logs_a, logs_b;
    host::Options().simulated_solve_time = 0.25;
    const std::vector<ActiveProblem> active =
        MinimallyInvasiveRecedingHorizonSimulator(3.0, 0.25, &original, &safety, &logs_a, &logs_b);
    host::Options().simulated_solve_time = -1.0;
    std::ofstream os(outdir + "/mi_sim_" + std::to_string(variant) + ".txt");
    os << std::setprecision(9) << "active";
    for (ActiveProblem which : active) os << " " << (which == ActiveProblem::SAFETY ? 1 : 0);
    os << "\ncalls " << logs_a.size() << " " << logs_b.size() << "\n";
    for (size_t r = 0; r < logs_a.size() && r < logs_b.size(); r++) {
      for (const auto* log : {logs_a[r].get(), logs_b[r].get()}) {
        os << "t0 " << log->FinalOperatingPoint().t0 << "\n";
        WriteLog(os, *log, true);
      }
    }
  }

  // 4. shared-state dynamics: TwoPlayerUnicycle4D
  {
    SolverParams up(params);
    up.max_solver_iters = 5;
    up.max_backtracking_steps = 100;
    up.initial_alpha_scaling = 0.5f;
    up.expected_decrease_fraction = 0.001f;
    up.convergence_tolerance = 0.01f;
    auto problem = std::make_shared<PushedUnicycle>();
    problem->Initialize();
    host::ProblemDescription description;
    std::string why;
    CHECK(host::DescribeProblem(*problem, up, ILQG_F64, &description, &why)) << why;
    std::ofstream(outdir + "/scene_unicycle.txt") << host::DumpDescription(description);
    ILQSolver solver(problem, up);
    bool success = false;
    const std::shared_ptr<SolverLog> log = solver.Solve(&success);
    std::ofstream os(outdir + "/unicycle_single.txt");
    os << std::setprecision(9) << "x0 " << problem->InitialState() << "\n";
    WriteLog(os, *log, success);
    // equilibrium checks on the adopted solution
    problem->OverwriteSolution(log->FinalOperatingPoint(), log->FinalStrategies());
    std::ofstream oc(outdir + "/unicycle_checks.txt");
    oc << std::setprecision(9) << "costs";
    for (float c : ComputeStrategyCosts(*problem)) oc << " " << c;
    oc << "\nnash_small " << (NumericalCheckLocalNashEquilibrium(*problem, 0.0f) ? 1 : 0) << "\nnash_large "
       << (NumericalCheckLocalNashEquilibrium(*problem, 0.5f) ? 1 : 0) << "\nsufficient "
       << (CheckSufficientLocalNashEquilibrium(*problem) ? 1 : 0) << "\n";
  }

  // 5. the LQ seam on its own
  RunLqGame(outdir, false);
  RunLqGame(outdir, true);
  std::cout << "host_solve_demo: done\n";
  return 0;
}
