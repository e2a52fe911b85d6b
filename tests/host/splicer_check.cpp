// CPU-only driver for the host-side SolutionSplicer mirror: builds two solver logs with synthetic
// operating points / strategies, splices the second into the first at several offsets and prints the
// result; tests/test_host_mirror.py compares with the oracle's restatement of
// src/solution_splicer.cpp:56-129.  No device call is made.
#include <ilqgames/solver/solution_splicer.h>
#include <ilqgames/utils/solver_log.h>

#include <iomanip>
#include <iostream>

namespace {

using namespace ilqgames;

constexpr int kN = 3, kM = 2;  // one "player" with a 3-state, 2-control toy system: only shapes matter here

SolverLog MakeLog(Time t0, float tag) {
  const size_t T = time::kNumTimeSteps;
  OperatingPoint op(T, 1, t0);
  std::vector<Strategy> strategies;
  strategies.emplace_back(T, kN, kM);
  for (size_t k = 0; k < T; k++) {
    op.xs[k] = VectorXf::Zero(kN);
    op.us[k][0] = VectorXf::Zero(kM);
    for (int e = 0; e < kN; e++) op.xs[k](e) = tag + 0.01f * static_cast<float>(k) + 0.001f * e;
    for (int e = 0; e < kM; e++) {
      op.us[k][0](e) = -tag - 0.02f * static_cast<float>(k) + 0.003f * e;
      strategies[0].alphas[k](e) = 0.5f * tag + 0.001f * static_cast<float>(k * (e + 1));
      for (int c = 0; c < kN; c++) strategies[0].Ps[k](e, c) = tag + 0.1f * e + 0.01f * c + 0.0001f * k;
    }
  }
  SolverLog log;
  log.AddSolverIterate(op, strategies, {0.0f}, 0.0, true);
  return log;
}

void Print(const SolutionSplicer& s) {
  const OperatingPoint& op = s.CurrentOperatingPoint();
  std::cout << std::setprecision(9) << "plan " << op.xs.size() << " " << op.t0 << "\n";
  for (size_t k = 0; k < op.xs.size(); k++) {
    std::cout << "row " << op.xs[k] << " " << op.us[k][0] << " " << s.CurrentStrategies()[0].alphas[k];
    for (int i = 0; i < kM * kN; i++) std::cout << " " << s.CurrentStrategies()[0].Ps[k].data()[i];
    std::cout << "\n";
  }
}

}  // namespace

int main() {
  const double starts[] = {0.0, 0.2, 0.5, 0.6, 3.0};
  for (double start : starts) {
    SolutionSplicer splicer(MakeLog(1.5, 1.0f));
    splicer.Splice(MakeLog(1.5 + start, 2.0f));
    std::cout << "case " << start << " contains " << splicer.ContainsTime(1.5 + start + 9.9) << " "
              << splicer.ContainsTime(1.0) << "\n";
    Print(splicer);
    splicer.Splice(MakeLog(splicer.CurrentOperatingPoint().t0 + 0.7, 3.0f));  // again, on a 100..105-row plan
    Print(splicer);
  }
  return 0;
}
