// Test tool (no GPU): fills a SolverLog through the mirrored API and saves it with SolverLog::Save /
// SaveLogs; tests/test_host_mirror.py checks the directory layout and the row formatting.
#include <ilqgames/utils/solver_log.h>

#include <iostream>

int main(int argc, char** argv) {
  using namespace ilqgames;
  if (argc < 2) return 2;
  const std::string experiment = argv[1];
  OperatingPoint op(3, 2, 1.5);
  for (size_t k = 0; k < 3; k++) {
    op.xs[k] = VectorXf::Zero(4);
    for (int e = 0; e < 4; e++) op.xs[k](e) = 0.25f * static_cast<float>(k) - 10.0f * e + (e == 2 ? 1234.5678f : 0.0f);
    op.us[k][0] = VectorXf::Zero(2);
    op.us[k][1] = VectorXf::Zero(1);
    op.us[k][0](0) = 0.5f * k;
    op.us[k][0](1) = -1.0f / 3.0f;
    op.us[k][1](0) = 100.0f + k;
  }
  std::vector<Strategy> strategies;
  strategies.emplace_back(3, 4, 2);
  strategies.emplace_back(3, 4, 1);
  auto log = std::make_shared<SolverLog>();
  log->AddSolverIterate(op, strategies, {1.5f, 2.25f}, 0.125, false);
  op.t0 = 2.0;
  log->AddSolverIterate(op, strategies, {0.75f, 1e-3f}, 0.25, true);
  if (!log->Save(false, experiment)) return 1;
  std::vector<std::shared_ptr<const SolverLog>> logs = {log, log};
  if (!SaveLogs(logs, true, experiment + "_list")) return 1;
  if (log->Save(false, experiment)) return 3;  // the directory exists now: MakeDirectory fails, as in the reference
  std::cout << "ok\n";
  return 0;
}
