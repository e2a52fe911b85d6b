// Test tool: instantiate a Problem subclass written against the reference API, flatten it with
// host::DescribeProblem and print the description (no GPU needed).  The Problem class comes from
// the translation unit this file is linked with: EXAMPLE_HEADER / EXAMPLE_CLASS are set on the
// compiler command line by tests/test_host_mirror.py.
#include EXAMPLE_HEADER

#include <iostream>

int main() {
  auto problem = std::make_shared<ilqgames::EXAMPLE_CLASS>();
  problem->Initialize();
  ilqgames::SolverParams params;
  ilqgames::host::ProblemDescription description;
  std::string why;
  if (!ilqgames::host::DescribeProblem(*problem, params, ILQG_F64, &description, &why)) {
    std::cerr << "unsupported: " << why << "\n";
    return 2;
  }
  std::cout << ilqgames::host::DumpDescription(description);
  std::cout.precision(9);
  std::cout << "x0";
  for (int i = 0; i < problem->InitialState().size(); i++) std::cout << " " << problem->InitialState()(i);
  std::cout << "\n";
  return 0;
}
