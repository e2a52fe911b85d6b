// GPU test program in the style of the reference's test/test_lq_solver.cpp, written against the mirrored
// API: LQFeedbackSolver / LQOpenLoopSolver run on the device through the C ABI, the properties are checked on
// the host with plain float arithmetic.
//   1. feedback gains of a long-horizon time-invariant two-player game match Lyapunov iterations;
//   2. the feedback solution is a Nash equilibrium: no player lowers its own cost by perturbing its own
//      strategy (alpha at one time step), with and without linear cost terms;
//   3. single player: open-loop and feedback solutions give the same first control.
// Prints one line per check and returns the number of failures.
#include <ilqgames/solver/lq_feedback_solver.h>
#include <ilqgames/solver/lq_open_loop_solver.h>

#include <cmath>
#include <iostream>

namespace {
using namespace ilqgames;

constexpr size_t kHorizon = 100;
constexpr float kDt = 0.1f;

// Shapes only: the LQ solvers ask their dynamics for dimensions.
class Shapes : public MultiPlayerDynamicalSystem {
 public:
  Shapes(Dimension n, std::vector<Dimension> udims) : MultiPlayerDynamicalSystem(n), udims_(std::move(udims)) {}
  Dimension UDim(PlayerIndex i) const override { return udims_[i]; }
  PlayerIndex NumPlayers() const override { return static_cast<PlayerIndex>(udims_.size()); }
  std::vector<Dimension> PositionDimensions() const override { return {0}; }

 private:
  std::vector<Dimension> udims_;
};

struct TwoPlayerGame {
  MatrixXf A, B1, B2, Q1, Q2;
  VectorXf l1, l2;
  float R11, R12, R21, R22, r11, r22;
  std::vector<LinearDynamicsApproximation> lin;
  std::vector<std::vector<QuadraticCostApproximation>> quad;

  explicit TwoPlayerGame(float nominal) {
    A = MatrixXf::Identity(2, 2);
    A(0, 1) = kDt;
    B1 = MatrixXf::Zero(2, 1);
    B2 = MatrixXf::Zero(2, 1);
    B1(0, 0) = 0.05f * kDt;
    B1(1, 0) = 1.0f * kDt;
    B2(0, 0) = 0.032f * kDt;
    B2(1, 0) = 0.11f * kDt;
    Q1 = MatrixXf::Identity(2, 2);
    Q2 = -1.0f * MatrixXf::Identity(2, 2);  // the second player pushes away from the origin, weakly
    Q2(0, 0) = -0.5f;
    Q2(1, 1) = -0.1f;
    // linear terms as a QuadraticCost with a nominal value would produce them: grad = -Q * nominal
    l1 = -1.0f * (Q1 * VectorXf::Constant(2, nominal));
    l2 = -1.0f * (Q2 * VectorXf::Constant(2, nominal));
    R11 = 1.0f;
    R12 = 0.5f;
    R21 = 0.25f;
    R22 = 1.0f;
    r11 = -R11 * nominal;
    r22 = -R22 * nominal;
    LinearDynamicsApproximation step;
    step.A = A;
    step.Bs = {B1, B2};
    lin.assign(kHorizon, step);
    QuadraticCostApproximation q1(2), q2(2);
    q1.state.hess = Q1;
    q1.state.grad = l1;
    q2.state.hess = Q2;
    q2.state.grad = l2;
    auto scalar = [](float h, float g) {
      MatrixXf H(1, 1);
      H(0, 0) = h;
      VectorXf G(1);
      G(0) = g;
      return SingleCostApproximation(H, G);
    };
    q1.Control(0, 1) = scalar(R11, r11);
    q1.Control(1, 1) = scalar(R12, 0.0f);
    q2.Control(0, 1) = scalar(R21, 0.0f);
    q2.Control(1, 1) = scalar(R22, r22);
    quad.assign(kHorizon, {q1, q2});
  }

  // Total cost of each player when the strategies are played from x0 on the (exactly linear) dynamics; the
  // operating point is the origin, so delta_x = x and u_ref = 0.
  void Costs(const std::vector<Strategy>& s, const VectorXf& x0, double* c1, double* c2) const {
    VectorXf x = x0;
    *c1 = *c2 = 0.0;
    for (size_t k = 0; k < kHorizon; k++) {
      const float u1 = s[0](k, x, VectorXf::Zero(1))(0), u2 = s[1](k, x, VectorXf::Zero(1))(0);
      *c1 += 0.5 * x.dot(Q1 * x) + l1.dot(x) + 0.5 * R11 * u1 * u1 + r11 * u1 + 0.5 * R12 * u2 * u2;
      *c2 += 0.5 * x.dot(Q2 * x) + l2.dot(x) + 0.5 * R22 * u2 * u2 + r22 * u2 + 0.5 * R21 * u1 * u1;
      VectorXf v1(1), v2(1);
      v1(0) = u1;
      v2(0) = u2;
      x = A * x + B1 * v1 + B2 * v2;
    }
  }
};

int failures = 0;
void Report(const char* name, bool ok, double detail) {
  std::cout << (ok ? "PASS " : "FAIL ") << name << " (" << detail << ")\n";
  if (!ok) failures++;
}

void FeedbackMatchesLyapunov() {
  const TwoPlayerGame g(0.0f);
  const auto dyn = std::make_shared<Shapes>(2, std::vector<Dimension>{1, 1});
  LQFeedbackSolver solver(dyn, kHorizon);
  const std::vector<Strategy> s = solver.Solve(g.lin, g.quad, VectorXf::Zero(2));
  // Lyapunov iterations for the stationary gains (scalar controls: the "solve" is a division)
  MatrixXf Z1 = g.Q1, Z2 = g.Q2, P1(1, 2), P2(1, 2);
  auto gain = [](const MatrixXf& B, const MatrixXf& Z, const MatrixXf& F, float R) {
    const MatrixXf num = B.transpose() * Z * F;
    const float den = R + (B.transpose() * Z * B)(0, 0);
    return (1.0f / den) * num;
  };
  P1 = gain(g.B1, Z1, g.A, g.R11);
  P2 = gain(g.B2, Z2, g.A, g.R22);
  for (int it = 0; it < 100; it++) {
    const MatrixXf old1 = P1, old2 = P2;
    P1 = gain(g.B1, Z1, g.A - g.B2 * old2, g.R11);
    P2 = gain(g.B2, Z2, g.A - g.B1 * old1, g.R22);
    const MatrixXf F = g.A - g.B1 * P1 - g.B2 * P2;
    Z1 = F.transpose() * Z1 * F + g.R11 * (P1.transpose() * P1) + g.R12 * (P2.transpose() * P2) + g.Q1;
    Z2 = F.transpose() * Z2 * F + g.R21 * (P1.transpose() * P1) + g.R22 * (P2.transpose() * P2) + g.Q2;
  }
  const float e1 = (P1 - s[0].Ps[0]).norm(), e2 = (P2 - s[1].Ps[0]).norm();
  Report("feedback gains at k=0 match Lyapunov iterations", e1 < 1e-3f && e2 < 1e-3f, std::fmax(e1, e2));
}

void FeedbackIsNash(float nominal) {
  const TwoPlayerGame g(nominal);
  const auto dyn = std::make_shared<Shapes>(2, std::vector<Dimension>{1, 1});
  LQFeedbackSolver solver(dyn, kHorizon);
  VectorXf x0(2);
  x0(0) = 1.0f;
  x0(1) = -0.5f;
  const std::vector<Strategy> s = solver.Solve(g.lin, g.quad, x0);
  double c1, c2;
  g.Costs(s, x0, &c1, &c2);
  double worst = 0.0;  // most negative (perturbed - nominal) over all single-step perturbations
  for (PlayerIndex i = 0; i < 2; i++)
    for (size_t k = 0; k + 1 < kHorizon; k += 7)
      for (float delta : {-0.1f, 0.1f}) {
        std::vector<Strategy> p = s;
        p[i].alphas[k](0) += delta;
        double d1, d2;
        g.Costs(p, x0, &d1, &d2);
        worst = std::fmin(worst, i == 0 ? d1 - c1 : d2 - c2);
      }
  Report(nominal == 0.0f ? "feedback solution is a Nash equilibrium" : "... also with linear cost terms",
         worst > -1e-3 * (1.0 + std::fabs(c1) + std::fabs(c2)), worst);
}

void SinglePlayerOpenLoopEqualsFeedback() {
  LinearDynamicsApproximation step;
  step.A = MatrixXf::Identity(2, 2);
  step.A(0, 1) = kDt;
  step.Bs = {(kDt * 0.41f) * MatrixXf::Identity(2, 2)};
  QuadraticCostApproximation q(2);
  q.state.hess = MatrixXf::Identity(2, 2);
  q.Control(0, 2) = SingleCostApproximation(MatrixXf::Identity(2, 2), VectorXf::Zero(2));
  const std::vector<LinearDynamicsApproximation> lin(kHorizon, step);
  const std::vector<std::vector<QuadraticCostApproximation>> quad(kHorizon, {q});
  const auto dyn = std::make_shared<Shapes>(2, std::vector<Dimension>{2});
  LQOpenLoopSolver ol(dyn, kHorizon);
  LQFeedbackSolver fb(dyn, kHorizon);
  const VectorXf x0 = VectorXf::Constant(2, 1.0f);
  const std::vector<Strategy> s_ol = ol.Solve(lin, quad, x0);
  const std::vector<Strategy> s_fb = fb.Solve(lin, quad, x0);
  const VectorXf u_ol = s_ol[0](0, VectorXf::Zero(2), VectorXf::Zero(2));
  const VectorXf u_fb = s_fb[0](0, x0, VectorXf::Zero(2));
  const float err = (u_ol - u_fb).norm(), scale = u_fb.norm();
  std::cout << "  u_ol = " << u_ol << "   u_fb = " << u_fb << "\n";
  Report("single player: open-loop and feedback give the same first control", scale > 0.1f && err < 0.01f * scale,
         err / scale);
}

}  // namespace

int main() {
  FeedbackMatchesLyapunov();
  FeedbackIsNash(0.0f);
  FeedbackIsNash(0.5f);
  SinglePlayerOpenLoopEqualsFeedback();
  std::cout << (failures == 0 ? "all checks passed\n" : "some checks FAILED\n");
  return failures;
}
