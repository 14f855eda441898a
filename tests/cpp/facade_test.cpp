// facade_test.cpp -- the reference's own QP tests, re-stated against the C++ facade (mini_opt_amd/cpp/mini_opt_hip.hpp).
// Each test cites the reference test it mirrors (test/qp_test.cc).  Expected values of the elimination test are the
// committed golden fixture (tests/golden/elimination.json, numpy full-system LU); the KATs use the analytic optima.
// Built by __graft_entry__.build() with hipcc (host code only) and run on the GPU box by tests/test_gpu_facade.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../mini_opt_amd/cpp/mini_opt_hip.hpp"

using namespace mini_opt_hip;

static int g_fail = 0;
#define EXPECT_NEAR(a, b, tol)                                                                       \
  do {                                                                                               \
    const double va__ = (a), vb__ = (b);                                                             \
    if (!(std::fabs(va__ - vb__) <= (tol))) {                                                        \
      std::printf("FAIL %s:%d: %s = %.17g vs %s = %.17g (tol %g)\n", __FILE__, __LINE__, #a, va__, #b, vb__, (double)(tol)); \
      ++g_fail;                                                                                      \
    }                                                                                                \
  } while (0)
#define EXPECT_TRUE(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

struct Root { double a, b; };
// QPSolverTest::BuildQuadratic, qp_test.cc:58-73
static void BuildQuadratic(const std::vector<Root>& roots, QP* out) {
  *out = QP((int)roots.size());
  for (size_t i = 0; i < roots.size(); ++i) {
    out->G_at((int)i, (int)i) = roots[i].a * roots[i].a;
    out->c[i] = -2 * roots[i].a * roots[i].b;
  }
}

// LinearInequalityConstraintTest, qp_test.cc:27-45
static void TestLinearInequalityConstraint() {
  const LinearInequalityConstraint c1(3, 2.0, -4.0);
  EXPECT_TRUE(c1.IsFeasible(2.1));
  EXPECT_TRUE(!c1.IsFeasible(1.9));
  const LinearInequalityConstraint shifted = c1.ShiftTo(1.0);
  EXPECT_TRUE(shifted.IsFeasible(1.1));
  EXPECT_TRUE(!shifted.IsFeasible(0.9));
  EXPECT_TRUE((Var(0) >= 0.3).IsFeasible(0.5));
  EXPECT_TRUE(!(Var(0) <= -0.9).IsFeasible(1.2));
  EXPECT_NEAR(0.0, (Var(0) >= 0.0).ClampX(-1.3), 1e-12);
  EXPECT_NEAR(0.5, (Var(0) >= 0.5).ClampX(-0.9), 1e-12);
  EXPECT_NEAR(1.5, (Var(0) >= 1.3).ClampX(1.5), 1e-12);
  EXPECT_NEAR(0.0, (Var(0) <= 0.0).ClampX(5.0), 1e-12);
  EXPECT_NEAR(-1.3, (Var(0) <= -1.3).ClampX(1.0), 1e-12);
  EXPECT_NEAR(6.0, (Var(0) <= 10.0).ClampX(6.0), 1e-12);
}

// TestEliminationAllConstraints, qp_test.cc:208-241 with the dummy state of :84-97; expected delta: golden fixture.
static void TestEliminationAllConstraints() {
  QP qp;
  BuildQuadratic({{0.5, 2.0}, {5.0, 25.0}, {3.0, 9.0}, {4.0, 1.0}, {1.2, 2.4}, {-1.0, 2.0}, {-0.5, 2.0}}, &qp);
  qp.ResizeEqualities(2);
  qp.A_at(0, 1) = 2.0; qp.A_at(0, 4) = -1.0; qp.A_at(1, 0) = 3.0;
  qp.b_eq = {0.5, -2.0};
  qp.constraints.emplace_back(3, 4.0, -8.0);
  qp.constraints.emplace_back(5, 2.0, 1.0);
  qp.constraints.emplace_back(6, 1.0, 0.0);
  QPInteriorPointSolver solver(&qp);
  std::vector<double> v = {0.0, 0.1, 0.2, 0.55, 0.3, 0.7, 1.0};
  for (int i = 0; i < 3; ++i) v.push_back(2.0 / (i + 1));          // s
  for (int q = 0; q < 2; ++q) v.push_back((q + 1.0) * (q + 1.0));  // y
  for (int i = 0; i < 3; ++i) v.push_back(0.5 * (i + 1));          // z
  solver.SetVariables(v);
  const std::vector<double>& delta = solver.NewtonStep(0.0);
  const double expected[15] = {0.6666666666666666, 8.35513654096229, 5.800000000000001, 0.35, 17.110273081924575, -1.5,
                               -1.2000000000000002, -6.4, -1.6, -0.8666666666666667, -20.31079323797139, -4.611111111111112,
                               1.1, 0.6, 0.44999999999999996};
  for (int i = 0; i < 15; ++i) EXPECT_NEAR(expected[i], delta[i], 1e-12);  // ASSERT_EIGEN_NEAR(update, solver.delta_, kPico)
}

// TestWithInequalitiesAndEqualities, qp_test.cc:439-471
static void TestWithInequalitiesAndEqualities() {
  QP qp;
  BuildQuadratic({{1.0, 1.0}, {5.0, -10.0}, {10.0, 2.0}}, &qp);
  qp.ResizeEqualities(1);
  qp.A_at(0, 2) = 1.0;
  qp.b_eq = {-2.0};
  qp.constraints.push_back(Var(0) <= 0.5);
  qp.constraints.push_back(Var(1) >= -1.0);
  QPInteriorPointSolver solver(&qp);
  for (InitialGuessMethod method : {InitialGuessMethod::NAIVE, InitialGuessMethod::SOLVE_EQUALITY_CONSTRAINED}) {
    QPInteriorPointSolver::Params params{};
    params.termination_kkt_tol = 1e-12;
    params.initial_mu = 0.1;
    params.sigma = 0.1;
    params.initial_guess_method = method;
    const auto outputs = solver.Solve(params);
    EXPECT_TRUE(outputs.termination_state == QPInteriorPointTerminationState::SATISFIED_KKT_TOL);
    EXPECT_NEAR(0.5, solver.x_block()[0], 1e-6);
    EXPECT_NEAR(-1.0, solver.x_block()[1], 1e-6);
    EXPECT_NEAR(2.0, solver.x_block()[2], 1e-6);
    EXPECT_NEAR(0.0, solver.s_block()[0], 1e-6);
    EXPECT_NEAR(0.0, solver.s_block()[1], 1e-6);
    EXPECT_TRUE(outputs.lagrange_multipliers.has_value());
    EXPECT_TRUE(!outputs.iterations.empty());
  }
}

// TestWithFullyConstrainedEqualities, qp_test.cc:414-436
static void TestWithFullyConstrainedEqualities() {
  QP qp;
  BuildQuadratic({{1.0, -0.5}, {1.0, -0.25}, {1.0, 1.0}}, &qp);
  qp.ResizeEqualities(3);
  for (int i = 0; i < 3; ++i) qp.A_at(i, i) = 1.0;
  qp.b_eq = {-1.0, -2.0, -3.0};
  QPInteriorPointSolver solver(&qp);
  QPInteriorPointSolver::Params params{};
  params.termination_kkt_tol = 1e-6;
  params.max_iterations = 1;
  const auto outputs = solver.Solve(params);
  EXPECT_TRUE(outputs.termination_state == QPInteriorPointTerminationState::SATISFIED_KKT_TOL);
  for (int i = 0; i < 3; ++i) EXPECT_NEAR(-qp.b_eq[i], solver.x_block()[i], 1e-9);
  for (int i = 0; i < 3; ++i) EXPECT_TRUE(solver.y_block()[i] > 1e-2);
}

// error convention: the reference throws assert::default_error / FailedFactorization (qp.cc:21-34, 303-307)
static void TestErrors() {
  bool threw = false;
  try { QPInteriorPointSolver s(nullptr); } catch (const default_error&) { threw = true; }
  EXPECT_TRUE(threw);
  QP qp(2);  // G = 0: singular reduced system -> FailedFactorization (zero pivot with a non-zero column is impossible here:
  qp.G_at(0, 0) = 1.0;  // one zero pivot is tolerated like Eigen, so make it indefinite-with-coupling instead)
  qp.G_at(1, 0) = 2.0;
  qp.G_at(1, 1) = 4.0;  // [[1,2],[2,4]] is singular: second pivot is exactly zero, nothing below it -> tolerated
  QPInteriorPointSolver s(&qp);
  threw = false;
  try { QPInteriorPointSolver::Params p{}; p.sigma = 2.0; (void)s.Solve(p); } catch (const default_error&) { threw = true; }
  EXPECT_TRUE(threw);  // CheckParams, qp.cc:76-82
}

int main() {
  TestLinearInequalityConstraint();
  TestEliminationAllConstraints();
  TestWithInequalitiesAndEqualities();
  TestWithFullyConstrainedEqualities();
  TestErrors();
  if (g_fail) { std::printf("%d FAILURES\n", g_fail); return 1; }
  std::printf("facade_test: all tests passed\n");
  return 0;
}
