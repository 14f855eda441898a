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

// TestRosenbrock / TestInequalityConstrainedRosenbrock (nonlinear_test.cc:390-500) through the batched NLS facade: every
// initial guess of the reference test is one problem of the batch; residuals are host functors as in the reference.
static void RosenbrockHost(const double* x, int64_t batch, double* r, double* J, double*, double*) {
  const double sb = std::sqrt(100.0);
  for (int64_t p = 0; p < batch; ++p) {
    const double x0 = x[2 * p], x1 = x[2 * p + 1];
    r[2 * p] = 1.0 - x0; r[2 * p + 1] = sb * (x1 - x0 * x0);
    if (J) { J[4 * p] = -1.0; J[4 * p + 1] = 0.0; J[4 * p + 2] = -2.0 * x0 * sb; J[4 * p + 3] = sb; }
  }
}
static void TestNlsRosenbrock() {
  const std::vector<double> guesses = {-5, -3, 10, 8, -20, 3, 0, -5, 4, 0, 100, 50, -35, 40, 1000, -50, 0.8, -0.3};
  const int64_t B = (int64_t)guesses.size() / 2;
  BatchedConstrainedNonlinearLeastSquares nls(2, 2, 0, {}, RosenbrockHost, B);
  BatchedConstrainedNonlinearLeastSquares::Params p{};
  p.max_iterations = 5; p.max_qp_iterations = 1;
  const auto term = nls.Solve(p, guesses);
  for (int64_t i = 0; i < B; ++i) {
    EXPECT_TRUE(term[i] == NLSTerminationState::SATISFIED_ABSOLUTE_TOL);
    EXPECT_NEAR(1.0, nls.variables()[2 * i], 1e-6);
    EXPECT_NEAR(1.0, nls.variables()[2 * i + 1], 1e-6);
  }
  // Var(0) >= 1.2, Var(1) <= 0.5: the optimum sits on both constraints
  const std::vector<double> g2 = {12, -5, 100.0, -20.0, 1423.0, -400.0, -20.0, 10.0, -120.0, 35.0, -50.0, 0.5};
  BatchedConstrainedNonlinearLeastSquares nls2(2, 2, 0, {Var(0) >= 1.2, Var(1) <= 0.5}, RosenbrockHost, 6);
  BatchedConstrainedNonlinearLeastSquares::Params p2{};
  p2.max_iterations = 10; p2.max_qp_iterations = 10;
  const auto term2 = nls2.Solve(p2, g2);
  for (int i = 0; i < 6; ++i) {
    EXPECT_TRUE(term2[i] != NLSTerminationState::MAX_ITERATIONS && term2[i] != NLSTerminationState::MAX_LAMBDA);
    EXPECT_NEAR(1.2, nls2.variables()[2 * i], 1e-6);
    EXPECT_NEAR(0.5, nls2.variables()[2 * i + 1], 1e-6);
  }
  // CheckParams (nonlinear.cc:48-73) surfaces as default_error
  bool threw = false;
  try { p2.armijo_search_tau = 1.5; (void)nls2.Solve(p2, g2); } catch (const default_error&) { threw = true; }
  EXPECT_TRUE(threw);
}

// QP::ComputeEigenvalueStats (qp.hpp:122-123, qp.cc:12-16) and Params::log_qp_eigenvalues (nonlinear.hpp:122-123, nonlinear.cc:138): the reference
// has no test of its own for either; the expected values are closed forms (a diagonal G, 2 x 2 matrices read from their LOWER triangle as
// SelfAdjointEigenSolver does, and J^T J of the Rosenbrock residual at the starting point).
static void TestEigenvalueStats() {
  QP qp;
  BuildQuadratic({{2.0, 1.0}, {-0.5, 3.0}, {3.0, -1.0}, {1.5, 0.0}}, &qp);     // G = diag(4, 0.25, 9, 2.25)
  QPEigenvalues e = qp.ComputeEigenvalueStats();
  EXPECT_NEAR(0.25, e.min, 1e-13); EXPECT_NEAR(9.0, e.max, 1e-13); EXPECT_NEAR(0.25, e.abs_min, 1e-13);
  QP q2(2);
  q2.G_at(0, 0) = 1.0; q2.G_at(1, 1) = 1.0; q2.G_at(1, 0) = 2.0; q2.G_at(0, 1) = 777.0;   // the strict upper triangle is never read
  e = q2.ComputeEigenvalueStats();
  EXPECT_NEAR(-1.0, e.min, 1e-13); EXPECT_NEAR(3.0, e.max, 1e-13); EXPECT_NEAR(1.0, e.abs_min, 1e-13);
  // the SQP loop records them per outer iteration: G = J^T J (lambda = 0 at the start) of Rosenbrock at (x0, x1): [[1 + 400 x0^2, -200 x0], [., 100]]
  const std::vector<double> guesses = {-5, -3, 0.8, -0.3};
  BatchedConstrainedNonlinearLeastSquares nls(2, 2, 0, {}, RosenbrockHost, 2);
  BatchedConstrainedNonlinearLeastSquares::Params p{};
  p.max_iterations = 3; p.max_qp_iterations = 1; p.log_qp_eigenvalues = true;
  (void)nls.Solve(p, guesses);
  for (int b = 0; b < 2; ++b) {
    const auto& its = nls.outputs()[(size_t)b].iterations;
    EXPECT_TRUE(!its.empty() && its[0].qp_eigenvalues.has_value());
    if (its.empty() || !its[0].qp_eigenvalues) continue;
    const double x0 = guesses[2 * b], a = 1.0 + 400.0 * x0 * x0, off = -200.0 * x0, d = 100.0;
    const double mid = 0.5 * (a + d), rad = std::sqrt(0.25 * (a - d) * (a - d) + off * off);
    EXPECT_NEAR(mid - rad, its[0].qp_eigenvalues->min, 1e-9 * (mid + rad));
    EXPECT_NEAR(mid + rad, its[0].qp_eigenvalues->max, 1e-12 * (mid + rad));
    EXPECT_NEAR(mid - rad, its[0].qp_eigenvalues->abs_min, 1e-9 * (mid + rad));
  }
  BatchedConstrainedNonlinearLeastSquares::Params p0{};
  p0.max_iterations = 2; p0.max_qp_iterations = 1;
  (void)nls.Solve(p0, guesses);
  EXPECT_TRUE(!nls.outputs()[0].iterations.empty() && !nls.outputs()[0].iterations[0].qp_eigenvalues.has_value());   // off by default
}

// TestSphereWithNonlinearEqualityConstraints (nonlinear_test.cc:745-826): cost x (6 variables), x0 x1 = 4, x2 x3 = 9
static void SphereHost(const double* x, int64_t batch, double* r, double* J, double* r_eq, double* J_eq) {
  for (int64_t p = 0; p < batch; ++p) {
    const double* xp = x + 6 * p;
    for (int i = 0; i < 6; ++i) r[6 * p + i] = xp[i];
    if (J) for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) J[36 * p + 6 * i + j] = i == j ? 1.0 : 0.0;
    r_eq[2 * p] = xp[0] * xp[1] - 4.0; r_eq[2 * p + 1] = xp[2] * xp[3] - 9.0;
    if (J_eq) {
      double* Je = J_eq + 12 * p;
      for (int i = 0; i < 12; ++i) Je[i] = 0.0;
      Je[0] = xp[1]; Je[1] = xp[0]; Je[6 + 2] = xp[3]; Je[6 + 3] = xp[2];
    }
  }
}
static void TestNlsSphereWithEqualities() {
  const std::vector<double> guesses = {5.0, 3.0, -7.0, -2.0, 4.0, -9.0,   -12.0, -1.5, 20.0, 6.0, -3.0, 8.0,   25.0, 14.0, 3.0, 22.0, -17.0, 1.0};
  BatchedConstrainedNonlinearLeastSquares nls(6, 6, 2, {}, SphereHost, 3);
  BatchedConstrainedNonlinearLeastSquares::Params p{};
  p.max_iterations = 100; p.max_qp_iterations = 1; p.relative_exit_tol = 1e-12; p.absolute_first_derivative_tol = 1e-9;
  p.termination_kkt_tolerance = 1e-6; p.lambda_initial = 0.001;
  const auto term = nls.Solve(p, guesses);
  for (int i = 0; i < 3; ++i) {
    EXPECT_TRUE(term[i] == NLSTerminationState::SATISFIED_ABSOLUTE_TOL || term[i] == NLSTerminationState::SATISFIED_RELATIVE_TOL ||
                term[i] == NLSTerminationState::SATISFIED_FIRST_ORDER_TOL);
    const double* v = nls.variables().data() + 6 * i;
    EXPECT_NEAR(2.0, std::fabs(v[0]), 5e-5); EXPECT_NEAR(v[0], v[1], 5e-5);
    EXPECT_NEAR(3.0, std::fabs(v[2]), 5e-5); EXPECT_NEAR(v[2], v[3], 5e-5);
    EXPECT_NEAR(0.0, v[4], 5e-5); EXPECT_NEAR(0.0, v[5], 5e-5);
  }
}

// The C ABI alone: mo_nls_solve with a callback that only launches the library's device residual kernels (mo_residual_eval) --
// the Himmelblau sweep of nonlinear_test.cc:597-664 without a single host-side residual evaluation.
struct HimmelblauCtx { mo_plan* plan; double *vars, *cand, *J, *r, *r_cand; int64_t batch; };
static int HimmelblauEval(void* user, int32_t what, void* stream) {
  auto* c = static_cast<HimmelblauCtx*>(user);
  const bool lin = what == MO_NLS_EVAL_LINEARIZE;
  return mo_residual_eval(c->plan, MO_RESIDUAL_HIMMELBLAU, 2, nullptr, lin ? c->vars : c->cand, 2, c->batch, lin ? c->r : c->r_cand, 2,
                          lin ? c->J : nullptr, 4, 2, MO_ROW_MAJOR, stream);
}
static void TestNlsDeviceResiduals() {
  std::vector<double> guesses;
  for (double x = -4.5; x <= 4.5; x += 0.3) for (double y = -4.5; y <= 4.5; y += 0.3) { guesses.push_back(x); guesses.push_back(y); }
  const int64_t B = (int64_t)guesses.size() / 2;
  mo_plan_desc d{}; d.n = 2; d.k = 0; d.m = 4; d.m_r = 2; d.dtype = MO_F64; d.device = 0; d.max_batch = B;
  mo_plan* plan = nullptr;
  EXPECT_TRUE(mo_plan_create(&d, &plan) == MO_OK);
  detail::DeviceBuffer<double> vars, cand((size_t)B * 2), J((size_t)B * 4), r((size_t)B * 2), r_cand((size_t)B * 2), ca, cb;
  detail::DeviceBuffer<int32_t> cv, term((size_t)B);
  vars.Upload(guesses.data(), guesses.size());
  const std::vector<int32_t> hv = {0, 0, 1, 1};                       // Var(i) >= -5, Var(i) <= 5
  const std::vector<double> ha = {1.0, -1.0, 1.0, -1.0}, hb = {5.0, 5.0, 5.0, 5.0};
  cv.Upload(hv.data(), 4); ca.Upload(ha.data(), 4); cb.Upload(hb.data(), 4);
  mo_nls_problem np{};
  np.vars = vars.get(); np.vars_stride = 2; np.candidate = cand.get(); np.candidate_stride = 2;
  np.J = J.get(); np.J_stride = 4; np.J_ld = 2; np.J_layout = MO_ROW_MAJOR; np.r = r.get(); np.r_stride = 2;
  np.r_cand = r_cand.get(); np.r_cand_stride = 2;
  np.cons_var = cv.get(); np.cons_a = ca.get(); np.cons_b = cb.get(); np.cons_stride = 0;
  mo_nls_params sp; mo_default_nls_params(&sp);
  sp.max_iterations = 20; sp.max_qp_iterations = 10; sp.relative_exit_tol = 1e-12; sp.absolute_first_derivative_tol = 1e-8;
  sp.termination_kkt_tolerance = 1e-6;
  HimmelblauCtx ctx{plan, vars.get(), cand.get(), J.get(), r.get(), r_cand.get(), B};
  EXPECT_TRUE(mo_nls_solve(plan, &np, B, &sp, HimmelblauEval, &ctx, term.get(), nullptr, nullptr, nullptr, nullptr) == MO_OK);
  std::vector<double> x((size_t)B * 2); vars.Download(x.data(), x.size());
  std::vector<int32_t> t((size_t)B); term.Download(t.data(), t.size());
  const double sols[4][2] = {{3.0, 2.0}, {-2.805118, 3.131312}, {-3.779310, -3.283186}, {3.584428, -1.848126}};
  int bad = 0;
  for (int64_t i = 0; i < B; ++i) {
    double best = 1e9;
    for (auto& s : sols) best = std::fmin(best, std::hypot(x[2 * i] - s[0], x[2 * i + 1] - s[1]));
    const bool satisfied = t[i] == MO_NLS_SATISFIED_ABSOLUTE_TOL || t[i] == MO_NLS_SATISFIED_RELATIVE_TOL || t[i] == MO_NLS_SATISFIED_FIRST_ORDER_TOL;
    if (!satisfied || !(best < 5e-5)) ++bad;
  }
  EXPECT_TRUE(bad == 0);
  EXPECT_TRUE(B == 961);
  mo_plan_destroy(plan);
}

// mini_opt::Problem built from MakeResidual<R, P>(index, functor) residuals (residual.hpp:119-143, nonlinear.hpp:33-52) through the
// drop-in ConstrainedNonlinearLeastSquares: TestRosenbrock (nonlinear_test.cc:390-430) with the reference's functor shape, the sphere
// with two product equalities on different index pairs (nonlinear_test.cc:722-826: the scatter of per-residual Jacobians into the
// dense stack), a custom Retraction and SetUserExitCallback.
static void TestProblemOfResiduals() {
  const double sb = std::sqrt(100.0);
  auto rosenbrock = [sb](const VectorN<2>& x, JacobianRP<2, 2>* J) -> VectorN<2> {
    if (J) { (*J)(0, 0) = -1.0; (*J)(0, 1) = 0.0; (*J)(1, 0) = -2.0 * x[0] * sb; (*J)(1, 1) = sb; }
    VectorN<2> r; r[0] = 1.0 - x[0]; r[1] = sb * (x[1] - x[0] * x[0]);
    return r;
  };
  Problem problem{};
  problem.dimension = 2;
  problem.costs.push_back(MakeResidual<2, 2>({0, 1}, rosenbrock));
  EXPECT_TRUE(problem.costs[0].Dimension() == 2);
  EXPECT_NEAR(0.5 * (16.0 + 100.0 * 4.0), problem.costs[0].QuadraticError({-3.0, 7.0}), 1e-12);
  ConstrainedNonlinearLeastSquares nls(&problem);
  ConstrainedNonlinearLeastSquares::Params p{};
  p.max_iterations = 5; p.max_qp_iterations = 1;
  const NLSSolverOutputs out = nls.Solve(p, {-5.0, -3.0});
  EXPECT_TRUE(out.termination_state == NLSTerminationState::SATISFIED_ABSOLUTE_TOL);
  EXPECT_NEAR(1.0, nls.variables()[0], 1e-6); EXPECT_NEAR(1.0, nls.variables()[1], 1e-6);
  EXPECT_TRUE(!out.iterations.empty() && out.iterations.size() <= 5);
  EXPECT_TRUE(out.iterations.front().iteration == 0 && out.iterations.front().errors_initial.f > 100.0);
  EXPECT_TRUE(out.NumLineSearchSteps() >= (int)out.iterations.size());

  // sphere: one 6-dimensional cost, two scalar equality residuals on index pairs {0, 1} and {2, 3}; a Dynamic-sized residual on {4, 5}
  auto sphere = [](const VectorN<6>& x, JacobianRP<6, 6>* J) -> VectorN<6> {
    if (J) { J->setZero(); for (int i = 0; i < 6; ++i) (*J)(i, i) = 1.0; }
    return x;
  };
  auto product = [](double target) {
    return [target](const VectorN<2>& x, JacobianRP<1, 2>* J) -> VectorN<1> {
      if (J) { (*J)(0, 0) = x[1]; (*J)(0, 1) = x[0]; }
      VectorN<1> r; r[0] = x[0] * x[1] - target;
      return r;
    };
  };
  auto tail = [](const VectorN<Dynamic>& x, JacobianRP<1, Dynamic>* J) -> VectorN<1> {
    if (J) { (*J)(0, 0) = 1.0; (*J)(0, 1) = -1.0; }
    VectorN<1> r; r[0] = x[0] - x[1];
    return r;
  };
  Problem sp{};
  sp.dimension = 6;
  sp.costs.push_back(MakeResidual<6, 6>({0, 1, 2, 3, 4, 5}, sphere));
  sp.costs.push_back(MakeResidual<1, Dynamic>({4, 5}, tail));
  sp.equality_constraints.push_back(MakeResidual<1, 2>({0, 1}, product(4.0)));
  sp.equality_constraints.push_back(MakeResidual<1, 2>({2, 3}, product(9.0)));
  ConstrainedNonlinearLeastSquares snls(&sp, nullptr, 2);
  ConstrainedNonlinearLeastSquares::Params q{};
  q.max_iterations = 100; q.max_qp_iterations = 1; q.relative_exit_tol = 1e-12; q.absolute_first_derivative_tol = 1e-9;
  q.termination_kkt_tolerance = 1e-6; q.lambda_initial = 0.001;
  (void)snls.Solve(q, {5.0, 3.0, -7.0, -2.0, 4.0, -9.0,   -12.0, -1.5, 20.0, 6.0, -3.0, 8.0});
  for (int b = 0; b < 2; ++b) {
    const double* v = snls.variables().data() + 6 * b;
    const auto t = snls.outputs()[(size_t)b].termination_state;
    EXPECT_TRUE(t == NLSTerminationState::SATISFIED_ABSOLUTE_TOL || t == NLSTerminationState::SATISFIED_RELATIVE_TOL ||
                t == NLSTerminationState::SATISFIED_FIRST_ORDER_TOL);
    EXPECT_NEAR(2.0, std::fabs(v[0]), 5e-5); EXPECT_NEAR(v[0], v[1], 5e-5);
    EXPECT_NEAR(3.0, std::fabs(v[2]), 5e-5); EXPECT_NEAR(v[2], v[3], 5e-5);
    EXPECT_NEAR(0.0, v[4], 5e-5); EXPECT_NEAR(0.0, v[5], 5e-5);
  }

  // custom Retraction (nonlinear.hpp:127): the plain x + alpha dx written by the caller gives the built-in's result; it is really called
  int retractions = 0;
  ConstrainedNonlinearLeastSquares rnls(&problem, [&retractions](std::vector<double>& x, const VectorBlock& dx, double alpha) {
    ++retractions;
    for (size_t i = 0; i < x.size(); ++i) x[i] += dx[(int)i] * alpha;
  });
  const NLSSolverOutputs rout = rnls.Solve(p, {-5.0, -3.0});
  EXPECT_TRUE(retractions >= (int)rout.iterations.size() && retractions > 0);
  EXPECT_TRUE(rout.termination_state == out.termination_state && rout.iterations.size() == out.iterations.size());
  EXPECT_NEAR(nls.variables()[0], rnls.variables()[0], 1e-12); EXPECT_NEAR(nls.variables()[1], rnls.variables()[1], 1e-12);

  // SetUserExitCallback (nonlinear.hpp:157): stop after the first iteration
  int calls = 0;
  nls.SetUserExitCallback([&calls](const NLSIteration& it) { ++calls; return it.iteration < 0; });
  const NLSSolverOutputs uout = nls.Solve(p, {-5.0, -3.0});
  EXPECT_TRUE(uout.termination_state == NLSTerminationState::USER_CALLBACK && uout.iterations.size() == 1 && calls == 1);
  nls.SetUserExitCallback(nullptr);
  bool threw = false;
  try { (void)MakeResidual<2, 2>({0}, rosenbrock); } catch (const default_error&) { threw = true; }   // F_ASSERT_EQ residual.hpp:137
  EXPECT_TRUE(threw);
}

int main() {
  TestLinearInequalityConstraint();
  TestEliminationAllConstraints();
  TestWithInequalitiesAndEqualities();
  TestWithFullyConstrainedEqualities();
  TestErrors();
  TestNlsRosenbrock();
  TestEigenvalueStats();
  TestNlsSphereWithEqualities();
  TestNlsDeviceResiduals();
  TestProblemOfResiduals();
  if (g_fail) { std::printf("%d FAILURES\n", g_fail); return 1; }
  std::printf("facade_test: all tests passed\n");
  return 0;
}
