// mini_opt_hip.hpp -- header-only C++17 facade over the C ABI (include/mini_opt_hip.h) that mirrors mini_opt's QP interface.
//
//   reference                                         facade (namespace mini_opt_hip)
//   ------------------------------------------------  --------------------------------------------------------------
//   LinearInequalityConstraint, Var   qp.hpp:28-92    same names, same members / operators
//   QP                                qp.hpp:104-124  QP (column-major std::vector storage; Eigen overloads if available)
//   QPInteriorPointSolver             qp.hpp:132-295  QPInteriorPointSolver: Setup / Solve / x_block()... / SetVariables /
//                                                     problem(), Params with the reference's defaults; the private step
//                                                     functions the tests reach through `friend` are public test hooks
//   BarrierStrategy, InitialGuessMethod, KKTError, IPIterationOutputs, QPInteriorPointIteration,
//   QPInteriorPointSolverOutputs, QPLagrangeMultipliers   structs.hpp:24-134   same names and fields
//   FailedFactorization               qp.hpp:331-333  same name; thrown for batch == 1 like the reference
//   assert::default_error             assertions.hpp:49-59  mini_opt_hip::default_error (argument / dimension errors)
//
// plus BatchedQPInteriorPointSolver for many problems at once on device pointers (what the hot path is built for).
// The facade does no arithmetic: every number comes from the HIP kernels behind the C ABI.  There is no CPU fallback.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <exception>
#include <functional>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mini_opt_hip.h"

#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define MINI_OPT_HIP_HAS_EIGEN 1
#endif

namespace mini_opt_hip {

struct default_error : public std::runtime_error { using std::runtime_error::runtime_error; };   // assertions.hpp:49-59
struct FailedFactorization : public std::runtime_error { using std::runtime_error::runtime_error; };  // qp.hpp:331-333
struct InfeasibleGuess : public std::runtime_error { using std::runtime_error::runtime_error; };      // qp.hpp:326-328

enum class BarrierStrategy { COMPLEMENTARITY = 0, FIXED_DECREASE, PREDICTOR_CORRECTOR };              // structs.hpp:24-31
enum class InitialGuessMethod { NAIVE = 0, SOLVE_EQUALITY_CONSTRAINED, USER_PROVIDED };               // structs.hpp:34-41
enum class QPInteriorPointTerminationState { SATISFIED_KKT_TOL = 0, MAX_ITERATIONS };                 // structs.hpp:97-102

struct AlphaValues { double primal{1.}; double dual{1.}; };                                          // structs.hpp:45-50
struct IPIterationOutputs {                                                                          // structs.hpp:53-64
  double mu{0.};
  AlphaValues alpha{};
  AlphaValues alpha_probe{std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN()};
  double mu_affine{std::numeric_limits<double>::quiet_NaN()};
};
struct KKTError {                                                                                    // structs.hpp:68-78
  double r_dual{0}, r_comp{0}, r_primal_eq{0}, r_primal_ineq{0};
  double Max() const noexcept { return std::max(std::max(r_dual, r_comp), std::max(r_primal_eq, r_primal_ineq)); }
};
struct QPInteriorPointIteration { KKTError kkt_initial{}; KKTError kkt_final{}; IPIterationOutputs ip_outputs{}; };  // :81-94
struct QPLagrangeMultipliers { double min; double l_infinity; };                                     // structs.hpp:108-113
struct QPInteriorPointSolverOutputs {                                                                // structs.hpp:116-134
  QPInteriorPointTerminationState termination_state{};
  std::vector<QPInteriorPointIteration> iterations;
  std::optional<QPLagrangeMultipliers> lagrange_multipliers;
};

struct LinearInequalityConstraint {                                                                  // qp.hpp:28-70
  int variable;
  double a;
  double b;
  constexpr bool IsFeasible(double x) const noexcept { return a * x + b >= 0.0; }
  double ClampX(double x) const {
    if (a == 0) throw default_error("`a` cannot be zero");
    return a < 0 ? std::min(x, b / -a) : std::max(x, -b / a);
  }
  constexpr LinearInequalityConstraint ShiftTo(double x) const noexcept { return {variable, a, a * x + b}; }
  constexpr LinearInequalityConstraint(int variable, double a, double b) noexcept : variable(variable), a(a), b(b) {}
};
struct Var {                                                                                         // qp.hpp:77-92
  explicit constexpr Var(int variable) noexcept : variable_(variable) {}
  constexpr LinearInequalityConstraint operator<=(double value) const noexcept { return {variable_, -1.0, value}; }
  constexpr LinearInequalityConstraint operator>=(double value) const noexcept { return {variable_, 1.0, -value}; }
 private:
  int variable_;
};

// mini_opt::QP (qp.hpp:104-124).  G (n x n, only the lower triangle is read) and A_eq (k x n) are COLUMN-major, exactly
// the bytes of the reference's Eigen::MatrixXd, so `FromEigen` is a memcpy.
struct QP {
  QP() = default;
  explicit QP(int x_dim) : n(x_dim), G((size_t)x_dim * x_dim, 0.0), c((size_t)x_dim, 0.0) {}
  int n{0};
  int k{0};
  std::vector<double> G, c, A_eq, b_eq;
  std::vector<LinearInequalityConstraint> constraints;
  double& G_at(int i, int j) { return G[(size_t)i + (size_t)j * n]; }
  double& A_at(int q, int j) { return A_eq[(size_t)q + (size_t)j * k]; }
  void ResizeEqualities(int rows) { k = rows; A_eq.assign((size_t)rows * n, 0.0); b_eq.assign((size_t)rows, 0.0); }
#ifdef MINI_OPT_HIP_HAS_EIGEN
  template <typename QPEigen> static QP FromEigen(const QPEigen& q) {  // any struct with the reference's members
    QP out((int)q.G.rows());
    out.G.assign(q.G.data(), q.G.data() + q.G.size());
    out.c.assign(q.c.data(), q.c.data() + q.c.size());
    out.ResizeEqualities((int)q.A_eq.rows());
    out.A_eq.assign(q.A_eq.data(), q.A_eq.data() + q.A_eq.size());
    out.b_eq.assign(q.b_eq.data(), q.b_eq.data() + q.b_eq.size());
    for (const auto& cst : q.constraints) out.constraints.emplace_back(cst.variable, cst.a, cst.b);
    return out;
  }
#endif
};

namespace detail {
inline void check(int rc) {
  if (rc != MO_OK) throw default_error(std::string("mini_opt_hip: ") + mo_last_error());
}
inline void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw default_error(std::string(what) + ": " + hipGetErrorString(e));
}
template <typename T> class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  explicit DeviceBuffer(size_t count) { Resize(count); }
  ~DeviceBuffer() { if (ptr_) (void)hipFree(ptr_); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  void Resize(size_t count) {
    if (count == count_) return;
    if (ptr_) hip_check(hipFree(ptr_), "hipFree");
    ptr_ = nullptr; count_ = count;
    if (count) hip_check(hipMalloc((void**)&ptr_, count * sizeof(T)), "hipMalloc");
  }
  void Upload(const T* src, size_t count) { Resize(count); if (count) hip_check(hipMemcpy(ptr_, src, count * sizeof(T), hipMemcpyHostToDevice), "H2D"); }
  void Download(T* dst, size_t count) const { if (count) hip_check(hipMemcpy(dst, ptr_, count * sizeof(T), hipMemcpyDeviceToHost), "D2H"); }
  T* get() const { return ptr_; }
  size_t size() const { return count_; }
 private:
  T* ptr_{nullptr};
  size_t count_{0};
};
}  // namespace detail

// A view of one block of the state vector (stands in for Eigen::VectorBlock, qp.cc:205-219).
struct VectorBlock {
  double* data; int size;
  double& operator[](int i) const { return data[i]; }
  double* begin() const { return data; }
  double* end() const { return data + size; }
};

// Drop-in for mini_opt::QPInteriorPointSolver (qp.hpp:132-295) for ONE problem; the batch-of-one runs on the GPU.
class QPInteriorPointSolver {
 public:
  struct Params {                                                                                    // qp.hpp:134-164
    double initial_mu{1.0};
    double sigma{0.5};
    double termination_kkt_tol{1.0e-9};
    double termination_complementarity_tol{1.0e-6};
    int max_iterations{10};
    BarrierStrategy barrier_strategy{BarrierStrategy::COMPLEMENTARITY};
    bool decrease_mu_only_on_small_error{false};
    InitialGuessMethod initial_guess_method{InitialGuessMethod::NAIVE};
    bool initialize_mu_with_complementarity{false};
  };

  QPInteriorPointSolver() = default;
  explicit QPInteriorPointSolver(const QP* problem, int device = 0) : device_(device) { Setup(problem); }
  ~QPInteriorPointSolver() { if (plan_) mo_plan_destroy(plan_); }
  QPInteriorPointSolver(const QPInteriorPointSolver&) = delete;
  QPInteriorPointSolver& operator=(const QPInteriorPointSolver&) = delete;

  // qp.cc:20-73
  void Setup(const QP* problem) {
    if (!problem) throw default_error("Must pass a non-null problem");
    p_ = problem;
    const int n = p_->n, k = p_->k, m = (int)p_->constraints.size();
    if ((int)p_->G.size() != n * n) throw default_error("G must be square");
    if ((int)p_->c.size() != n) throw default_error("Dims of G and c must match");
    if ((int)p_->b_eq.size() != k || (int)p_->A_eq.size() != k * n) throw default_error("Rows of A_e and b_e must match");
    for (const auto& c : p_->constraints)
      if (c.variable >= n || c.variable < 0) throw default_error("Constraint index is out of bounds");
    if (plan_ && (n != n_ || k != k_ || m != m_)) { mo_plan_destroy(plan_); plan_ = nullptr; }
    n_ = n; k_ = k; m_ = m;
    if (!plan_) {
      mo_plan_desc d{}; d.n = n; d.k = k; d.m = m; d.m_r = 0; d.dtype = MO_F64; d.device = device_; d.max_batch = 1;
      detail::check(mo_plan_create(&d, &plan_));
    }
    variables_.assign((size_t)V(), 0.0);
    delta_.assign((size_t)V(), 0.0);
    // upload the problem (the solver holds a non-owning pointer like the reference, the device copy is refreshed here)
    G_.Upload(p_->G.data(), p_->G.size());
    c_.Upload(p_->c.data(), p_->c.size());
    A_.Upload(p_->A_eq.data(), p_->A_eq.size());
    b_.Upload(p_->b_eq.data(), p_->b_eq.size());
    std::vector<int32_t> var(m); std::vector<double> ca(m), cb(m);
    for (int i = 0; i < m; ++i) { var[i] = p_->constraints[i].variable; ca[i] = p_->constraints[i].a; cb[i] = p_->constraints[i].b; }
    cv_.Upload(var.data(), m); ca_.Upload(ca.data(), m); cb_.Upload(cb.data(), m);
    vars_dev_.Resize(V()); delta_dev_.Resize(V()); scalar_dev_.Resize(16); status_dev_.Resize(1);
  }

  // qp.cc:100-151
  [[nodiscard]] QPInteriorPointSolverOutputs Solve(const Params& params) {
    if (!p_) throw default_error("Must have a valid problem");
    mo_solve_params sp; mo_default_solve_params(&sp);
    sp.initial_mu = params.initial_mu; sp.sigma = params.sigma; sp.termination_kkt_tol = params.termination_kkt_tol;
    sp.termination_complementarity_tol = params.termination_complementarity_tol; sp.max_iterations = params.max_iterations;
    sp.barrier_strategy = (int)params.barrier_strategy; sp.decrease_mu_only_on_small_error = params.decrease_mu_only_on_small_error;
    sp.initial_guess_method = (int)params.initial_guess_method;
    sp.initialize_mu_with_complementarity = params.initialize_mu_with_complementarity;
    const int iters = params.max_iterations > 0 ? params.max_iterations : 1;
    detail::DeviceBuffer<int32_t> term(1), nit(1);
    detail::DeviceBuffer<double> its((size_t)iters * MO_ITER_RECORD), lag(2);
    vars_dev_.Upload(variables_.data(), variables_.size());
    mo_problem prob = Problem();
    detail::check(mo_qp_solve(plan_, &prob, 1, &sp, vars_dev_.get(), V(), term.get(), nit.get(), its.get(), lag.get(),
                              status_dev_.get(), nullptr));
    detail::hip_check(hipDeviceSynchronize(), "sync");
    int32_t st = 0, t = 0, ni = 0;
    status_dev_.Download(&st, 1); term.Download(&t, 1); nit.Download(&ni, 1);
    vars_dev_.Download(variables_.data(), variables_.size());
    ThrowOnStatus(st);
    std::vector<double> rec((size_t)iters * MO_ITER_RECORD);
    its.Download(rec.data(), rec.size());
    QPInteriorPointSolverOutputs out;
    out.termination_state = (QPInteriorPointTerminationState)t;
    for (int i = 0; i < ni; ++i) {
      const double* r = rec.data() + (size_t)i * MO_ITER_RECORD;
      QPInteriorPointIteration it;
      it.kkt_initial = {r[0], r[1], r[2], r[3]};
      it.kkt_final = {r[4], r[5], r[6], r[7]};
      it.ip_outputs.mu = r[8]; it.ip_outputs.alpha = {r[9], r[10]}; it.ip_outputs.alpha_probe = {r[11], r[12]};
      it.ip_outputs.mu_affine = r[13];
      out.iterations.push_back(it);
    }
    if (k_ > 0) { double l[2]; lag.Download(l, 2); out.lagrange_multipliers = QPLagrangeMultipliers{l[0], l[1]}; }
    return out;
  }

  // Test hook replacing `friend class QPSolverTest` (qp.hpp:293): EvaluateKKTConditions -> ComputeLDLT ->
  // SolveForUpdate(mu) -> ComputeAlpha(tau) on the current state (qp_test.cc:132-134).  Returns delta_ = [dx|ds|dy|dz].
  const std::vector<double>& NewtonStep(double mu, double tau = 0.995, AlphaValues* alpha = nullptr,
                                        bool include_inequalities = true) {
    vars_dev_.Upload(variables_.data(), variables_.size());
    detail::hip_check(hipMemcpy(scalar_dev_.get(), &mu, sizeof(double), hipMemcpyHostToDevice), "H2D");
    mo_problem prob = Problem();
    detail::check(mo_newton_step(plan_, &prob, 1, vars_dev_.get(), V(), scalar_dev_.get(), 0, tau,
                                 include_inequalities ? 0u : MO_STEP_NO_INEQUALITIES, delta_dev_.get(), V(),
                                 scalar_dev_.get() + 2, status_dev_.get(), nullptr));
    detail::hip_check(hipDeviceSynchronize(), "sync");
    int32_t st = 0; status_dev_.Download(&st, 1);
    ThrowOnStatus(st);
    delta_dev_.Download(delta_.data(), delta_.size());
    if (alpha) { double a2[2]; detail::hip_check(hipMemcpy(a2, scalar_dev_.get() + 2, 16, hipMemcpyDeviceToHost), "D2H"); *alpha = {a2[0], a2[1]}; }
    return delta_;
  }

  // qp.cc:205-226
  VectorBlock x_block() { return {variables_.data(), n_}; }
  VectorBlock s_block() { return {variables_.data() + n_, m_}; }
  VectorBlock y_block() { return {variables_.data() + n_ + m_, k_}; }
  VectorBlock z_block() { return {variables_.data() + n_ + m_ + k_, m_}; }
  const std::vector<double>& variables() const noexcept { return variables_; }
  void SetVariables(const std::vector<double>& v) {
    if ((int)v.size() != V()) throw default_error("SetVariables: wrong dimension");
    variables_ = v;
  }
  const QP& problem() const { if (!p_) throw default_error("Cannot call unless initialized"); return *p_; }

 private:
  int V() const { return n_ + 2 * m_ + k_; }
  mo_problem Problem() const {
    mo_problem pr{};
    pr.G = G_.get(); pr.G_stride = 0; pr.G_ld = n_; pr.c = c_.get(); pr.c_stride = 0;
    pr.A_eq = k_ ? A_.get() : nullptr; pr.A_stride = 0; pr.A_ld = k_; pr.b_eq = k_ ? b_.get() : nullptr; pr.b_stride = 0;
    pr.cons_var = m_ ? cv_.get() : nullptr; pr.cons_a = m_ ? ca_.get() : nullptr; pr.cons_b = m_ ? cb_.get() : nullptr;
    pr.cons_stride = 0;
    return pr;
  }
  static void ThrowOnStatus(int32_t st) {
    if (st == MO_STATUS_FACTORIZATION_FAILED)
      throw FailedFactorization("Failed to solve self-adjoint (lower) system. The hessian may not be semi-definite.");  // qp.cc:303-307
    if (st == MO_STATUS_NONPOSITIVE_SLACK) throw default_error("Some slack variables s <= 0");                       // qp.cc:285
    if (st != MO_STATUS_OK) throw default_error(std::string("mini_opt_hip status: ") + mo_status_string(st));
  }
  const QP* p_{nullptr};
  int device_{0};
  int n_{0}, k_{0}, m_{0};
  mo_plan* plan_{nullptr};
  std::vector<double> variables_, delta_;
  detail::DeviceBuffer<double> G_, c_, A_, b_, ca_, cb_, vars_dev_, delta_dev_, scalar_dev_;
  detail::DeviceBuffer<int32_t> cv_, status_dev_;
};

// The batched form the hot path is built for: device pointers in, device pointers out, one launch per step.
class BatchedQPInteriorPointSolver {
 public:
  BatchedQPInteriorPointSolver(int n, int k, int m, int m_r, int64_t max_batch, int device = 0, bool fp32 = false) {
    mo_plan_desc d{}; d.n = n; d.k = k; d.m = m; d.m_r = m_r; d.dtype = fp32 ? MO_F32 : MO_F64; d.device = device; d.max_batch = max_batch;
    detail::check(mo_plan_create(&d, &plan_));
  }
  ~BatchedQPInteriorPointSolver() { if (plan_) mo_plan_destroy(plan_); }
  BatchedQPInteriorPointSolver(const BatchedQPInteriorPointSolver&) = delete;
  BatchedQPInteriorPointSolver& operator=(const BatchedQPInteriorPointSolver&) = delete;
  // EvaluateKKTConditions -> ComputeLDLT -> SolveForUpdate(mu) -> ComputeAlpha(tau) for every problem of the batch.
  void NewtonStep(const mo_problem& prob, int64_t batch, const void* vars, int64_t vars_stride, const void* mu, int64_t mu_stride,
                  double tau, void* delta, int64_t delta_stride, void* alpha, int32_t* status, hipStream_t stream = nullptr) {
    detail::check(mo_newton_step(plan_, &prob, batch, vars, vars_stride, mu, mu_stride, tau, 0, delta, delta_stride, alpha, status, stream));
  }
  void Solve(const mo_problem& prob, int64_t batch, const mo_solve_params& params, void* vars, int64_t vars_stride, int32_t* termination,
             int32_t* num_iterations, void* iterations, void* lagrange, int32_t* status, hipStream_t stream = nullptr) {
    detail::check(mo_qp_solve(plan_, &prob, batch, &params, vars, vars_stride, termination, num_iterations, iterations, lagrange, status, stream));
  }
  const char* StepKernel(const mo_problem& prob) const { return mo_plan_step_kernel(plan_, &prob); }
  mo_plan* plan() const { return plan_; }
 private:
  mo_plan* plan_{nullptr};
};

// ---------------------------------------------------------------------------------------------------------------------
// ConstrainedNonlinearLeastSquares (nonlinear.hpp:127-230) for a batch of problems of one structure.  The reference's
// residuals are host functors (residual.hpp:28-143); this facade keeps them on the host -- `HostResiduals` is called with the
// evaluation points of ALL problems and fills dense stacks -- and ships the stacks to the device, where the whole SQP
// iteration (QP, penalty, line search, lambda state machine) runs behind mo_nls_solve.  Callers with device-side residual
// kernels bind mo_nls_solve directly and skip the PCIe hops.
enum class NLSTerminationState { MAX_ITERATIONS = 0, SATISFIED_ABSOLUTE_TOL, SATISFIED_RELATIVE_TOL, SATISFIED_FIRST_ORDER_TOL,
                                 MAX_LAMBDA, QP_INDEFINITE, USER_CALLBACK, QP_FAILURE };                  // structs.hpp:233-248
enum class LineSearchStrategy { ARMIJO_BACKTRACK = 0, POLYNOMIAL_APPROXIMATION = 1 };                    // structs.hpp:148-153

class BatchedConstrainedNonlinearLeastSquares {
 public:
  struct Params {                                                                                        // nonlinear.hpp:64-124
    int max_iterations{10};
    int max_qp_iterations{10};
    double termination_kkt_tolerance{1.0e-6};
    double absolute_exit_tol{1.0e-12};
    double relative_exit_tol{1.0e-5};
    double absolute_first_derivative_tol{1.0e-6};
    int max_line_search_iterations{2};
    LineSearchStrategy line_search_strategy{LineSearchStrategy::POLYNOMIAL_APPROXIMATION};
    double armijo_search_tau{0.8};
    double equality_penalty_initial{1.0};
    double equality_penalty_scale_factor{1.01};
    double equality_penalty_rho{0.1};
    double lambda_initial{0.0};
    double lambda_failure_init{1.0e-2};
    double lambda_decrease_on_success{0.1};
    double lambda_decrease_on_restore{0.8};
    double max_lambda{1.};
    double min_lambda{0.};
  };
  // x: [batch][n] evaluation points.  r: [batch][m_r], J: [batch][m_r][n] row-major (NULL when only errors are wanted),
  // r_eq: [batch][k], J_eq: [batch][k][n] row-major (both NULL when k == 0).
  using HostResiduals = std::function<void(const double* x, int64_t batch, double* r, double* J, double* r_eq, double* J_eq)>;

  BatchedConstrainedNonlinearLeastSquares(int n, int m_r, int k, std::vector<LinearInequalityConstraint> inequality_constraints,
                                          HostResiduals residuals, int64_t batch, int device = 0)
      : n_(n), m_r_(m_r), k_(k), m_((int)inequality_constraints.size()), batch_(batch), residuals_(std::move(residuals)) {
    mo_plan_desc d{}; d.n = n; d.k = k; d.m = m_; d.m_r = m_r; d.dtype = MO_F64; d.device = device; d.max_batch = batch;
    detail::check(mo_plan_create(&d, &plan_));
    std::vector<int32_t> cv; std::vector<double> ca, cb;
    for (const auto& c : inequality_constraints) {
      if (c.variable < 0 || c.variable >= n) throw default_error("constraint index out of range");   // F_ASSERT_LT qp.hpp:63
      cv.push_back(c.variable); ca.push_back(c.a); cb.push_back(c.b);
    }
    cv_.Upload(cv.data(), cv.size()); ca_.Upload(ca.data(), ca.size()); cb_.Upload(cb.data(), cb.size());
    const size_t B = (size_t)batch;
    vars_.Resize(B * n); cand_.Resize(B * n); J_.Resize(B * m_r * n); r_.Resize(B * m_r); r_cand_.Resize(B * m_r);
    Jeq_.Resize(B * k * n); req_.Resize(B * k); req_cand_.Resize(B * k);
    h_x_.resize(B * n); h_J_.resize(B * m_r * n); h_r_.resize(B * m_r); h_Jeq_.resize(B * k * n); h_JeqT_.resize(B * k * n); h_req_.resize(B * k);
  }
  ~BatchedConstrainedNonlinearLeastSquares() { if (plan_) mo_plan_destroy(plan_); }
  BatchedConstrainedNonlinearLeastSquares(const BatchedConstrainedNonlinearLeastSquares&) = delete;
  BatchedConstrainedNonlinearLeastSquares& operator=(const BatchedConstrainedNonlinearLeastSquares&) = delete;

  // Solve(params, variables), nonlinear.cc:75-158: variables is [batch][n]; returns the termination state of every problem.
  std::vector<NLSTerminationState> Solve(const Params& p, const std::vector<double>& variables) {
    if ((int64_t)variables.size() != batch_ * n_) throw default_error("variables must be batch x n");
    vars_.Upload(variables.data(), variables.size());
    mo_nls_params sp; mo_default_nls_params(&sp);
    sp.max_iterations = p.max_iterations; sp.max_qp_iterations = p.max_qp_iterations;
    sp.termination_kkt_tolerance = p.termination_kkt_tolerance; sp.absolute_exit_tol = p.absolute_exit_tol;
    sp.relative_exit_tol = p.relative_exit_tol; sp.absolute_first_derivative_tol = p.absolute_first_derivative_tol;
    sp.max_line_search_iterations = p.max_line_search_iterations; sp.line_search_strategy = (int32_t)p.line_search_strategy;
    sp.armijo_search_tau = p.armijo_search_tau; sp.equality_penalty_initial = p.equality_penalty_initial;
    sp.equality_penalty_scale_factor = p.equality_penalty_scale_factor; sp.equality_penalty_rho = p.equality_penalty_rho;
    sp.lambda_initial = p.lambda_initial; sp.lambda_failure_init = p.lambda_failure_init;
    sp.lambda_decrease_on_success = p.lambda_decrease_on_success; sp.lambda_decrease_on_restore = p.lambda_decrease_on_restore;
    sp.max_lambda = p.max_lambda; sp.min_lambda = p.min_lambda;
    mo_nls_problem np{};
    np.vars = vars_.get(); np.vars_stride = n_; np.candidate = cand_.get(); np.candidate_stride = n_;
    np.J = J_.get(); np.J_stride = (int64_t)m_r_ * n_; np.J_ld = n_; np.J_layout = MO_ROW_MAJOR; np.r = r_.get(); np.r_stride = m_r_;
    np.r_cand = r_cand_.get(); np.r_cand_stride = m_r_;
    if (k_ > 0) {
      np.J_eq = Jeq_.get(); np.J_eq_stride = (int64_t)k_ * n_; np.J_eq_ld = k_; np.r_eq = req_.get(); np.r_eq_stride = k_;
      np.r_eq_cand = req_cand_.get(); np.r_eq_cand_stride = k_;
    }
    if (m_ > 0) { np.cons_var = cv_.get(); np.cons_a = ca_.get(); np.cons_b = cb_.get(); np.cons_stride = 0; }
    detail::DeviceBuffer<int32_t> term((size_t)batch_), nit((size_t)batch_);
    callback_error_ = nullptr;
    const int rc = mo_nls_solve(plan_, &np, batch_, &sp, &BatchedConstrainedNonlinearLeastSquares::Eval, this, term.get(), nit.get(),
                                nullptr, nullptr, nullptr);
    if (callback_error_) std::rethrow_exception(callback_error_);
    detail::check(rc);
    variables_.resize((size_t)batch_ * n_); vars_.Download(variables_.data(), variables_.size());
    num_iterations_.resize((size_t)batch_); nit.Download(num_iterations_.data(), num_iterations_.size());
    std::vector<int32_t> t((size_t)batch_); term.Download(t.data(), t.size());
    std::vector<NLSTerminationState> out;
    for (int32_t v : t) out.push_back((NLSTerminationState)v);
    return out;
  }
  const std::vector<double>& variables() const { return variables_; }          // [batch][n]
  const std::vector<int32_t>& num_iterations() const { return num_iterations_; }

 private:
  static int Eval(void* user, int32_t what, void* stream) {
    auto* self = static_cast<BatchedConstrainedNonlinearLeastSquares*>(user);
    try {
      const bool lin = what == MO_NLS_EVAL_LINEARIZE;
      const size_t B = (size_t)self->batch_, n = (size_t)self->n_, k = (size_t)self->k_;
      (void)hipStreamSynchronize((hipStream_t)stream);  // the evaluation point is produced by device work on this stream
      (lin ? self->vars_ : self->cand_).Download(self->h_x_.data(), B * n);
      self->residuals_(self->h_x_.data(), self->batch_, self->h_r_.data(), lin ? self->h_J_.data() : nullptr,
                       k ? self->h_req_.data() : nullptr, (lin && k) ? self->h_Jeq_.data() : nullptr);
      (lin ? self->r_ : self->r_cand_).Upload(self->h_r_.data(), self->h_r_.size());
      if (k) (lin ? self->req_ : self->req_cand_).Upload(self->h_req_.data(), self->h_req_.size());
      if (lin) {
        self->J_.Upload(self->h_J_.data(), self->h_J_.size());
        if (k) {  // QP::A_eq is k x n column-major (Eigen default): transpose the row-major stack
          for (size_t p = 0; p < B; ++p)
            for (size_t i = 0; i < k; ++i)
              for (size_t j = 0; j < n; ++j) self->h_JeqT_[p * k * n + j * k + i] = self->h_Jeq_[p * k * n + i * n + j];
          self->Jeq_.Upload(self->h_JeqT_.data(), self->h_JeqT_.size());
        }
      }
      return 0;
    } catch (...) {
      self->callback_error_ = std::current_exception();
      return 1;
    }
  }
  int n_, m_r_, k_, m_;
  int64_t batch_;
  HostResiduals residuals_;
  mo_plan* plan_{nullptr};
  detail::DeviceBuffer<double> vars_, cand_, J_, r_, r_cand_, Jeq_, req_, req_cand_, ca_, cb_;
  detail::DeviceBuffer<int32_t> cv_;
  std::vector<double> h_x_, h_J_, h_r_, h_Jeq_, h_JeqT_, h_req_, variables_;
  std::vector<int32_t> num_iterations_;
  std::exception_ptr callback_error_{nullptr};
};

}  // namespace mini_opt_hip
