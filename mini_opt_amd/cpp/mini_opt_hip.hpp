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
#include <algorithm>
#include <array>
#include <exception>
#include <memory>
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
struct QPEigenvalues { double min; double max; double abs_min; };                                    // structs.hpp:267-275
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
  // QP::ComputeEigenvalueStats (qp.hpp:122-123, qp.cc:12-16): the eigenvalue summary of G (its lower triangle, as SelfAdjointEigenSolver reads
  // it), computed on the device through mo_qp_eigenvalue_stats.  Defined below (needs the device helpers).
  inline QPEigenvalues ComputeEigenvalueStats(int device = 0) const;
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

inline QPEigenvalues QP::ComputeEigenvalueStats(int device) const {
  mo_plan_desc d{}; d.n = n; d.k = 0; d.m = 0; d.m_r = 0; d.dtype = MO_F64; d.device = device; d.max_batch = 1;
  mo_plan* plan = nullptr;
  detail::check(mo_plan_create(&d, &plan));
  struct Guard { mo_plan* p; ~Guard() { mo_plan_destroy(p); } } guard{plan};
  detail::DeviceBuffer<double> Gd, cd, out(3);
  Gd.Upload(G.data(), G.size()); cd.Upload(c.data(), c.size());
  mo_problem pr{};
  pr.G = Gd.get(); pr.G_stride = (int64_t)n * n; pr.G_ld = n; pr.c = cd.get(); pr.c_stride = n;
  detail::check(mo_qp_eigenvalue_stats(plan, &pr, 1, out.get(), nullptr));
  detail::hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
  double h[3];
  out.Download(h, 3);
  return QPEigenvalues{h[0], h[1], h[2]};
}

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

enum class OptimizerState { NOMINAL = 0, ATTEMPTING_RESTORE_LM = 1 };                                    // structs.hpp:156-165
enum class StepSizeSelectionResult { SUCCESS = 0, MAX_ITERATIONS, FIRST_ORDER_SATISFIED, POSITIVE_DERIVATIVE, FAILURE_NON_FINITE_COST,
                                     FAILURE_INVALID_ALPHA };                                            // structs.hpp:215-228
struct Errors {                                                                                          // structs.hpp:169-186
  double f{0.}, equality{0.};
  double Total(double penalty) const noexcept { return f + penalty * equality; }
  double LInfinity() const noexcept { return std::max(f, equality); }
};
struct DirectionalDerivatives { double d_f{0.}, d_equality{0.}; double Total(double penalty) const noexcept { return d_f + penalty * d_equality; } };  // :189-203
struct LineSearchStep { double alpha; Errors errors; };                                                  // structs.hpp:206-213
// NLSIteration (structs.hpp:277-330).  qp_outputs is summarised: termination state and iteration count of the interior-point QP (or the
// null-space solver's state for equality-only problems, nonlinear.cc:249-258) and the QP's per-problem status word.
struct NLSIteration {
  int iteration{0};
  OptimizerState optimizer_state{OptimizerState::NOMINAL};
  double lambda{0.};
  Errors errors_initial{};
  int qp_termination_state{0}, qp_num_iterations{0}, qp_status{0};
  std::optional<QPEigenvalues> qp_eigenvalues;   // when Params::log_qp_eigenvalues (structs.hpp:306-307, nonlinear.cc:138)
  DirectionalDerivatives directional_derivatives{};
  double penalty{0.};
  StepSizeSelectionResult step_result{StepSizeSelectionResult::SUCCESS};
  std::vector<LineSearchStep> line_search_steps;
};
struct NLSSolverOutputs {                                                                                // structs.hpp:332-347
  NLSTerminationState termination_state{NLSTerminationState::MAX_ITERATIONS};
  std::vector<NLSIteration> iterations;
  int NumLineSearchSteps() const { int c = 0; for (const auto& it : iterations) c += (int)it.line_search_steps.size(); return c; }
  int NumQPIterations() const { int c = 0; for (const auto& it : iterations) c += it.qp_num_iterations; return c; }
};

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
    bool log_qp_eigenvalues{false};   // nonlinear.hpp:122-123
  };
  // x: [batch][n] evaluation points.  r: [batch][m_r], J: [batch][m_r][n] row-major (NULL when only errors are wanted),
  // r_eq: [batch][k], J_eq: [batch][k][n] row-major (both NULL when k == 0).
  using HostResiduals = std::function<void(const double* x, int64_t batch, double* r, double* J, double* r_eq, double* J_eq)>;
  // The reference's Retraction (nonlinear.hpp:127; applied in RetractCandidateVars, nonlinear.cc:160-168): x enters as the current
  // variables of one problem and leaves as the trial point for step dx and step length alpha.
  using Retraction = std::function<void(std::vector<double>& x, const VectorBlock& dx, double alpha)>;
  // SetUserExitCallback (nonlinear.hpp:157): called per problem after every outer iteration it took part in; false stops that problem
  // (USER_CALLBACK) unless the iteration terminated it anyway (nonlinear.cc:142-149).
  using UserExitCallback = std::function<bool(int64_t problem, const NLSIteration& iteration)>;
  void SetRetraction(Retraction r) { retraction_ = std::move(r); }
  void SetUserExitCallback(UserExitCallback cb) { user_exit_ = std::move(cb); }

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
    sp.retraction = retraction_ ? MO_RETRACT_CALLBACK : MO_RETRACT_EUCLIDEAN;
    sp.log_qp_eigenvalues = p.log_qp_eigenvalues ? 1 : 0;
    mo_nls_problem np{};
    np.vars = vars_.get(); np.vars_stride = n_; np.candidate = cand_.get(); np.candidate_stride = n_;
    np.J = J_.get(); np.J_stride = (int64_t)m_r_ * n_; np.J_ld = n_; np.J_layout = MO_ROW_MAJOR; np.r = r_.get(); np.r_stride = m_r_;
    np.r_cand = r_cand_.get(); np.r_cand_stride = m_r_;
    if (k_ > 0) {
      np.J_eq = Jeq_.get(); np.J_eq_stride = (int64_t)k_ * n_; np.J_eq_ld = k_; np.r_eq = req_.get(); np.r_eq_stride = k_;
      np.r_eq_cand = req_cand_.get(); np.r_eq_cand_stride = k_;
    }
    if (m_ > 0) { np.cons_var = cv_.get(); np.cons_a = ca_.get(); np.cons_b = cb_.get(); np.cons_stride = 0; }
    if (retraction_) { step_.Resize((size_t)batch_ * n_); step_alpha_.Resize((size_t)batch_); np.step = step_.get(); np.step_stride = n_; np.step_alpha = step_alpha_.get(); }
    if (user_exit_) { exit_flags_.Resize((size_t)batch_); np.user_exit = exit_flags_.get(); }
    detail::DeviceBuffer<int32_t> term((size_t)batch_), nit((size_t)batch_);
    rec_ = MO_NLS_ITER_RECORD(p.max_line_search_iterations);
    max_iterations_ = p.max_iterations > 0 ? p.max_iterations : 1;
    records_.Resize((size_t)batch_ * max_iterations_ * rec_);
    {  // records of iterations a problem never ran stay NaN
      std::vector<double> nanfill((size_t)batch_ * max_iterations_ * rec_, std::numeric_limits<double>::quiet_NaN());
      records_.Upload(nanfill.data(), nanfill.size());
    }
    detail::DeviceBuffer<double> eig;
    if (p.log_qp_eigenvalues) {   // [max_iterations][batch][3], NaN where a problem never ran the iteration
      std::vector<double> nanfill((size_t)batch_ * max_iterations_ * 3, std::numeric_limits<double>::quiet_NaN());
      eig.Upload(nanfill.data(), nanfill.size());
      np.qp_eigenvalues = eig.get();
    }
    iterations_done_ = 0;
    term_dev_ = term.get(); nit_dev_ = nit.get();
    callback_error_ = nullptr;
    const int rc = mo_nls_solve(plan_, &np, batch_, &sp, &BatchedConstrainedNonlinearLeastSquares::Eval, this, term.get(), nit.get(),
                                records_.get(), nullptr, nullptr);
    if (callback_error_) std::rethrow_exception(callback_error_);
    detail::check(rc);
    variables_.resize((size_t)batch_ * n_); vars_.Download(variables_.data(), variables_.size());
    num_iterations_.resize((size_t)batch_); nit.Download(num_iterations_.data(), num_iterations_.size());
    std::vector<int32_t> t((size_t)batch_); term.Download(t.data(), t.size());
    std::vector<double> rec((size_t)batch_ * max_iterations_ * rec_); records_.Download(rec.data(), rec.size());
    std::vector<double> eigh;
    if (p.log_qp_eigenvalues) { eigh.resize((size_t)batch_ * max_iterations_ * 3); eig.Download(eigh.data(), eigh.size()); }
    outputs_.assign((size_t)batch_, NLSSolverOutputs{});
    std::vector<NLSTerminationState> out;
    for (int64_t b = 0; b < batch_; ++b) {
      out.push_back((NLSTerminationState)t[(size_t)b]);
      outputs_[(size_t)b].termination_state = out.back();
      for (int it = 0; it < num_iterations_[(size_t)b] && it < max_iterations_; ++it) {
        const double* r = rec.data() + ((size_t)b * max_iterations_ + it) * rec_;
        if (std::isnan(r[1])) break;  // QP_INDEFINITE ends a problem without logging the iteration (nonlinear.cc:103-105)
        outputs_[(size_t)b].iterations.push_back(Record(r, it));
        if (p.log_qp_eigenvalues) {
          const double* ev = eigh.data() + ((size_t)it * (size_t)batch_ + (size_t)b) * 3;
          if (!std::isnan(ev[0])) outputs_[(size_t)b].iterations.back().qp_eigenvalues = QPEigenvalues{ev[0], ev[1], ev[2]};
        }
      }
    }
    return out;
  }
  // NLSSolverOutputs (structs.hpp:332-347) of every problem of the last Solve.
  const std::vector<NLSSolverOutputs>& outputs() const { return outputs_; }
  const std::vector<double>& variables() const { return variables_; }          // [batch][n]
  const std::vector<int32_t>& num_iterations() const { return num_iterations_; }

 private:
  static int Eval(void* user, int32_t what, void* stream) {
    auto* self = static_cast<BatchedConstrainedNonlinearLeastSquares*>(user);
    try {
      if (what == MO_NLS_EVAL_RETRACT) { self->Retract((hipStream_t)stream); return 0; }
      if (what == MO_NLS_EVAL_ITERATION_DONE) { self->IterationDone((hipStream_t)stream); return 0; }
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
  NLSIteration Record(const double* r, int it) const {   // MO_NLS_ITER_RECORD layout, include/mini_opt_hip.h
    NLSIteration o;
    o.iteration = it; o.optimizer_state = (OptimizerState)(int)r[0]; o.lambda = r[1]; o.errors_initial = {r[2], r[3]};
    o.directional_derivatives = {r[4], r[5]}; o.penalty = r[6];
    o.step_result = std::isnan(r[7]) ? StepSizeSelectionResult::SUCCESS : (StepSizeSelectionResult)(int)r[7];
    const int ns = std::isnan(r[8]) ? 0 : (int)r[8];
    o.qp_termination_state = std::isnan(r[9]) ? 0 : (int)r[9]; o.qp_num_iterations = std::isnan(r[10]) ? 0 : (int)r[10];
    o.qp_status = std::isnan(r[11]) ? 0 : (int)r[11];
    for (int s = 0; s < ns; ++s) o.line_search_steps.push_back({r[MO_NLS_ITER_HEADER + 3 * s], {r[MO_NLS_ITER_HEADER + 3 * s + 1], r[MO_NLS_ITER_HEADER + 3 * s + 2]}});
    return o;
  }
  void Retract(hipStream_t stream) {   // the caller's Retraction on the host, problem by problem (nonlinear.cc:160-168)
    const size_t B = (size_t)batch_, n = (size_t)n_;
    (void)hipStreamSynchronize(stream);
    h_x_.resize(B * n); h_step_.resize(B * n); h_alpha_.resize(B);
    vars_.Download(h_x_.data(), B * n); step_.Download(h_step_.data(), B * n); step_alpha_.Download(h_alpha_.data(), B);
    std::vector<double> x(n);
    for (size_t b = 0; b < B; ++b) {
      x.assign(h_x_.begin() + b * n, h_x_.begin() + (b + 1) * n);
      retraction_(x, VectorBlock{h_step_.data() + b * n, (int)n}, h_alpha_[b]);
      std::copy(x.begin(), x.end(), h_x_.begin() + b * n);
    }
    cand_.Upload(h_x_.data(), B * n);
  }
  void IterationDone(hipStream_t stream) {   // SetUserExitCallback, nonlinear.cc:142-149
    const size_t B = (size_t)batch_;
    (void)hipStreamSynchronize(stream);
    std::vector<int32_t> nit(B), flags(B, 0);
    detail::hip_check(hipMemcpy(nit.data(), nit_dev_, B * sizeof(int32_t), hipMemcpyDeviceToHost), "D2H");
    std::vector<double> rec(B * (size_t)max_iterations_ * rec_);
    records_.Download(rec.data(), rec.size());
    for (size_t b = 0; b < B; ++b) {
      if (nit[b] != iterations_done_ + 1) continue;   // this problem did not take part in the iteration that just ended
      const double* r = rec.data() + (b * max_iterations_ + iterations_done_) * rec_;
      if (std::isnan(r[1])) continue;
      if (!user_exit_((int64_t)b, Record(r, iterations_done_))) flags[b] = 1;
    }
    exit_flags_.Upload(flags.data(), B);
    ++iterations_done_;
  }
  int n_, m_r_, k_, m_;
  int64_t batch_;
  HostResiduals residuals_;
  Retraction retraction_;
  UserExitCallback user_exit_;
  detail::DeviceBuffer<double> step_, step_alpha_, records_;
  detail::DeviceBuffer<int32_t> exit_flags_;
  std::vector<double> h_step_, h_alpha_;
  std::vector<NLSSolverOutputs> outputs_;
  int rec_{0}, max_iterations_{1}, iterations_done_{0};
  int32_t* term_dev_{nullptr}; int32_t* nit_dev_{nullptr};
  mo_plan* plan_{nullptr};
  detail::DeviceBuffer<double> vars_, cand_, J_, r_, r_cand_, Jeq_, req_, req_cand_, ca_, cb_;
  detail::DeviceBuffer<int32_t> cv_;
  std::vector<double> h_x_, h_J_, h_r_, h_Jeq_, h_JeqT_, h_req_, variables_;
  std::vector<int32_t> num_iterations_;
  std::exception_ptr callback_error_{nullptr};
};

// ---------------------------------------------------------------------------------------------------------------------
// mini_opt::Residual / MakeResidual / Problem / ConstrainedNonlinearLeastSquares (residual.hpp:28-143, nonlinear.hpp:33-52, 127-230)
// as a drop-in: the same functor shape -- ResidualType(const ParamType&, JacobianType* J_or_null) on the residual's OWN parameters,
// picked out of the full vector by an index list -- evaluated on the host like the reference's, with the per-residual Jacobians
// scattered into the dense stacks the device consumes (UpdateJacobian's J_out.col(index[l]) = J.col(l), residual.hpp:230-250; summing
// J^T J over residuals, residual.hpp:206-224, is the J^T J of the stacked rows).  With Eigen on the include path ParamType / ResidualType
// / JacobianType ARE the reference's Eigen types (functors compile unchanged; not exercised in this image, which has no Eigen);
// without it they are the small fixed-size types below.
constexpr int Dynamic = -1;
#ifdef MINI_OPT_HIP_HAS_EIGEN
template <int N> using VectorN = Eigen::Matrix<double, N, 1>;
template <int R, int P> using JacobianRP = Eigen::Matrix<double, R, P>;
#else
template <int N> struct VectorN {
  std::array<double, (size_t)N> v{};
  double& operator[](int i) { return v[(size_t)i]; }
  double operator[](int i) const { return v[(size_t)i]; }
  double& operator()(int i) { return v[(size_t)i]; }
  double operator()(int i) const { return v[(size_t)i]; }
  int rows() const { return N; }
  int size() const { return N; }
  const double* data() const { return v.data(); }
  double* data() { return v.data(); }
};
template <> struct VectorN<Dynamic> {
  std::vector<double> v;
  VectorN() = default;
  explicit VectorN(int n) : v((size_t)n, 0.0) {}
  double& operator[](int i) { return v[(size_t)i]; }
  double operator[](int i) const { return v[(size_t)i]; }
  double& operator()(int i) { return v[(size_t)i]; }
  double operator()(int i) const { return v[(size_t)i]; }
  int rows() const { return (int)v.size(); }
  int size() const { return (int)v.size(); }
  const double* data() const { return v.data(); }
  double* data() { return v.data(); }
};
template <int R, int P> struct JacobianRP {   // column-major like Eigen's fixed-size matrices
  std::vector<double> a = std::vector<double>((size_t)R * (size_t)(P > 0 ? P : 0), 0.0);
  int cols_{P > 0 ? P : 0};
  void resize(int rows, int cols) { (void)rows; cols_ = cols; a.assign((size_t)R * (size_t)cols, 0.0); }
  double& operator()(int r, int c) { return a[(size_t)r + (size_t)c * R]; }
  double operator()(int r, int c) const { return a[(size_t)r + (size_t)c * R]; }
  int rows() const { return R; }
  int cols() const { return cols_; }
  void setZero() { std::fill(a.begin(), a.end(), 0.0); }
};
#endif

class Residual final {   // residual.hpp:28-117
 public:
  int Dimension() const { return impl_->Dimension(); }
  // h(x) into b_out[0 .. Dimension())
  void ErrorVector(const double* params, double* b_out) const { impl_->Evaluate(params, b_out, nullptr, 0); }
  // rows of the dense stack: J_out is Dimension() x n ROW-major with leading dimension ld (zeroed by the caller), b_out = h(x)
  void UpdateJacobian(const double* params, double* J_out, int ld, double* b_out) const { impl_->Evaluate(params, b_out, J_out, ld); }
  double QuadraticError(const std::vector<double>& params) const {   // 0.5 |h|^2, residual.cc
    std::vector<double> b((size_t)Dimension());
    ErrorVector(params.data(), b.data());
    double s = 0; for (double v : b) s += v * v;
    return 0.5 * s;
  }
  struct Concept {
    virtual ~Concept() = default;
    virtual int Dimension() const noexcept = 0;
    virtual void Evaluate(const double* params, double* b_out, double* J_out, int ld) const = 0;
  };
  explicit Residual(std::unique_ptr<Concept> impl) noexcept : impl_(std::move(impl)) {}
 private:
  std::unique_ptr<Concept> impl_;
};

namespace detail {
template <int R, int P, typename F> class ResidualModel final : public Residual::Concept {   // Residual::Model, residual.hpp:88-111
 public:
  ResidualModel(std::vector<int> index, F func) : index_(std::move(index)), func_(std::move(func)) {}
  int Dimension() const noexcept override { return R; }
  void Evaluate(const double* params, double* b_out, double* J_out, int ld) const override {
    const int np = (int)index_.size();
    auto local = MakeParams(np);
    for (int l = 0; l < np; ++l) local[l] = params[index_[(size_t)l]];   // GatherParams, residual.hpp:150-163
    if (!J_out) {
      const auto r = func_(local, static_cast<JacobianRP<R, P>*>(nullptr));
      for (int i = 0; i < R; ++i) b_out[i] = r[i];
      return;
    }
    JacobianRP<R, P> J;
    if constexpr (P == Dynamic) J.resize(R, np);
    const auto r = func_(local, &J);
    for (int i = 0; i < R; ++i) {
      b_out[i] = r[i];
      for (int l = 0; l < np; ++l) J_out[(size_t)i * ld + index_[(size_t)l]] = J(i, l);   // J_out.col(index[l]) = J.col(l), residual.hpp:240-246
    }
  }
 private:
  static VectorN<P> MakeParams(int np) { if constexpr (P == Dynamic) return VectorN<P>(np); else { (void)np; return VectorN<P>(); } }
  std::vector<int> index_;
  F func_;
};
}  // namespace detail

// MakeResidual<R, P>({indices...}, functor), residual.hpp:119-143.
template <int R, int P, typename F> Residual MakeResidual(std::initializer_list<int> index, F&& func) {
  using FuncType = std::remove_const_t<std::remove_reference_t<F>>;
  if (P != Dynamic && (int)index.size() != P) throw default_error("MakeResidual: index list does not have P entries");
  return Residual(std::make_unique<detail::ResidualModel<R, P, FuncType>>(std::vector<int>(index), std::forward<F>(func)));
}

struct Problem {   // nonlinear.hpp:33-52
  int dimension{0};
  std::vector<Residual> costs;
  std::vector<LinearInequalityConstraint> inequality_constraints;
  std::vector<Residual> equality_constraints;
};

// Drop-in for mini_opt::ConstrainedNonlinearLeastSquares (nonlinear.hpp:127-230) on ONE problem given as Residuals -- or on `batch`
// initial guesses of it at once (Solve takes batch x dimension values), which is what the device is for.
class ConstrainedNonlinearLeastSquares {
 public:
  using Params = BatchedConstrainedNonlinearLeastSquares::Params;
  using Retraction = BatchedConstrainedNonlinearLeastSquares::Retraction;
  explicit ConstrainedNonlinearLeastSquares(const Problem* problem, Retraction retraction = nullptr, int64_t batch = 1, int device = 0) : p_(problem) {
    if (!p_) throw default_error("Must have a valid problem");   // F_ASSERT nonlinear.cc:21
    int m_r = 0, k = 0;
    for (const auto& c : p_->costs) m_r += c.Dimension();
    for (const auto& c : p_->equality_constraints) k += c.Dimension();
    const int n = p_->dimension;
    auto eval = [this, n, m_r, k](const double* x, int64_t B, double* r, double* J, double* r_eq, double* J_eq) {
      for (int64_t b = 0; b < B; ++b) {
        const double* xb = x + (size_t)b * n;
        double* rb = r + (size_t)b * m_r; double* Jb = J ? J + (size_t)b * m_r * n : nullptr;
        if (Jb) std::fill(Jb, Jb + (size_t)m_r * n, 0.0);
        int row = 0;
        for (const auto& c : p_->costs) {
          if (Jb) c.UpdateJacobian(xb, Jb + (size_t)row * n, n, rb + row); else c.ErrorVector(xb, rb + row);
          row += c.Dimension();
        }
        if (!k) continue;
        double* qb = r_eq + (size_t)b * k; double* Qb = J_eq ? J_eq + (size_t)b * k * n : nullptr;
        if (Qb) std::fill(Qb, Qb + (size_t)k * n, 0.0);
        row = 0;
        for (const auto& c : p_->equality_constraints) {
          if (Qb) c.UpdateJacobian(xb, Qb + (size_t)row * n, n, qb + row); else c.ErrorVector(xb, qb + row);
          row += c.Dimension();
        }
      }
    };
    impl_ = std::make_unique<BatchedConstrainedNonlinearLeastSquares>(n, m_r, k, p_->inequality_constraints, eval, batch, device);
    if (retraction) impl_->SetRetraction(std::move(retraction));
  }
  // Solve(params, variables), nonlinear.cc:75-158; for batch > 1 the outputs of problem b are outputs()[b]
  NLSSolverOutputs Solve(const Params& params, const std::vector<double>& variables) {
    (void)impl_->Solve(params, variables);
    return impl_->outputs().front();
  }
  const std::vector<NLSSolverOutputs>& outputs() const { return impl_->outputs(); }
  const std::vector<double>& variables() const { return impl_->variables(); }
  void SetUserExitCallback(std::function<bool(const NLSIteration&)> cb) {   // nonlinear.hpp:157
    if (!cb) { impl_->SetUserExitCallback(nullptr); return; }
    impl_->SetUserExitCallback([cb](int64_t, const NLSIteration& it) { return cb(it); });
  }
 private:
  const Problem* p_;
  std::unique_ptr<BatchedConstrainedNonlinearLeastSquares> impl_;
};

}  // namespace mini_opt_hip
