/*
 * mini_opt_hip.h -- C ABI of the MI355X-native batched interior-point Newton-step solver.
 *
 * This is the drop-in boundary for mini_opt's dense KKT hot path (source/qp.cc).  The reference has no FFI:
 * its boundary is the C++ class API (include/mini_opt/qp.hpp:132-205, residual.hpp:28-117,
 * nonlinear.hpp:33-52,127-157).  Each entry point below names the reference member function(s) it replaces;
 * INTEGRATION.md shows the binding a mini_opt maintainer would add inside QPInteriorPointSolver /
 * ConstrainedNonlinearLeastSquares, and mini_opt_amd/cpp/ holds the C++ facade that mirrors those classes.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no exceptions across the boundary.  Every call returns an int:
 *    MO_OK (0) or a negative MO_ERR_* (the reference's F_ASSERT / assert::default_error cases,
 *    assertions.hpp:68-73).  Per-problem numeric failures never abort the batch: they are written to the
 *    caller's `status[batch]` array (MO_STATUS_*), mirroring the reference's two exceptions.
 *  - All data pointers are DEVICE pointers on the plan's GPU (caller-owned).  The plan owns only scratch.
 *  - A "batch" is `batch` independent QPs of identical dimensions (n, k, m, m_r).  Problem p of tensor X lives at
 *    X + p * X_stride (strides in ELEMENTS; stride 0 = one instance shared by the whole batch).
 *  - Matrices G (n x n, only the lower triangle is read -- qp.cc:289, :404) and A_eq (k x n) are COLUMN-MAJOR
 *    with a leading dimension, exactly Eigen's default, so `MatrixXd::data()` can be passed unchanged.
 *    The stacked Jacobian J (m_r x n; the reference never materialises it -- residual.hpp:198-224 accumulates
 *    J^T J residual by residual) is ROW-MAJOR by default (each residual appends its rows); column-major is
 *    accepted via J_layout.
 *  - State / residual / delta vectors use the reference block order [x(n) | s(m) | y(k) | z(m)]
 *    (qp.cc:36-42, 548-582); V = n + 2m + k.
 *  - dtype MO_F64 is the reference's arithmetic (qp.hpp:15: double only); MO_F32 is the BASELINE.json cfg-4
 *    extension: every floating-point tensor (incl. mu, alpha, ip records) is then float.
 *  - Thread-safety: a plan may be used from one host thread -- and one stream -- at a time (it owns the device work counter
 *    of the fused kernels and the scratch of mo_qp_solve); distinct plans are independent.
 *    Multi-GPU = one plan (and one process or thread) per device; there is no cross-device traffic.
 */
#ifndef MINI_OPT_HIP_H_
#define MINI_OPT_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MO_VERSION_MAJOR 0
#define MO_VERSION_MINOR 1

/* return codes */
#define MO_OK 0
#define MO_ERR_INVALID_ARGUMENT (-1) /* null pointer, bad enum, bad params (CheckParams qp.cc:76-82) */
#define MO_ERR_DIMENSION (-2)        /* dimension mismatch (Setup asserts, qp.cc:24-34) */
#define MO_ERR_UNSUPPORTED (-3)      /* unknown dtype; state / residual vectors of one problem beyond the 160 KiB of LDS (n in the thousands) ... */
#define MO_ERR_HIP (-4)              /* a HIP runtime call failed; see mo_last_error() */
#define MO_ERR_NO_DEVICE (-5)        /* no usable gfx950 device: the HIP path never falls back to the CPU */

/* per-problem status words */
#define MO_STATUS_OK 0
#define MO_STATUS_NONPOSITIVE_SLACK 1    /* some s <= 0 at factorisation time (F_ASSERT qp.cc:285) */
#define MO_STATUS_FACTORIZATION_FAILED 2 /* FailedFactorization (qp.cc:303-307, qp.hpp:331-333) */
#define MO_STATUS_NONFINITE 3            /* the computed direction contains NaN/Inf */
#define MO_STATUS_BAD_INDEX 4            /* constraint variable index outside [0,n) (F_ASSERT qp.cc:70-72) */
#define MO_STATUS_NOT_POSITIVE_DEFINITE 5 /* mo_nullspace_solve: QPNullSpaceTerminationState::NOT_POSITIVE_DEFINITE (qp.cc:711-713) */

typedef enum { MO_F64 = 0, MO_F32 = 1 } mo_dtype;
typedef enum { MO_COL_MAJOR = 0, MO_ROW_MAJOR = 1 } mo_layout;

/* BarrierStrategy (structs.hpp:24-31), InitialGuessMethod (structs.hpp:34-41), termination (structs.hpp:97-102) */
typedef enum { MO_COMPLEMENTARITY = 0, MO_FIXED_DECREASE = 1, MO_PREDICTOR_CORRECTOR = 2 } mo_barrier_strategy;
typedef enum { MO_GUESS_NAIVE = 0, MO_GUESS_SOLVE_EQUALITY_CONSTRAINED = 1, MO_GUESS_USER_PROVIDED = 2 } mo_initial_guess;
typedef enum { MO_SATISFIED_KKT_TOL = 0, MO_MAX_ITERATIONS = 1 } mo_termination;

/* plan flags */
#define MO_PLAN_FORCE_GENERIC 1u /* always use the shape-generic LDS kernel (testing / A-B measurements) */
#define MO_PLAN_NO_TINY 2u       /* do not use the one-tile kernels for n + k <= 15 (testing / A-B: the 32-variable tile grid instead) */
/* How the fused kernels hand out problems to the waves of their persistent grid.  Default: launches of a few problems per wave are split
 * statically, larger ones take tickets from a device counter in guided chunks.  The two flags pin one scheme for every launch of the plan
 * (testing: the suite's small batches would otherwise never take a ticket). */
#define MO_PLAN_TICKETS_ALWAYS 4u
#define MO_PLAN_STATIC_ROUNDS_ALWAYS 8u

typedef struct {
  int32_t n;   /* variables (QPInteriorPointSolver::dims_.N, qp.cc:37) */
  int32_t k;   /* equality rows (dims_.K) */
  int32_t m;   /* LinearInequalityConstraint entries (dims_.M); a two-sided box is two entries */
  int32_t m_r; /* stacked residual rows for J-level input; 0 if only QP-level (G, c) input is used */
  int32_t dtype;  /* mo_dtype */
  int32_t device; /* HIP device ordinal */
  uint32_t flags; /* MO_PLAN_* */
  int32_t reserved;
  int64_t max_batch; /* sizes plan-owned scratch: the (G, c) of mo_qp_solve with J-level input on the generic kernel; beyond the
                        LDS-resident range of the generic kernel (n + k >= 72 since round 4) also the number of H workspaces
                        mo_plan_create allocates (0: one per workgroup of the full persistent grid) */
} mo_plan_desc;

typedef struct mo_plan mo_plan; /* opaque; replaces the solver-owned scratch of qp.hpp:221-231 */

/* One batch of QPs (mini_opt::QP, qp.hpp:104-124) or of linearised least-squares problems
 * (ConstrainedNonlinearLeastSquares::LinearizeAndFillQP, nonlinear.cc:170-214).
 * Cost: give EITHER (J, r, lambda) -- then G = J^T J + lambda I (lower) and c = J^T r are formed on device
 * (residual.hpp:206-224 summed, nonlinear.cc:187-189) -- OR (G, c).  J == NULL selects (G, c). */
typedef struct {
  const void* J;      int64_t J_stride; int32_t J_ld; int32_t J_layout; /* m_r x n */
  const void* r;      int64_t r_stride;                                /* m_r */
  double lambda;                                                      /* Levenberg-Marquardt damping, added iff > 0 */
  const void* G;      int64_t G_stride; int32_t G_ld; int32_t reserved0; /* n x n col-major, lower read */
  const void* c;      int64_t c_stride;                                /* n */
  const void* A_eq;   int64_t A_stride; int32_t A_ld; int32_t reserved1; /* k x n col-major (NULL iff k == 0) */
  const void* b_eq;   int64_t b_stride;                                /* k */
  const int32_t* cons_var; const void* cons_a; const void* cons_b; int64_t cons_stride; /* m each: a*x[var]+b >= 0 */
  const void* lambda_vec; int64_t lambda_stride; /* optional per-problem damping (the lambda state of nonlinear.cc:92-96);
                                                    overrides `lambda` when non-NULL */
} mo_problem;

/* QPInteriorPointSolver::Params (qp.hpp:134-164); mo_default_solve_params fills the reference defaults. */
typedef struct {
  double initial_mu;
  double sigma;
  double termination_kkt_tol;
  double termination_complementarity_tol;
  int32_t max_iterations;
  int32_t barrier_strategy;                 /* mo_barrier_strategy */
  int32_t decrease_mu_only_on_small_error;
  int32_t initial_guess_method;             /* mo_initial_guess */
  int32_t initialize_mu_with_complementarity;
  int32_t reserved;
} mo_solve_params;

/* Records are arrays of the plan's scalar type, laid out as:
 *   KKTError (structs.hpp:68-78)            kkt[4]  = {r_dual, r_comp, r_primal_eq, r_primal_ineq}
 *   IPIterationOutputs (structs.hpp:53-64)  ip[6]   = {mu, alpha.primal, alpha.dual, alpha_probe.primal,
 *                                                      alpha_probe.dual, mu_affine}   (NaN where the reference has NaN)
 *   QPInteriorPointIteration (:81-94)       iter[14] = {kkt_initial[4], kkt_final[4], ip[6]}
 */
#define MO_KKT_RECORD 4
#define MO_IP_RECORD 6
#define MO_ITER_RECORD 14

/* step flags */
#define MO_STEP_NO_INEQUALITIES 1u    /* EvaluateKKTConditions(false) + ComputeLDLT(false) + SolveForUpdateNoInequalities
                                         (qp.cc:366-386, used by the initial guess :455-460): only dx, dy are written,
                                         ds = dz = 0, alpha = 1 */
#define MO_STEP_PREDICTOR_CORRECTOR 2u /* mo_iterate only: the Mehrotra double solve of qp.cc:170-187 */

const char* mo_version_string(void);
const char* mo_status_string(int32_t status);
/* Message of the last failing call on this host thread ("" if none). */
const char* mo_last_error(void);

void mo_default_solve_params(mo_solve_params* params);

/* Replaces QPInteriorPointSolver::Setup (qp.cc:20-73): validates dimensions, selects kernels, allocates scratch. */
int mo_plan_create(const mo_plan_desc* desc, mo_plan** plan);
int mo_plan_destroy(mo_plan* plan);
/* Name of the kernel variant mo_newton_step will launch for this plan and problem layout ("generic", "fused_n64", ...). */
const char* mo_plan_step_kernel(const mo_plan* plan, const mo_problem* prob);
/* The same for mo_qp_solve / mo_iterate / mo_kkt_residual ("generic", "fused_solve_mfma_f64_n64", "fused_solve_mfma_f32_n128", ...). */
const char* mo_plan_solve_kernel(const mo_plan* plan, const mo_problem* prob);

/* Replaces LinearizeAndFillQP's cost part (nonlinear.cc:182-189; Residual::Model::UpdateHessian, residual.hpp:186-226):
 * G_out (n x n col-major, leading dim G_ld; lower triangle written, strict upper written as 0) = J^T J + lambda I,
 * c_out = J^T r, half_sq_out[p] = 0.5 |r|^2 (may be NULL). */
int mo_linearize(mo_plan* plan, const mo_problem* prob, int64_t batch, void* G_out, int64_t G_stride, int32_t G_ld,
                 void* c_out, int64_t c_stride, void* half_sq_out, void* stream);

/* QP::ComputeEigenvalueStats (qp.hpp:122-123, qp.cc:12-16): out[p] = {min, max, min |.|} of the eigenvalues of the QP Hessian of problem p
 * (QPEigenvalues, structs.hpp:267-275) -- of sym(G) from the lower triangle of (G, c) input, as Eigen's SelfAdjointEigenSolver reads it, or of
 * G = J^T J + lambda I for (J, r, lambda) input (what LinearizeAndFillQP hands the reference's QP, nonlinear.cc:182-189).  Any n the plan
 * accepts; computed in fp64 (Householder tridiagonalisation + Sturm-count multisection), written in the plan's dtype.  out: [batch][3]. */
int mo_qp_eigenvalue_stats(mo_plan* plan, const mo_problem* prob, int64_t batch, void* out, void* stream);

/* Replaces the whole of LinearizeAndFillQP (nonlinear.cc:170-214) for dense residual stacks.  The caller hands over the cost
 * stack (J, r, lambda[_vec]) and -- as prob->A_eq / prob->b_eq -- the equality residuals' Jacobian and values at the
 * linearisation point x (UpdateJacobian, residual.hpp:230-250, is a plain copy for a dense stack), plus the problem's
 * UNSHIFTED inequality constraints.  Written: G_out / c_out as mo_linearize; cons_b_out[m] = a * x[var] + b
 * (LinearInequalityConstraint::ShiftTo, qp.hpp:57-65); errors_out[p] = {f = 0.5 |r|^2, equality = |b_eq|_1}
 * (Errors, structs.hpp:169-186); status[p] = MO_STATUS_BAD_INDEX for a constraint variable outside [0, n). */
int mo_fill_qp(mo_plan* plan, const mo_problem* prob, int64_t batch, const void* x, int64_t x_stride, void* G_out,
               int64_t G_stride, int32_t G_ld, void* c_out, int64_t c_stride, void* cons_b_out, int64_t cons_b_stride,
               void* errors_out, int32_t* status, void* stream);

/* Replaces EvaluateNonlinearErrors (nonlinear.cc:279-293) for dense residual stacks evaluated by the caller:
 * errors_out[p] = {0.5 |r|^2, |r_eq|_1}.  r: m_r values per problem, r_eq: k values per problem (NULL iff k == 0). */
int mo_nonlinear_errors(mo_plan* plan, const void* r, int64_t r_stride, const void* r_eq, int64_t r_eq_stride,
                        int64_t batch, void* errors_out, void* stream);

/* Replaces ComputeQPCostDerivative (nonlinear.cc:452-483): deriv_out[p] = {d_f = c^T dx, d_equality = sum_i sign(b_eq_i)
 * (A_eq dx)_i} (DirectionalDerivatives, structs.hpp:189-203).  With J-level input c^T dx is evaluated as r^T (J dx).
 * quad_out[p] (may be NULL) = dx^T G dx, the curvature term SelectPenalty needs (nonlinear.cc:496-498). */
int mo_qp_cost_derivative(mo_plan* plan, const mo_problem* prob, int64_t batch, const void* dx, int64_t dx_stride,
                          void* deriv_out, void* quad_out, void* stream);

/* Replaces EvaluateKKTConditions (qp.cc:391-420) + ComputeErrors (qp.cc:423-437):
 * r_out [V] = [r_d | r_comp | r_pe | r_pi] (mu NOT applied, as in the reference), kkt_out [4] (may be NULL) with mu[p]. */
int mo_kkt_residual(mo_plan* plan, const mo_problem* prob, int64_t batch, const void* vars, int64_t vars_stride,
                    const void* mu, int64_t mu_stride, uint32_t flags, void* r_out, int64_t r_stride, void* kkt_out,
                    void* stream);

/* THE HOT PATH.  One dense KKT Newton step per problem on the caller's state; replaces the sequence
 *   EvaluateKKTConditions (qp.cc:391-420) -> ComputeLDLT (qp.cc:275-316) -> SolveForUpdate(mu) (qp.cc:318-364)
 *   -> ComputeAlpha(tau) (qp.cc:485-507)
 * (what qp_test.cc:132-134 calls through `friend`), including the J^T J assembly when J-level input is given.
 * The direction is obtained by a direct LDL^T solve; the reference's explicit inverse (qp.cc:310-311) is not formed.
 *   vars  [batch][V]  in   strictly interior state (s > 0)
 *   mu    per-problem barrier parameter (mu_stride 0: one shared value); ignored (0) if m == 0 (qp.cc:165-167)
 *   delta [batch][V]  out  [dx | ds | dy | dz]   (NaN-filled for problems whose status != 0)
 *   alpha [batch][2]  out  {primal, dual} step lengths (may be NULL)
 *   status[batch]     out  MO_STATUS_* (may be NULL) */
int mo_newton_step(mo_plan* plan, const mo_problem* prob, int64_t batch, const void* vars, int64_t vars_stride,
                   const void* mu, int64_t mu_stride, double tau, uint32_t flags, void* delta, int64_t delta_stride,
                   void* alpha, int32_t* status, void* stream);

/* Replaces QPInteriorPointSolver::Iterate (qp.cc:153-201): Newton step (optionally predictor-corrector,
 * qp.cc:170-187 with ComputePredictorCorrectorMuAffine :519-537), ComputeAlpha(0.995), and the state update
 * x,s += alpha_p (dx,ds); y,z += alpha_d (dy,dz) IN PLACE in vars.  ip_out [batch][6] (may be NULL),
 * delta (may be NULL). */
int mo_iterate(mo_plan* plan, const mo_problem* prob, int64_t batch, void* vars, int64_t vars_stride, const void* mu,
               int64_t mu_stride, int32_t barrier_strategy, void* delta, int64_t delta_stride, void* ip_out,
               int32_t* status, void* stream);

/* Replaces QPInteriorPointSolver::Solve (qp.cc:100-151) incl. ComputeInitialGuess (qp.cc:439-482): the whole
 * interior-point loop runs on device, one problem per workgroup, with per-problem early termination.
 *   vars            [batch][V] in/out (input only read for MO_GUESS_USER_PROVIDED)
 *   termination     [batch] out mo_termination
 *   num_iterations  [batch] out
 *   iterations      [batch][max_iterations][14] out (may be NULL)
 *   lagrange        [batch][2] out {min(y), |y|_inf} (may be NULL; qp.cc:539-546) */
int mo_qp_solve(mo_plan* plan, const mo_problem* prob, int64_t batch, const mo_solve_params* params, void* vars,
                int64_t vars_stride, int32_t* termination, int32_t* num_iterations, void* iterations, void* lagrange,
                int32_t* status, void* stream);

/* Replaces QPNullSpaceSolver::Solve (qp.cc:679-729): minimise 1/2 x^T G x + c^T x subject to A_eq x + b_eq = 0 (k >= 1, no
 * inequalities; the plan's m is ignored).  x_out [batch][n] = QPNullSpaceSolver::variables(); termination [batch] =
 * QPNullSpaceTerminationState (structs.hpp:137-142): 0 SUCCESS, 1 NOT_POSITIVE_DEFINITE (x_out is NaN then).
 * Same algorithm as the reference, one workgroup per problem with G and A_eq^T resident in LDS: Householder QR of A_eq^T with column
 * pivoting (qp.cc:687; rank = Eigen's default threshold |R_jj| > |R|_max eps min(n, k), :697), u = Q1 R1^-T P^T (-b_eq) (:703-704),
 * G_reduced = Q2^T G Q2 by two-sided application of the reflectors (:708), LLT that fails iff a pivot is <= 0 (:711-714 ->
 * NOT_POSITIVE_DEFINITE), y from -(Q2^T (c + G u)) (:718-721), x = u + Q2 y (:725).  What decides SUCCESS is the reduced Hessian
 * alone: a singular or indefinite G is fine when it is positive definite on null(A_eq).  k <= n is required; for a rank-deficient
 * A_eq (rank r < k) the solve uses R's leading r x r block (the reference's k x k solve against Q1's r columns is a size mismatch
 * there).  Limits: LDS-resident, (n | 1) (n + k) + 5 n + 4 k scalars <= 160 KiB (n = 128, k = 16 fits in fp64). */
typedef enum { MO_NULLSPACE_SUCCESS = 0, MO_NULLSPACE_NOT_POSITIVE_DEFINITE = 1 } mo_nullspace_termination;
int mo_nullspace_solve(mo_plan* plan, const mo_problem* prob, int64_t batch, void* x_out, int64_t x_stride,
                       int32_t* termination, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Batched constrained nonlinear least squares: ConstrainedNonlinearLeastSquares::Solve (nonlinear.cc:75-158) for a batch of
 * independent problems of one shape, every problem with its own lambda / penalty / optimizer state, in lock step.
 * The residual functions stay with the caller (the reference's Residual objects are host functors, residual.hpp:28-143):
 * `eval` is called on the host and must ENQUEUE, on `stream`, device work that fills the buffers named in mo_nls_problem:
 *   MO_NLS_EVAL_LINEARIZE : J, r (cost stack) and J_eq, r_eq (equality stack) at `vars`        (LinearizeAndFillQP)
 *   MO_NLS_EVAL_ERRORS    : r_cand, r_eq_cand at `candidate`                                   (EvaluateNonlinearErrors)
 * for ALL problems of the batch (finished problems are ignored afterwards).  Everything else -- the errors, the shifted
 * constraints, the interior-point QP (mo_qp_solve), directional derivatives, penalty selection, the line search with
 * quadratic / cubic interpolation, the lambda state machine and the exit tests -- runs on the device.
 * fp64 plans only.  A problem whose QP fails (status != MO_STATUS_OK; the reference throws there) ends with
 * MO_NLS_QP_FAILURE and its QP status in status[p].  With equality constraints and no inequalities the reference switches
 * to QPNullSpaceSolver (nonlinear.cc:83-86, 249-258) and so does this call (the null-space kernel behind mo_nullspace_solve: a
 * singular G = J^T J is fine as long as the reduced Hessian is positive definite); the penalty then follows the no-multiplier
 * branch of SelectPenalty (nonlinear.cc:491-499), and NOT_POSITIVE_DEFINITE ends the problem with MO_NLS_QP_INDEFINITE
 * (nonlinear.cc:103-105). */
typedef enum { MO_NLS_MAX_ITERATIONS = 0, MO_NLS_SATISFIED_ABSOLUTE_TOL = 1, MO_NLS_SATISFIED_RELATIVE_TOL = 2,
               MO_NLS_SATISFIED_FIRST_ORDER_TOL = 3, MO_NLS_MAX_LAMBDA = 4, MO_NLS_QP_INDEFINITE = 5,
               MO_NLS_USER_CALLBACK = 6, MO_NLS_QP_FAILURE = 7 } mo_nls_termination;   /* NLSTerminationState, structs.hpp:233-248 */
typedef enum { MO_LS_SUCCESS = 0, MO_LS_MAX_ITERATIONS = 1, MO_LS_FIRST_ORDER_SATISFIED = 2, MO_LS_POSITIVE_DERIVATIVE = 3,
               MO_LS_FAILURE_NON_FINITE_COST = 4, MO_LS_FAILURE_INVALID_ALPHA = 5 } mo_step_result; /* StepSizeSelectionResult, structs.hpp:215-228 */
typedef enum { MO_ARMIJO_BACKTRACK = 0, MO_POLYNOMIAL_APPROXIMATION = 1 } mo_line_search;      /* structs.hpp:148-153 */
typedef enum { MO_NLS_EVAL_LINEARIZE = 0, MO_NLS_EVAL_ERRORS = 1, MO_NLS_EVAL_RETRACT = 2, MO_NLS_EVAL_ITERATION_DONE = 3 } mo_nls_eval;
/* Retraction (nonlinear.hpp:127, RetractCandidateVars nonlinear.cc:160-168): how a trial point is formed from vars, the step dx and
 * the step length alpha.  EUCLIDEAN = the reference's default x + alpha dx; WRAP_PI = every variable wrapped into [-pi, pi) afterwards
 * (math::ModPi: the custom Retraction of the reference's robot tests, nonlinear_test.cc:874-880, 1077-1084); CALLBACK = the caller's
 * own: before each evaluation the library writes dx to mo_nls_problem.step and alpha to mo_nls_problem.step_alpha and calls
 * eval(user, MO_NLS_EVAL_RETRACT, stream), which must enqueue work that fills `candidate` for all problems. */
typedef enum { MO_RETRACT_EUCLIDEAN = 0, MO_RETRACT_WRAP_PI = 1, MO_RETRACT_CALLBACK = 2 } mo_retraction;

/* ConstrainedNonlinearLeastSquares::Params (nonlinear.hpp:64-124); mo_default_nls_params fills the reference defaults. */
typedef struct {
  int32_t max_iterations;
  int32_t max_qp_iterations;
  double termination_kkt_tolerance;
  double absolute_exit_tol;
  double relative_exit_tol;
  double absolute_first_derivative_tol;
  int32_t max_line_search_iterations;
  int32_t line_search_strategy;        /* mo_line_search */
  double armijo_search_tau;
  double equality_penalty_initial;
  double equality_penalty_scale_factor;
  double equality_penalty_rho;
  double lambda_initial;
  double lambda_failure_init;
  double lambda_decrease_on_success;
  double lambda_decrease_on_restore;
  double max_lambda;
  double min_lambda;
  int32_t retraction;                  /* mo_retraction */
  int32_t log_qp_eigenvalues;          /* Params::log_qp_eigenvalues (nonlinear.hpp:122-123): non-zero = every outer iteration records the
                                          QPEigenvalues of its QP Hessian into mo_nls_problem.qp_eigenvalues (nonlinear.cc:138) */
} mo_nls_params;

/* Device buffers of one batch (all owned by the caller; shapes per problem, strides in elements, plan dims n, k, m, m_r). */
typedef struct {
  void* vars;      int64_t vars_stride;      /* n : in = initial guess, out = solution (Solve(params, variables), variables()) */
  void* candidate; int64_t candidate_stride; /* n : the line search's trial point (RetractCandidateVars, nonlinear.cc:160-168) */
  void* J;     int64_t J_stride; int32_t J_ld; int32_t J_layout;   /* m_r x n, filled by eval(LINEARIZE) */
  void* r;     int64_t r_stride;                                  /* m_r */
  void* J_eq;  int64_t J_eq_stride; int32_t J_eq_ld; int32_t reserved0; /* k x n column-major (= QP::A_eq), eval(LINEARIZE) */
  void* r_eq;  int64_t r_eq_stride;                                /* k (= QP::b_eq) */
  void* r_cand;    int64_t r_cand_stride;    /* m_r, filled by eval(ERRORS) */
  void* r_eq_cand; int64_t r_eq_cand_stride; /* k */
  const int32_t* cons_var; const void* cons_a; const void* cons_b; int64_t cons_stride; /* m: Problem::inequality_constraints */
  void* step; int64_t step_stride;   /* n : dx of the current outer iteration (written only with MO_RETRACT_CALLBACK) */
  void* step_alpha;                  /* 1 per problem: the alpha of the trial point to form (MO_RETRACT_CALLBACK) */
  int32_t* user_exit;                /* NULL, or [batch] flags for SetUserExitCallback (nonlinear.hpp:157, nonlinear.cc:142-149): after every
                                        outer iteration eval(user, MO_NLS_EVAL_ITERATION_DONE, stream) is called (the iteration's record is in
                                        `iterations`, the state in `vars`); a problem whose flag it sets non-zero ends with
                                        MO_NLS_USER_CALLBACK unless that iteration terminated it anyway.  Zeroed by mo_nls_solve on entry. */
  void* qp_iterations;               /* NULL, or [max_iterations][batch][max_qp_iterations][MO_ITER_RECORD]: the QPInteriorPointIteration records
                                        of every outer iteration's QP (NLSIteration::qp_outputs, structs.hpp:288); the caller pre-fills NaN */
  void* qp_lagrange;                 /* NULL, or [max_iterations][batch][2]: QPLagrangeMultipliers {min, l_infinity} of each QP (k > 0) */
  void* qp_eigenvalues;              /* required with params.log_qp_eigenvalues: [max_iterations][batch][3] = NLSIteration::qp_eigenvalues
                                        {min, max, abs_min} of G = J^T J + lambda I of each outer iteration (structs.hpp:267-310); the caller
                                        pre-fills NaN (iterations a problem never ran keep it) */
} mo_nls_problem;

typedef int (*mo_nls_eval_fn)(void* user, int32_t what, void* stream);  /* non-zero return aborts mo_nls_solve with MO_ERR_CALLBACK */
#define MO_ERR_CALLBACK (-6)

/* NLSIteration (structs.hpp:277-330) as doubles: {optimizer_state, lambda, errors_initial.f, errors_initial.equality, d_f,
 * d_equality, penalty, step_result, n_line_search_steps, qp_termination, qp_num_iterations, qp_status} followed by
 * (max_line_search_iterations + 1) x {alpha, f, equality} (LineSearchStep, structs.hpp:206-213; NaN when not taken). */
#define MO_NLS_ITER_HEADER 12
#define MO_NLS_ITER_RECORD(max_line_search_iterations) (MO_NLS_ITER_HEADER + 3 * ((max_line_search_iterations) + 1))

void mo_default_nls_params(mo_nls_params* params);
/* 1 when mo_nls_solve takes QPNullSpaceSolver's path for this plan's shape (equalities, no inequalities: nonlinear.cc:83-86 -- and the
 * shape fits the null-space kernel), 0 when its QPs go through the interior-point Solve: tells which record layout `iterations` carries
 * (NLSIteration::qp_outputs / qp_term_state, structs.hpp:277-326). */
int mo_plan_nls_uses_nullspace(const mo_plan* plan);
/* termination [batch] (mo_nls_termination), num_iterations [batch], iterations [batch][max_iterations][MO_NLS_ITER_RECORD]
 * (may be NULL), status [batch] (may be NULL).  Synchronises `stream` (one small read-back per outer iteration and per
 * line-search step to learn whether any problem is still active). */
int mo_nls_solve(mo_plan* plan, const mo_nls_problem* prob, int64_t batch, const mo_nls_params* params, mo_nls_eval_fn eval,
                 void* user, int32_t* termination, int32_t* num_iterations, void* iterations, int32_t* status, void* stream);

/* Device residual families: the residual functions of the reference's own NLS tests as kernels (its Residual functors are host
 * lambdas, residual.hpp:28-143), so that an mo_nls_solve callback can be two launches instead of a PCIe round trip.
 *   ROSENBROCK     n >= 2, rows = 2 (n - 1): r_2i = 1 - x_i, r_2i+1 = 10 (x_i+1 - x_i^2)     (nonlinear_test.cc:375-386, 502-521)
 *   HIMMELBLAU     n = 2, rows = 2: x^2 + y - 11, x + y^2 - 7                                  (nonlinear_test.cc:578-593)
 *   SPHERE         rows = n: r = x                                                             (nonlinear_test.cc:722-730)
 *   PRODUCT_PAIRS  rows <= n / 2: r_q = x_2q x_2q+1 - params[q]  (params: `rows` device scalars) (nonlinear_test.cc:737-743)
 *   ACTUATOR_CHAIN the kinematic chains of the reference's robot tests (test/transform_chains.cc:23-82 ComputeChain, :125-158
 *                  ActuatorLink::Compute, :165-244 ActuatorChain::Update; used at nonlinear_test.cc:828-1136): every row is affine in
 *                  the effector translations of up to 4 chains and in x,  r_q = const_q + sum_i lin_qi x_i + sum_c w_qc . t_c(x),
 *                  with the Jacobian through translation_D_params.  params (device scalars of the plan's dtype, integers stored as
 *                  scalars):  [C,  then per chain: L, then per link 12 values: euler-xyz rotation (R = Rx Ry Rz), translation, and for
 *                  each of the 6 degrees of freedom (rx, ry, rz, tx, ty, tz) the index in x of the parameter that replaces it or -1;
 *                  then per row: const, n_lin, n_lin x (index, coefficient), n_terms, n_terms x (chain, wx, wy, wz)].  L <= 8.
 * r [batch][rows]; J (NULL = values only) dense rows x n per problem, row-major (cost stacks) or column-major (equality stacks,
 * = QP::A_eq), zeros written. */
typedef enum { MO_RESIDUAL_ROSENBROCK = 0, MO_RESIDUAL_HIMMELBLAU = 1, MO_RESIDUAL_SPHERE = 2, MO_RESIDUAL_PRODUCT_PAIRS = 3,
               MO_RESIDUAL_ACTUATOR_CHAIN = 4 } mo_residual_family;
int mo_residual_eval(mo_plan* plan, int32_t family, int32_t rows, const void* params, const void* x, int64_t x_stride,
                     int64_t batch, void* r, int64_t r_stride, void* J, int64_t J_stride, int32_t J_ld, int32_t J_layout,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MINI_OPT_HIP_H_ */
