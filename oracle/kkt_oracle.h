/*
 * kkt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, double precision, no Eigen) of the dense KKT Newton-step hot path of
 * gareth-cross/mini_opt (source/qp.cc, include/mini_opt/residual.hpp, source/nonlinear.cc).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library, and only
 * as the checker / reported CPU baseline.  The product path (mini_opt_amd/, include/mini_opt_hip.h) never
 * links, imports or calls anything in oracle/.
 *
 * Parity pin: the reference cannot be compiled here (Eigen is an un-vendored submodule, absent from the
 * image), so this restatement is pinned by the reference's OWN differential tests and known-answer tests
 * (test/qp_test.cc:101-138,168-249,252-471; test/residual_test.cc:51-182), restated as committed fixtures
 * under tests/golden/ whose expected values are computed independently with numpy (full-system LU).
 *
 * Every function cites the reference lines it follows.  All matrices are COLUMN-MAJOR (Eigen default).
 * State / residual / delta vectors use the reference block order [x(N) | s(M) | y(K) | z(M)]
 * (qp.cc:36-42, 548-582).
 */
#ifndef KKT_ORACLE_H_
#define KKT_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* status codes of orc_compute_ldlt (mirror the reference's two failure exceptions) */
#define ORC_OK 0
#define ORC_NONPOSITIVE_SLACK 1    /* F_ASSERT at qp.cc:285 */
#define ORC_FACTORIZATION_FAILED 2 /* FailedFactorization, qp.cc:303-307 */

/* BarrierStrategy, structs.hpp:24-31 */
#define ORC_COMPLEMENTARITY 0
#define ORC_FIXED_DECREASE 1
#define ORC_PREDICTOR_CORRECTOR 2

/* InitialGuessMethod, structs.hpp:34-41 */
#define ORC_GUESS_NAIVE 0
#define ORC_GUESS_SOLVE_EQUALITY_CONSTRAINED 1
#define ORC_GUESS_USER_PROVIDED 2

/* QPInteriorPointTerminationState, structs.hpp:97-102 */
#define ORC_SATISFIED_KKT_TOL 0
#define ORC_MAX_ITERATIONS 1

/* QP (qp.hpp:104-124).  Non-owning views. */
typedef struct {
  int n, k, m;
  const double* G;    /* n x n col-major, only the lower triangle is read (qp.cc:289, :404) */
  const double* c;    /* n */
  const double* A_eq; /* k x n col-major */
  const double* b_eq; /* k */
  const int* cons_var;  /* m  LinearInequalityConstraint::variable (qp.hpp:28-70) */
  const double* cons_a; /* m */
  const double* cons_b; /* m */
} orc_qp;

/* QPInteriorPointSolver::Params (qp.hpp:134-164), same defaults via orc_default_params */
typedef struct {
  double initial_mu;
  double sigma;
  double termination_kkt_tol;
  double termination_complementarity_tol;
  int max_iterations;
  int barrier_strategy;
  int decrease_mu_only_on_small_error;
  int initial_guess_method;
  int initialize_mu_with_complementarity;
} orc_params;

/* KKTError (structs.hpp:68-78) */
typedef struct {
  double r_dual, r_comp, r_primal_eq, r_primal_ineq;
} orc_kkt_error;

/* IPIterationOutputs (structs.hpp:53-64) */
typedef struct {
  double mu;
  double alpha_primal, alpha_dual;
  double alpha_probe_primal, alpha_probe_dual;
  double mu_affine;
} orc_ip_outputs;

/* QPInteriorPointIteration (structs.hpp:81-94) */
typedef struct {
  orc_kkt_error kkt_initial, kkt_final;
  orc_ip_outputs ip;
} orc_iteration;

/* Solver scratch: the members of QPInteriorPointSolver (qp.hpp:209-231). */
typedef struct {
  orc_qp qp;
  int N, M, K, P, V;
  double* variables;    /* V */
  double* r;            /* V */
  double* r_dual_aug;   /* N */
  double* H;            /* P x P col-major, zeroed in setup (qp.cc:47) */
  double* H_inv;        /* P x P */
  double* delta;        /* V */
  double* delta_affine; /* V */
  /* LDLT object state (Eigen LDLT<MatrixXd,Lower>): factor matrix, transpositions, temp */
  double* ldlt_mat;     /* P x P */
  int* ldlt_transp;     /* P */
  double* ldlt_temp;    /* P */
  double* work;         /* P, rhs for the direct-solve variant */
} orc_solver;

void orc_default_params(orc_params* p);

/* Setup, qp.cc:20-73.  Returns 0, or <0 on a dimension / index error (the F_ASSERTs). */
int orc_solver_setup(orc_solver* s, const orc_qp* qp);
void orc_solver_free(orc_solver* s);

/* Residual::Model::UpdateHessian, residual.hpp:186-226.  J is R x Ploc col-major. Returns 0.5*|r|^2. */
double orc_update_hessian(int R, int Ploc, const int* index, const double* J, const double* r, int n,
                          double* H, double* b);
/* Residual::Model::UpdateJacobian, residual.hpp:230-250: J_out (rows x n block with leading dim ld). */
void orc_update_jacobian(int R, int Ploc, const int* index, const double* J, const double* r, int ld,
                         double* J_out, double* b_out);
/* LinearInequalityConstraint::ShiftTo (qp.hpp:57-65) for a constraint list, nonlinear.cc:209-212 */
void orc_shift_constraints(int m, const int* var, const double* a, const double* b, const double* x,
                           double* b_out);
/* Dense stacked variant of nonlinear.cc:182-189: G_lower = J^T J (+ lambda on the diagonal), c = J^T r for ONE
 * dense residual with identity index.  J is m_r x n, row-major if row_major else col-major. */
double orc_linearize_dense(int m_r, int n, const double* J, int row_major, const double* r, double lambda,
                           double* G, double* c);

/* EvaluateKKTConditions, qp.cc:391-420 */
void orc_evaluate_kkt(orc_solver* s, int include_inequalities);
/* ComputeLDLT, qp.cc:275-316 (assembly + Eigen LDLT + explicit inverse).  Returns an ORC_* status. */
int orc_compute_ldlt(orc_solver* s, int include_inequalities);
/* SolveForUpdate, qp.cc:318-364 (uses H_inv) */
void orc_solve_for_update(orc_solver* s, double mu);
/* Same right-hand side and back-substitution, but a direct LDLT solve instead of the explicit inverse. */
void orc_solve_for_update_direct(orc_solver* s, double mu);
/* SolveForUpdateNoInequalities, qp.cc:366-386 */
void orc_solve_no_inequalities(orc_solver* s);
/* ComputeAlpha, qp.cc:485-507 */
double orc_compute_alpha_vec(int n, const double* val, const double* d_val, double tau);
void orc_compute_alpha(const orc_solver* s, double tau, double* primal, double* dual);
/* ComputeMu qp.cc:509-516; ComputePredictorCorrectorMuAffine qp.cc:519-537 */
double orc_compute_mu(const orc_solver* s);
double orc_compute_mu_affine(const orc_solver* s, double mu, double alpha_p, double alpha_d);
/* ComputeErrors qp.cc:423-437 */
void orc_compute_errors(const orc_solver* s, double mu, orc_kkt_error* out);
/* ComputeInitialGuess qp.cc:439-482; returns ORC_* status of the inner ComputeLDLT(false) */
int orc_initial_guess(orc_solver* s, const orc_params* p);
/* Iterate qp.cc:153-201; returns ORC_* status */
int orc_iterate(orc_solver* s, double mu_input, int strategy, orc_ip_outputs* out);
/* Solve qp.cc:100-151.  iterations[] must hold max_iterations records.  Returns termination state (>=0) or
 * -(ORC status) if the factorisation failed (the reference throws). */
int orc_solve(orc_solver* s, const orc_params* p, orc_iteration* iterations, int* num_iterations);

/* The "Newton step" of the metric: EvaluateKKTConditions -> ComputeLDLT -> SolveForUpdate(mu) -> ComputeAlpha(tau)
 * (qp_test.cc:132-134 + qp.cc:192) on the caller's state.  use_inverse selects the reference's explicit inverse
 * (qp.cc:310-311) or the direct solve.  Writes delta (V), alpha[2]; returns ORC_* status. */
int orc_newton_step(orc_solver* s, const double* vars, double mu, double tau, int use_inverse, double* delta,
                    double* alpha);

/* BuildFullSystem qp.cc:595-655 (H_full V x V col-major, r_full V). */
void orc_build_full_system(const orc_solver* s, double* H_full, double* r_full);
/* PartialPivLU solve used by qp_test.cc:120-124 (independent cross-check). A is n x n col-major (destroyed).
 * Returns 0, or 1 if singular. */
int orc_lu_solve(int n, double* A, double* b);
/* delta = flip_yz(LU_solve(H_full, -r_full)), qp_test.cc:120-129 */
int orc_full_system_step(const orc_solver* s, double* delta);

/* Eigen LDLT<MatrixXd,Lower> restatement (Eigen 3.4 src/Cholesky/LDLT.h, recalled).  A is P x P col-major,
 * lower triangle referenced, factorised in place.  Returns 1 on success, 0 on failure (NumericalIssue). */
int orc_ldlt_inplace(int P, double* A, int* transp, double* temp);
/* LDLT::solveInPlace for nrhs right-hand sides stored col-major in B (P x nrhs). */
void orc_ldlt_solve_inplace(int P, const double* A, const int* transp, double* B, int nrhs);

/* Batched, OpenMP-parallel Newton steps for the CPU baseline (bench.py cpu_baseline leg).  Contiguous slabs:
 * J [batch][m_r*n] (row-major if row_major), r [batch][m_r], lambda scalar, A_eq [batch][k*n] col-major,
 * b_eq [batch][k], cons_* [batch][m], vars [batch][V], mu [batch]; outputs delta [batch][V], alpha [batch][2],
 * status [batch].  If J == NULL, G [batch][n*n] and c [batch][n] are used instead.  Returns threads used. */
int orc_batched_newton_step(int batch, int n, int k, int m, int m_r, const double* J, int row_major,
                            const double* r, double lambda, const double* G, const double* c,
                            const double* A_eq, const double* b_eq, const int* cons_var, const double* cons_a,
                            const double* cons_b, const double* vars, const double* mu, double tau,
                            int use_inverse, int num_threads, double* delta, double* alpha, int* status);

/* Batched, OpenMP-parallel Solve (qp.cc:100-151 per problem through orc_solve, one re-used solver per thread like qp_test.cc:531-539):
 * the checker and CPU baseline of `bench.py --mode solve`.  Same slabs as orc_batched_newton_step; vars_io [batch][V] holds the caller's
 * state on entry (read only with ORC_GUESS_USER_PROVIDED) and the final iterate on exit; termination [batch] = orc_solve's return value,
 * num_iterations [batch].  Returns threads used. */
int orc_batched_solve(int batch, int n, int k, int m, int m_r, const double* J, int row_major, const double* r,
                      double lambda, const double* G, const double* c, const double* A_eq, const double* b_eq,
                      const int* cons_var, const double* cons_a, const double* cons_b, const orc_params* params,
                      int num_threads, double* vars_io, int* termination, int* num_iterations);

#ifdef __cplusplus
}
#endif
#endif /* KKT_ORACLE_H_ */
