/*
 * kkt_oracle.c -- TEST INFRASTRUCTURE ONLY (see kkt_oracle.h).
 *
 * Plain-C restatement of mini_opt's dense KKT Newton step.  Citations are into /root/reference.
 * Parity pin: tests/golden/ JSON fixtures (the reference's own differential tests / KATs, expected values from an
 * independent numpy full-system LU), checked by tests/test_oracle_golden.py.
 */
#include "kkt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- block accessors, qp.cc:548-582 */
static inline double* xblk(const orc_solver* s, double* v) { (void)s; return v; }
static inline double* sblk(const orc_solver* s, double* v) { return v + s->N; }
static inline double* yblk(const orc_solver* s, double* v) { return v + s->N + s->M; }
static inline double* zblk(const orc_solver* s, double* v) { return v + s->N + s->M + s->K; }

void orc_default_params(orc_params* p) {
  /* qp.hpp:134-164 */
  p->initial_mu = 1.0;
  p->sigma = 0.5;
  p->termination_kkt_tol = 1.0e-9;
  p->termination_complementarity_tol = 1.0e-6;
  p->max_iterations = 10;
  p->barrier_strategy = ORC_COMPLEMENTARITY;
  p->decrease_mu_only_on_small_error = 0;
  p->initial_guess_method = ORC_GUESS_NAIVE;
  p->initialize_mu_with_complementarity = 0;
}

/* ---------------------------------------------------------------- Setup, qp.cc:20-73 */
int orc_solver_setup(orc_solver* s, const orc_qp* qp) {
  if (!qp) return -1;                         /* qp.cc:21 */
  if (qp->n < 0 || qp->k < 0 || qp->m < 0) return -2;
  for (int i = 0; i < qp->m; ++i) {           /* qp.cc:70-72 (index >= 0 is implied by Eigen's own asserts) */
    if (qp->cons_var[i] >= qp->n || qp->cons_var[i] < 0) return -3;
  }
  memset(s, 0, sizeof(*s));
  s->qp = *qp;
  s->N = qp->n;                               /* qp.cc:37-39 */
  s->M = qp->m;
  s->K = qp->k;
  s->P = s->N + s->K;                         /* qp.cc:45 */
  s->V = s->N + 2 * s->M + s->K;              /* qp.cc:42 */
  const size_t P = (size_t)s->P, V = (size_t)s->V;
  s->variables = (double*)calloc(V + 1, sizeof(double));
  s->r = (double*)calloc(V + 1, sizeof(double));
  s->r_dual_aug = (double*)calloc((size_t)s->N + 1, sizeof(double));
  s->H = (double*)calloc(P * P + 1, sizeof(double)); /* H_.setZero(), qp.cc:47 */
  s->H_inv = (double*)calloc(P * P + 1, sizeof(double));
  s->delta = (double*)calloc(V + 1, sizeof(double));
  s->delta_affine = (double*)calloc(V + 1, sizeof(double));
  s->ldlt_mat = (double*)calloc(P * P + 1, sizeof(double));
  s->ldlt_transp = (int*)calloc(P + 1, sizeof(int));
  s->ldlt_temp = (double*)calloc(P + 1, sizeof(double));
  s->work = (double*)calloc(P + 1, sizeof(double));
  return 0;
}

void orc_solver_free(orc_solver* s) {
  free(s->variables); free(s->r); free(s->r_dual_aug); free(s->H); free(s->H_inv); free(s->delta);
  free(s->delta_affine); free(s->ldlt_mat); free(s->ldlt_transp); free(s->ldlt_temp); free(s->work);
  memset(s, 0, sizeof(*s));
}

/* ---------------------------------------------------------------- residual.hpp:186-226 */
double orc_update_hessian(int R, int Ploc, const int* index, const double* J, const double* r, int n,
                          double* H, double* b) {
  for (int row_local = 0; row_local < Ploc; ++row_local) {
    const int row_global = index[row_local];                       /* :208 */
    for (int col_local = 0; col_local <= row_local; ++col_local) { /* :210 */
      const int col_global = index[col_local];
      double JtT = 0.0;                                            /* :214 J.col(row).dot(J.col(col)) */
      for (int q = 0; q < R; ++q) JtT += J[q + (size_t)row_local * R] * J[q + (size_t)col_local * R];
      if (col_global <= row_global) {                              /* :216-220 lower triangle only */
        H[row_global + (size_t)col_global * n] += JtT;
      } else {
        H[col_global + (size_t)row_global * n] += JtT;
      }
    }
    double Jtr = 0.0;                                              /* :223 */
    for (int q = 0; q < R; ++q) Jtr += J[q + (size_t)row_local * R] * r[q];
    b[row_global] += Jtr;
  }
  double sq = 0.0;
  for (int q = 0; q < R; ++q) sq += r[q] * r[q];
  return 0.5 * sq;                                                 /* :225 */
}

/* residual.hpp:230-250 */
void orc_update_jacobian(int R, int Ploc, const int* index, const double* J, const double* r, int ld,
                         double* J_out, double* b_out) {
  for (int q = 0; q < R; ++q) b_out[q] = r[q];                     /* :243 */
  for (int col_local = 0; col_local < Ploc; ++col_local) {         /* :245-249 */
    const int col_global = index[col_local];
    for (int q = 0; q < R; ++q) J_out[q + (size_t)col_global * ld] = J[q + (size_t)col_local * R];
  }
}

/* qp.hpp:57-65 applied as in nonlinear.cc:209-212 */
void orc_shift_constraints(int m, const int* var, const double* a, const double* b, const double* x,
                           double* b_out) {
  for (int i = 0; i < m; ++i) b_out[i] = a[i] * x[var[i]] + b[i];
}

/* nonlinear.cc:182-189 with ONE dense residual and identity index (G, c start from zero). */
double orc_linearize_dense(int m_r, int n, const double* J, int row_major, const double* r, double lambda,
                           double* G, double* c) {
  memset(G, 0, sizeof(double) * (size_t)n * n); /* :182 */
  memset(c, 0, sizeof(double) * (size_t)n);     /* :183 */
  if (row_major) {
    /* same sums as residual.hpp:214/:223, accumulated row by row of J (rank-1 updates of the lower triangle) */
    for (int q = 0; q < m_r; ++q) {
      const double* Jq = J + (size_t)q * n;
      const double rq = r[q];
      for (int j = 0; j < n; ++j) {
        const double a = Jq[j];
        double* Gj = G + (size_t)j * n;
        for (int i = j; i < n; ++i) Gj[i] += Jq[i] * a;
        c[j] += a * rq;
      }
    }
  } else {
    for (int i = 0; i < n; ++i) {
      const double* Ji = J + (size_t)i * m_r;
      for (int j = 0; j <= i; ++j) {
        const double* Jj = J + (size_t)j * m_r;
        double acc = 0.0;
        for (int q = 0; q < m_r; ++q) acc += Ji[q] * Jj[q];
        G[i + (size_t)j * n] += acc;
      }
      double acc = 0.0;
      for (int q = 0; q < m_r; ++q) acc += Ji[q] * r[q];
      c[i] += acc;
    }
  }
  if (lambda > 0) {                              /* :187-189 */
    for (int i = 0; i < n; ++i) G[i + (size_t)i * n] += lambda;
  }
  double sq = 0.0;
  for (int q = 0; q < m_r; ++q) sq += r[q] * r[q];
  return 0.5 * sq;
}

/* ---------------------------------------------------------------- EvaluateKKTConditions, qp.cc:391-420 */
void orc_evaluate_kkt(orc_solver* s, int include_inequalities) {
  const int N = s->N, M = s->M, K = s->K;
  const orc_qp* p = &s->qp;
  const double* x = xblk(s, s->variables);
  const double* sv = sblk(s, s->variables);
  const double* y = yblk(s, s->variables);
  const double* z = zblk(s, s->variables);
  double* r_d = xblk(s, s->r);
  double* r_comp = sblk(s, s->r);
  double* r_pe = yblk(s, s->r);
  double* r_pi = zblk(s, s->r);

  /* :404  r_d = G.selfadjointView<Lower>() * x + c */
  for (int i = 0; i < N; ++i) {
    double acc = 0.0;
    for (int j = 0; j < N; ++j) {
      const double g = (j <= i) ? p->G[i + (size_t)j * N] : p->G[j + (size_t)i * N];
      acc += g * x[j];
    }
    r_d[i] = acc + p->c[i];
  }
  if (K > 0) {
    /* :406  r_d -= A_eq^T y */
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
      for (int q = 0; q < K; ++q) acc += p->A_eq[q + (size_t)i * K] * y[q];
      r_d[i] -= acc;
    }
    /* :408  r_pe = A_eq x + b_eq */
    for (int q = 0; q < K; ++q) {
      double acc = 0.0;
      for (int j = 0; j < N; ++j) acc += p->A_eq[q + (size_t)j * K] * x[j];
      r_pe[q] = acc + p->b_eq[q];
    }
  }
  if (include_inequalities) { /* :412-419 */
    for (int i = 0; i < M; ++i) {
      const int v = p->cons_var[i];
      const double a = p->cons_a[i];
      r_d[v] -= a * z[i];
      r_pi[i] = a * x[v] + p->cons_b[i] - sv[i];
      r_comp[i] = sv[i] * z[i];
    }
  }
}

/* ---------------------------------------------------------------- Eigen LDLT<MatrixXd, Lower>
 * Restated from Eigen 3.4 src/Cholesky/LDLT.h `ldlt_inplace<Lower>::unblocked` (recalled; Eigen itself is
 * not in the image).  Pivot = largest |diagonal| of the trailing part AS STORED at step k; the algorithm is
 * left-looking (only column k is updated at step k), so that diagonal has not yet received its Schur
 * updates.  A zero pivot is tolerated; failure only if a non-zero pivot follows a zero one, or a column
 * below a zero pivot is non-zero. */
#define AT(A, i, j) (A)[(size_t)(i) + (size_t)(j) * (size_t)P]

int orc_ldlt_inplace(int P, double* A, int* transp, double* temp) {
  int found_zero_pivot = 0;
  int ret = 1;
  if (P <= 1) {
    for (int i = 0; i < P; ++i) transp[i] = i;
    return 1;
  }
  for (int k = 0; k < P; ++k) {
    /* mat.diagonal().tail(size-k).cwiseAbs().maxCoeff(&idx): first occurrence of the maximum */
    int idx = k;
    double best = fabs(AT(A, k, k));
    for (int i = k + 1; i < P; ++i) {
      const double v = fabs(AT(A, i, i));
      if (v > best) { best = v; idx = i; }
    }
    transp[k] = idx;
    if (k != idx) {
      /* symmetric transposition touching only the lower triangle */
      const int sz = P - idx - 1;
      for (int j = 0; j < k; ++j) { /* row(k).head(k) <-> row(idx).head(k) */
        const double t = AT(A, k, j); AT(A, k, j) = AT(A, idx, j); AT(A, idx, j) = t;
      }
      for (int q = 0; q < sz; ++q) { /* col(k).tail(s) <-> col(idx).tail(s) */
        const int i = idx + 1 + q;
        const double t = AT(A, i, k); AT(A, i, k) = AT(A, i, idx); AT(A, i, idx) = t;
      }
      { const double t = AT(A, k, k); AT(A, k, k) = AT(A, idx, idx); AT(A, idx, idx) = t; }
      for (int i = k + 1; i < idx; ++i) {
        const double t = AT(A, i, k); AT(A, i, k) = AT(A, idx, i); AT(A, idx, i) = t;
      }
    }
    const int rs = P - k - 1;
    if (k > 0) {
      /* temp.head(k) = D.head(k) * A10^T ; A11 -= A10 * temp ; A21 -= A20 * temp */
      double acc = 0.0;
      for (int j = 0; j < k; ++j) {
        temp[j] = AT(A, j, j) * AT(A, k, j);
        acc += AT(A, k, j) * temp[j];
      }
      AT(A, k, k) -= acc;
      for (int i = k + 1; i < P; ++i) {
        double a = 0.0;
        for (int j = 0; j < k; ++j) a += AT(A, i, j) * temp[j];
        AT(A, i, k) -= a;
      }
    }
    const double realAkk = AT(A, k, k);
    const int pivot_is_valid = fabs(realAkk) > 0.0;
    if (k == 0 && !pivot_is_valid) {
      /* matrix is identically zero (if it is not, fail) */
      for (int j = 0; j < P; ++j) {
        transp[j] = j;
        for (int i = j + 1; i < P; ++i) ret = ret && (AT(A, i, j) == 0.0);
      }
      return ret;
    }
    if (rs > 0 && pivot_is_valid) {
      for (int i = k + 1; i < P; ++i) AT(A, i, k) /= realAkk;
    } else if (rs > 0) {
      for (int i = k + 1; i < P; ++i) ret = ret && (AT(A, i, k) == 0.0);
    }
    if (found_zero_pivot && pivot_is_valid) ret = 0;
    else if (!pivot_is_valid) found_zero_pivot = 1;
  }
  return ret;
}

/* LDLT::_solve_impl (solveInPlace): P^T L^-T D^+ L^-1 P b, D^+ zeroes entries with |d| <= DBL_MIN */
void orc_ldlt_solve_inplace(int P, const double* A, const int* transp, double* B, int nrhs) {
  for (int c = 0; c < nrhs; ++c) {
    double* b = B + (size_t)c * P;
    for (int k = 0; k < P; ++k) {
      const int t = transp[k];
      if (t != k) { const double tmp = b[k]; b[k] = b[t]; b[t] = tmp; }
    }
    for (int j = 0; j < P; ++j) { /* unit-lower forward substitution */
      const double bj = b[j];
      if (bj != 0.0) for (int i = j + 1; i < P; ++i) b[i] -= AT(A, i, j) * bj;
    }
    for (int i = 0; i < P; ++i) {
      const double d = AT(A, i, i);
      if (fabs(d) > DBL_MIN) b[i] /= d; else b[i] = 0.0;
    }
    for (int j = P - 1; j >= 0; --j) { /* unit-upper (L^T) back substitution */
      double acc = b[j];
      for (int i = j + 1; i < P; ++i) acc -= AT(A, i, j) * b[i];
      b[j] = acc;
    }
    for (int k = P - 1; k >= 0; --k) {
      const int t = transp[k];
      if (t != k) { const double tmp = b[k]; b[k] = b[t]; b[t] = tmp; }
    }
  }
}
#undef AT

/* ---------------------------------------------------------------- ComputeLDLT, qp.cc:275-316 */
int orc_compute_ldlt(orc_solver* s, int include_inequalities) {
  const int N = s->N, M = s->M, K = s->K, P = s->P;
  const orc_qp* p = &s->qp;
  const double* sv = sblk(s, s->variables);
  const double* z = zblk(s, s->variables);
  if (include_inequalities) { /* :285 */
    for (int i = 0; i < M; ++i) if (!(sv[i] > 0.0)) return ORC_NONPOSITIVE_SLACK;
  }
  /* :289  H.topLeft.lower = G.lower */
  for (int j = 0; j < N; ++j)
    for (int i = j; i < N; ++i) s->H[i + (size_t)j * P] = p->G[i + (size_t)j * N];
  /* :290-292  H.bottomLeft = A_eq */
  for (int j = 0; j < N; ++j)
    for (int q = 0; q < K; ++q) s->H[(N + q) + (size_t)j * P] = p->A_eq[q + (size_t)j * K];
  if (include_inequalities) { /* :293-298 */
    for (int i = 0; i < M; ++i) {
      const int v = p->cons_var[i];
      const double a = p->cons_a[i];
      s->H[v + (size_t)v * P] += a * (z[i] / sv[i]) * a;
    }
  }
  /* :302  const LDLT<MatrixXd, Lower> ldlt(H_) -- factorises a copy */
  memcpy(s->ldlt_mat, s->H, sizeof(double) * (size_t)P * P);
  const int ok = orc_ldlt_inplace(P, s->ldlt_mat, s->ldlt_transp, s->ldlt_temp);
  if (!ok) return ORC_FACTORIZATION_FAILED; /* :303-307 */
  /* :310-311  H_inv = I ; ldlt.solveInPlace(H_inv) */
  memset(s->H_inv, 0, sizeof(double) * (size_t)P * P);
  for (int i = 0; i < P; ++i) s->H_inv[i + (size_t)i * P] = 1.0;
  orc_ldlt_solve_inplace(P, s->ldlt_mat, s->ldlt_transp, s->H_inv, P);
  /* :314-315 */
  memset(s->delta, 0, sizeof(double) * (size_t)s->V);
  memset(s->delta_affine, 0, sizeof(double) * (size_t)s->V);
  return ORC_OK;
}

/* r_dual_aug of qp.cc:337-342 */
static void build_r_dual_aug(orc_solver* s, double mu) {
  const int N = s->N, M = s->M;
  const orc_qp* p = &s->qp;
  const double* sv = sblk(s, s->variables);
  const double* z = zblk(s, s->variables);
  const double* r_d = xblk(s, s->r);
  const double* r_comp = sblk(s, s->r);
  const double* r_pi = zblk(s, s->r);
  const double* ds_aff = sblk(s, s->delta_affine);
  const double* dz_aff = zblk(s, s->delta_affine);
  for (int i = 0; i < N; ++i) s->r_dual_aug[i] = r_d[i]; /* :337 */
  for (int i = 0; i < M; ++i) {                          /* :338-342 */
    const int v = p->cons_var[i];
    const double a = p->cons_a[i];
    s->r_dual_aug[v] += a * (z[i] / sv[i]) * r_pi[i];
    s->r_dual_aug[v] += a * (r_comp[i] + (ds_aff[i] * dz_aff[i]) - mu) / sv[i];
  }
}

/* ds, dz of qp.cc:359-363 */
static void back_substitute(orc_solver* s, double mu) {
  const int M = s->M;
  const orc_qp* p = &s->qp;
  const double* sv = sblk(s, s->variables);
  const double* z = zblk(s, s->variables);
  const double* r_comp = sblk(s, s->r);
  const double* r_pi = zblk(s, s->r);
  const double* ds_aff = sblk(s, s->delta_affine);
  const double* dz_aff = zblk(s, s->delta_affine);
  const double* dx = xblk(s, s->delta);
  double* ds = sblk(s, s->delta);
  double* dz = zblk(s, s->delta);
  for (int i = 0; i < M; ++i) {
    const int v = p->cons_var[i];
    ds[i] = p->cons_a[i] * dx[v] + r_pi[i];
    dz[i] = -(z[i] / sv[i]) * ds[i] - (1 / sv[i]) * (r_comp[i] + (ds_aff[i] * dz_aff[i]) - mu);
  }
}

/* SolveForUpdate, qp.cc:318-364 */
void orc_solve_for_update(orc_solver* s, double mu) {
  const int N = s->N, K = s->K, P = s->P;
  build_r_dual_aug(s, mu);
  const double* r_pe = yblk(s, s->r);
  double* dx = xblk(s, s->delta);
  double* dy = yblk(s, s->delta);
  /* :350  dx = Hinv[0:N,0:N] * -r_aug */
  for (int i = 0; i < N; ++i) {
    double acc = 0.0;
    for (int j = 0; j < N; ++j) acc += s->H_inv[i + (size_t)j * P] * -s->r_dual_aug[j];
    dx[i] = acc;
  }
  if (K > 0) {
    for (int i = 0; i < N; ++i) { /* :352 */
      double acc = 0.0;
      for (int q = 0; q < K; ++q) acc += s->H_inv[i + (size_t)(N + q) * P] * -r_pe[q];
      dx[i] += acc;
    }
    for (int q = 0; q < K; ++q) { /* :354-355 */
      double acc = 0.0;
      for (int j = 0; j < N; ++j) acc += s->H_inv[(N + q) + (size_t)j * P] * s->r_dual_aug[j];
      double acc2 = 0.0;
      for (int t = 0; t < K; ++t) acc2 += s->H_inv[(N + q) + (size_t)(N + t) * P] * r_pe[t];
      dy[q] = acc + acc2;
    }
  }
  back_substitute(s, mu);
}

/* Same system solved directly with the factorisation: [dx; -dy] = H^-1 * -[r_aug; r_pe] */
void orc_solve_for_update_direct(orc_solver* s, double mu) {
  const int N = s->N, K = s->K, P = s->P;
  build_r_dual_aug(s, mu);
  const double* r_pe = yblk(s, s->r);
  for (int i = 0; i < N; ++i) s->work[i] = -s->r_dual_aug[i];
  for (int q = 0; q < K; ++q) s->work[N + q] = -r_pe[q];
  orc_ldlt_solve_inplace(P, s->ldlt_mat, s->ldlt_transp, s->work, 1);
  double* dx = xblk(s, s->delta);
  double* dy = yblk(s, s->delta);
  for (int i = 0; i < N; ++i) dx[i] = s->work[i];
  for (int q = 0; q < K; ++q) dy[q] = -s->work[N + q]; /* py is negated in the solution vector, qp.cc:353 */
  back_substitute(s, mu);
}

/* SolveForUpdateNoInequalities, qp.cc:366-386 */
void orc_solve_no_inequalities(orc_solver* s) {
  const int N = s->N, K = s->K, P = s->P;
  const double* r_d = xblk(s, s->r);
  const double* r_pe = yblk(s, s->r);
  double* dx = xblk(s, s->delta);
  double* dy = yblk(s, s->delta);
  for (int i = 0; i < N; ++i) {
    double acc = 0.0;
    for (int j = 0; j < N; ++j) acc += s->H_inv[i + (size_t)j * P] * -r_d[j];
    dx[i] = acc;
  }
  if (K > 0) {
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
      for (int q = 0; q < K; ++q) acc += s->H_inv[i + (size_t)(N + q) * P] * -r_pe[q];
      dx[i] += acc;
    }
    for (int q = 0; q < K; ++q) {
      double acc = 0.0;
      for (int j = 0; j < N; ++j) acc += s->H_inv[(N + q) + (size_t)j * P] * r_d[j];
      double acc2 = 0.0;
      for (int t = 0; t < K; ++t) acc2 += s->H_inv[(N + q) + (size_t)(N + t) * P] * r_pe[t];
      dy[q] = acc + acc2;
    }
  }
}

/* ---------------------------------------------------------------- ComputeAlpha, qp.cc:485-507 */
double orc_compute_alpha_vec(int n, const double* val, const double* d_val, double tau) {
  double alpha = 1.0;
  for (int i = 0; i < n; ++i) {
    const double updated_val = val[i] + d_val[i];
    if (updated_val <= 0.0 && fabs(d_val[i]) > 0) {
      const double candidate_alpha = -tau * val[i] / d_val[i];
      if (candidate_alpha < alpha) alpha = candidate_alpha;
    }
  }
  return alpha;
}

void orc_compute_alpha(const orc_solver* s, double tau, double* primal, double* dual) {
  orc_solver* ms = (orc_solver*)s;
  *primal = orc_compute_alpha_vec(s->M, sblk(s, ms->variables), sblk(s, ms->delta), tau);
  *dual = orc_compute_alpha_vec(s->M, zblk(s, ms->variables), zblk(s, ms->delta), tau);
}

/* ComputeMu, qp.cc:509-516 */
double orc_compute_mu(const orc_solver* s) {
  if (s->M == 0) return 0.0;
  orc_solver* ms = (orc_solver*)s;
  const double* sv = sblk(s, ms->variables);
  const double* z = zblk(s, ms->variables);
  double acc = 0.0;
  for (int i = 0; i < s->M; ++i) acc += sv[i] * z[i];
  return acc / (double)s->M;
}

/* ComputePredictorCorrectorMuAffine, qp.cc:519-537 */
double orc_compute_mu_affine(const orc_solver* s, double mu, double alpha_p, double alpha_d) {
  orc_solver* ms = (orc_solver*)s;
  const int M = s->M;
  const double* sv = sblk(s, ms->variables);
  const double* z = zblk(s, ms->variables);
  const double* ds = sblk(s, ms->delta_affine);
  const double* dz = zblk(s, ms->delta_affine);
  double s_dz = 0, z_ds = 0, ds_dz = 0;
  for (int i = 0; i < M; ++i) { s_dz += sv[i] * dz[i]; z_ds += z[i] * ds[i]; ds_dz += ds[i] * dz[i]; }
  double mu_affine = mu;
  mu_affine += alpha_d * s_dz / (double)M;
  mu_affine += alpha_p * z_ds / (double)M;
  mu_affine += (alpha_d * alpha_p) * ds_dz / (double)M;
  return mu_affine > 0.0 ? mu_affine : 0.0;
}

/* ComputeErrors, qp.cc:423-437 */
void orc_compute_errors(const orc_solver* s, double mu, orc_kkt_error* out) {
  orc_solver* ms = (orc_solver*)s;
  out->r_dual = out->r_comp = out->r_primal_eq = out->r_primal_ineq = 0.0;
  double acc = 0.0;
  const double* r_d = xblk(s, ms->r);
  for (int i = 0; i < s->N; ++i) acc += r_d[i] * r_d[i];
  out->r_dual = sqrt(acc);
  if (s->K > 0) {
    const double* r_pe = yblk(s, ms->r);
    acc = 0.0;
    for (int i = 0; i < s->K; ++i) acc += r_pe[i] * r_pe[i];
    out->r_primal_eq = sqrt(acc);
  }
  if (s->M > 0) {
    const double* ds = sblk(s, ms->r);
    double sq = 0.0, sum = 0.0;
    for (int i = 0; i < s->M; ++i) { sq += ds[i] * ds[i]; sum += ds[i]; }
    const double corrected = sq - 2 * (sum * mu) + (mu * mu) * (double)s->M; /* :432 */
    out->r_comp = sqrt(corrected > 0. ? corrected : 0.);
    const double* r_pi = zblk(s, ms->r);
    acc = 0.0;
    for (int i = 0; i < s->M; ++i) acc += r_pi[i] * r_pi[i];
    out->r_primal_ineq = sqrt(acc);
  }
}

/* LinearInequalityConstraint::ClampX, qp.hpp:43-53 */
static double clamp_x(double a, double b, double x) {
  if (a < 0) {
    const double lim = b / -a;
    return x < lim ? x : lim;
  } else {
    const double lim = -b / a;
    return x > lim ? x : lim;
  }
}

/* ComputeInitialGuess, qp.cc:439-482 */
int orc_initial_guess(orc_solver* s, const orc_params* p) {
  if (p->initial_guess_method == ORC_GUESS_USER_PROVIDED) return ORC_OK; /* :440-442 */
  double* x = xblk(s, s->variables);
  double* y = yblk(s, s->variables);
  for (int i = 0; i < s->N; ++i) x[i] = 0.0; /* :445-446 */
  for (int i = 0; i < s->K; ++i) y[i] = 0.0;
  if (p->initial_guess_method == ORC_GUESS_SOLVE_EQUALITY_CONSTRAINED) { /* :455-460 */
    const int st = orc_compute_ldlt(s, 0);
    if (st != ORC_OK) return st;
    orc_evaluate_kkt(s, 0);
    orc_solve_no_inequalities(s);
    const double* dx = xblk(s, s->delta);
    const double* dy = yblk(s, s->delta);
    for (int i = 0; i < s->N; ++i) x[i] = dx[i];
    for (int i = 0; i < s->K; ++i) y[i] = dy[i];
  }
  for (int i = 0; i < s->M; ++i) { /* :464-467 */
    const int v = s->qp.cons_var[i];
    x[v] = clamp_x(s->qp.cons_a[i], s->qp.cons_b[i], x[v]);
  }
  double* sv = sblk(s, s->variables);
  double* z = zblk(s, s->variables);
  for (int i = 0; i < s->M; ++i) { /* :470-481 */
    const double s_val = s->qp.cons_a[i] * x[s->qp.cons_var[i]] + s->qp.cons_b[i];
    sv[i] = s_val > 1.0e-9 ? s_val : 1.0e-9;
    z[i] = 1.0 / sv[i];
  }
  return ORC_OK;
}

/* Iterate, qp.cc:153-201 */
int orc_iterate(orc_solver* s, double mu_input, int strategy, orc_ip_outputs* out) {
  orc_evaluate_kkt(s, 1); /* :156 */
  out->mu = mu_input;     /* structs.hpp:53-64 defaults */
  out->alpha_primal = 1.0;
  out->alpha_dual = 1.0;
  out->alpha_probe_primal = NAN;
  out->alpha_probe_dual = NAN;
  out->mu_affine = NAN;
  const int st = orc_compute_ldlt(s, 1); /* :163 */
  if (st != ORC_OK) return st;
  if (s->M == 0) {
    orc_solve_for_update(s, 0.0); /* :165-167 */
  } else if (strategy != ORC_PREDICTOR_CORRECTOR) {
    orc_solve_for_update(s, out->mu); /* :169 */
  } else {
    orc_solve_for_update(s, 0.0); /* :173 */
    orc_compute_alpha(s, 1.0, &out->alpha_probe_primal, &out->alpha_probe_dual); /* :174 */
    memcpy(s->delta_affine, s->delta, sizeof(double) * (size_t)s->V);            /* :177 */
    out->mu_affine = orc_compute_mu_affine(s, out->mu, out->alpha_probe_primal, out->alpha_probe_dual);
    const double sigma = pow(out->mu_affine / out->mu, 3); /* :182 */
    out->mu = sigma * mu_input;                            /* :183 */
    orc_solve_for_update(s, out->mu);                      /* :187 */
  }
  if (s->M > 0) orc_compute_alpha(s, 0.995, &out->alpha_primal, &out->alpha_dual); /* :191-193 */
  /* :196-199 */
  double* v = s->variables;
  const double* d = s->delta;
  for (int i = 0; i < s->N; ++i) v[i] += d[i] * out->alpha_primal;
  for (int i = 0; i < s->M; ++i) v[s->N + i] += d[s->N + i] * out->alpha_primal;
  for (int i = 0; i < s->K; ++i) v[s->N + s->M + i] += d[s->N + s->M + i] * out->alpha_dual;
  for (int i = 0; i < s->M; ++i) v[s->N + s->M + s->K + i] += d[s->N + s->M + s->K + i] * out->alpha_dual;
  return ORC_OK;
}

static double kkt_max(const orc_kkt_error* e) { /* structs.hpp:75-77 */
  double m = e->r_dual;
  if (e->r_comp > m) m = e->r_comp;
  if (e->r_primal_eq > m) m = e->r_primal_eq;
  if (e->r_primal_ineq > m) m = e->r_primal_ineq;
  return m;
}

/* Solve, qp.cc:100-151 */
int orc_solve(orc_solver* s, const orc_params* p, orc_iteration* iterations, int* num_iterations) {
  *num_iterations = 0;
  /* CheckParams, qp.cc:76-82 */
  if (!(p->initial_mu > 0) || !(p->sigma > 0) || !(p->sigma <= 1.0) || !(p->termination_kkt_tol > 0) ||
      !(p->max_iterations > 0)) {
    return -100;
  }
  int st = orc_initial_guess(s, p); /* :106 */
  if (st != ORC_OK) return -st;
  orc_evaluate_kkt(s, 1); /* :110 */
  double mu = p->initialize_mu_with_complementarity ? orc_compute_mu(s) : p->initial_mu; /* :115 */
  for (int iter = 0; iter < p->max_iterations; ++iter) {
    orc_iteration* rec = &iterations[iter];
    orc_compute_errors(s, mu, &rec->kkt_initial);               /* :118 */
    st = orc_iterate(s, mu, p->barrier_strategy, &rec->ip);     /* :122 */
    if (st != ORC_OK) return -st;
    orc_evaluate_kkt(s, 1);                                     /* :125 */
    orc_compute_errors(s, mu, &rec->kkt_final);                 /* :127 */
    *num_iterations = iter + 1;                                 /* :130 */
    if (kkt_max(&rec->kkt_final) < p->termination_kkt_tol &&
        orc_compute_mu(s) < p->termination_complementarity_tol) { /* :132-137 */
      return ORC_SATISFIED_KKT_TOL;
    }
    if (kkt_max(&rec->kkt_final) <= mu || !p->decrease_mu_only_on_small_error) { /* :140-146 */
      if (p->barrier_strategy == ORC_FIXED_DECREASE) mu *= p->sigma;
      else mu = p->sigma * orc_compute_mu(s);
    }
  }
  return ORC_MAX_ITERATIONS; /* :149 */
}

/* The metric's unit of work: qp_test.cc:132-134 + ComputeAlpha(tau) on the caller's state. */
int orc_newton_step(orc_solver* s, const double* vars, double mu, double tau, int use_inverse, double* delta,
                    double* alpha) {
  memcpy(s->variables, vars, sizeof(double) * (size_t)s->V);
  orc_evaluate_kkt(s, 1);
  const int st = orc_compute_ldlt(s, 1);
  if (st != ORC_OK) {
    for (int i = 0; i < s->V; ++i) delta[i] = NAN;
    alpha[0] = alpha[1] = NAN;
    return st;
  }
  if (use_inverse) orc_solve_for_update(s, s->M > 0 ? mu : 0.0);
  else orc_solve_for_update_direct(s, s->M > 0 ? mu : 0.0);
  memcpy(delta, s->delta, sizeof(double) * (size_t)s->V);
  alpha[0] = alpha[1] = 1.0;
  if (s->M > 0) orc_compute_alpha(s, tau, &alpha[0], &alpha[1]);
  return ORC_OK;
}

/* ---------------------------------------------------------------- BuildFullSystem, qp.cc:595-655 */
void orc_build_full_system(const orc_solver* s, double* H, double* r) {
  orc_solver* ms = (orc_solver*)s;
  const int N = s->N, M = s->M, K = s->K;
  const int V = s->V;
  const orc_qp* p = &s->qp;
  const double* x = xblk(s, ms->variables);
  const double* sv = sblk(s, ms->variables);
  const double* y = yblk(s, ms->variables);
  const double* z = zblk(s, ms->variables);
#define HF(i, j) H[(size_t)(i) + (size_t)(j) * (size_t)V]
  memset(H, 0, sizeof(double) * (size_t)V * V);
  memset(r, 0, sizeof(double) * (size_t)V);
  for (int i = 0; i < N; ++i) /* :613 */
    for (int j = 0; j < N; ++j) HF(i, j) = (j <= i) ? p->G[i + (size_t)j * N] : p->G[j + (size_t)i * N];
  for (int q = 0; q < K; ++q) /* :614-617 */
    for (int j = 0; j < N; ++j) {
      HF(j, N + M + q) = p->A_eq[q + (size_t)j * K];
      HF(N + M + q, j) = p->A_eq[q + (size_t)j * K];
    }
  for (int i = 0; i < M; ++i) { /* :619-637 */
    const int v = p->cons_var[i];
    HF(v, N + M + K + i) = p->cons_a[i];  /* topRightCorner(N,M) = A_i^T */
    HF(N + M + K + i, v) = p->cons_a[i];  /* bottomLeftCorner(M,N) = A_i */
    HF(N + M + K + i, N + i) = -1.0;      /* bottomRows(M).middleCols(N,M).diagonal = -1 */
    HF(N + i, N + i) = z[i] / sv[i];      /* Sigma */
    HF(N + i, N + M + K + i) = -1.0;      /* topRows(N+M).bottomRightCorner(M,M).diagonal = -1 */
  }
  /* NOTE: with duplicate variables, A_i(i, v) entries are distinct rows, so no accumulation is needed. */
  double* r_d = r;
  double* s_inv_r_comp = r + N;
  double* r_pe = r + N + M;
  double* r_pi = r + N + M + K;
  for (int i = 0; i < N; ++i) { /* :644 */
    double acc = 0.0;
    for (int j = 0; j < N; ++j) acc += HF(i, j) * x[j];
    r_d[i] = acc + p->c[i];
  }
  for (int i = 0; i < N; ++i) { /* :646 */
    double acc = 0.0;
    for (int q = 0; q < K; ++q) acc += p->A_eq[q + (size_t)i * K] * y[q];
    r_d[i] -= acc;
  }
  for (int q = 0; q < K; ++q) { /* :648 */
    double acc = 0.0;
    for (int j = 0; j < N; ++j) acc += p->A_eq[q + (size_t)j * K] * x[j];
    r_pe[q] = acc + p->b_eq[q];
  }
  for (int i = 0; i < M; ++i) { /* :650-654 */
    const int v = p->cons_var[i];
    r_d[v] -= p->cons_a[i] * z[i];
    s_inv_r_comp[i] = z[i];
    r_pi[i] = p->cons_a[i] * x[v] + p->cons_b[i] - sv[i];
  }
#undef HF
}

/* PartialPivLU (row pivoting) solve, as used by qp_test.cc:120-124 */
int orc_lu_solve(int n, double* A, double* b) {
#define LA(i, j) A[(size_t)(i) + (size_t)(j) * (size_t)n]
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(LA(k, k));
    for (int i = k + 1; i < n; ++i) {
      const double v = fabs(LA(i, k));
      if (v > best) { best = v; piv = i; }
    }
    if (best == 0.0) return 1;
    if (piv != k) {
      for (int j = 0; j < n; ++j) { const double t = LA(k, j); LA(k, j) = LA(piv, j); LA(piv, j) = t; }
      const double t = b[k]; b[k] = b[piv]; b[piv] = t;
    }
    const double d = LA(k, k);
    for (int i = k + 1; i < n; ++i) {
      const double l = LA(i, k) / d;
      LA(i, k) = l;
      if (l != 0.0) {
        for (int j = k + 1; j < n; ++j) LA(i, j) -= l * LA(k, j);
        b[i] -= l * b[k];
      }
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double acc = b[i];
    for (int j = i + 1; j < n; ++j) acc -= LA(i, j) * b[j];
    b[i] = acc / LA(i, i);
  }
#undef LA
  return 0;
}

/* qp_test.cc:120-129: signed_update = LU.solve(-r_full); flip dy and dz */
int orc_full_system_step(const orc_solver* s, double* delta) {
  const int V = s->V;
  double* H = (double*)malloc(sizeof(double) * (size_t)V * V);
  double* r = (double*)malloc(sizeof(double) * (size_t)V);
  orc_build_full_system(s, H, r);
  for (int i = 0; i < V; ++i) r[i] = -r[i];
  const int rc = orc_lu_solve(V, H, r);
  for (int i = 0; i < V; ++i) delta[i] = r[i];
  for (int i = 0; i < s->K; ++i) delta[s->N + s->M + i] *= -1.0;
  for (int i = 0; i < s->M; ++i) delta[s->N + s->M + s->K + i] *= -1.0;
  free(H);
  free(r);
  return rc;
}

/* ---------------------------------------------------------------- batched CPU baseline driver */
int orc_batched_newton_step(int batch, int n, int k, int m, int m_r, const double* J, int row_major,
                            const double* r, double lambda, const double* G, const double* c,
                            const double* A_eq, const double* b_eq, const int* cons_var, const double* cons_a,
                            const double* cons_b, const double* vars, const double* mu, double tau,
                            int use_inverse, int num_threads, double* delta, double* alpha, int* status) {
  const int V = n + 2 * m + k;
  int used = 1;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
  used = num_threads > 0 ? num_threads : omp_get_max_threads();
#else
  (void)num_threads;
#endif
#pragma omp parallel
  {
    double* Gl = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* cl = (double*)malloc(sizeof(double) * (size_t)n);
    /* one solver per thread, re-used across problems like qp_test.cc:531-539 (Setup re-zeroes H_, qp.cc:47) */
    orc_solver s;
    int have_solver = 0;
#pragma omp for schedule(static)
    for (int p = 0; p < batch; ++p) {
      orc_qp qp;
      qp.n = n; qp.k = k; qp.m = m;
      if (J) {
        orc_linearize_dense(m_r, n, J + (size_t)p * m_r * n, row_major, r + (size_t)p * m_r, lambda, Gl, cl);
        qp.G = Gl; qp.c = cl;
      } else {
        qp.G = G + (size_t)p * n * n; qp.c = c + (size_t)p * n;
      }
      qp.A_eq = A_eq ? A_eq + (size_t)p * k * n : NULL;
      qp.b_eq = b_eq ? b_eq + (size_t)p * k : NULL;
      qp.cons_var = cons_var ? cons_var + (size_t)p * m : NULL;
      qp.cons_a = cons_a ? cons_a + (size_t)p * m : NULL;
      qp.cons_b = cons_b ? cons_b + (size_t)p * m : NULL;
      int bad = 0;
      for (int i = 0; i < m; ++i) bad |= (qp.cons_var[i] < 0 || qp.cons_var[i] >= n);
      if (bad) { status[p] = -3; continue; }
      if (!have_solver) {
        if (orc_solver_setup(&s, &qp) != 0) { status[p] = -1; continue; }
        have_solver = 1;
      } else {
        s.qp = qp;
        memset(s.H, 0, sizeof(double) * (size_t)s.P * s.P);
      }
      status[p] = orc_newton_step(&s, vars + (size_t)p * V, mu[p], tau, use_inverse, delta + (size_t)p * V,
                                  alpha + (size_t)p * 2);
    }
    if (have_solver) orc_solver_free(&s);
    free(Gl);
    free(cl);
  }
  return used;
}

int orc_batched_solve(int batch, int n, int k, int m, int m_r, const double* J, int row_major, const double* r,
                      double lambda, const double* G, const double* c, const double* A_eq, const double* b_eq,
                      const int* cons_var, const double* cons_a, const double* cons_b, const orc_params* params,
                      int num_threads, double* vars_io, int* termination, int* num_iterations) {
  const int V = n + 2 * m + k;
  int used = 1;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
  used = num_threads > 0 ? num_threads : omp_get_max_threads();
#else
  (void)num_threads;
#endif
#pragma omp parallel
  {
    double* Gl = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* cl = (double*)malloc(sizeof(double) * (size_t)n);
    orc_iteration* its = (orc_iteration*)malloc(sizeof(orc_iteration) * (size_t)(params->max_iterations > 0 ? params->max_iterations : 1));
    orc_solver s;
    int have_solver = 0;
#pragma omp for schedule(dynamic, 16)
    for (int p = 0; p < batch; ++p) {
      orc_qp qp;
      qp.n = n; qp.k = k; qp.m = m;
      if (J) {
        orc_linearize_dense(m_r, n, J + (size_t)p * m_r * n, row_major, r + (size_t)p * m_r, lambda, Gl, cl);
        qp.G = Gl; qp.c = cl;
      } else {
        qp.G = G + (size_t)p * n * n; qp.c = c + (size_t)p * n;
      }
      qp.A_eq = A_eq ? A_eq + (size_t)p * k * n : NULL;
      qp.b_eq = b_eq ? b_eq + (size_t)p * k : NULL;
      qp.cons_var = cons_var ? cons_var + (size_t)p * m : NULL;
      qp.cons_a = cons_a ? cons_a + (size_t)p * m : NULL;
      qp.cons_b = cons_b ? cons_b + (size_t)p * m : NULL;
      int bad = 0;
      for (int i = 0; i < m; ++i) bad |= (qp.cons_var[i] < 0 || qp.cons_var[i] >= n);
      num_iterations[p] = 0;
      if (bad) { termination[p] = -3; continue; }
      if (!have_solver) {
        if (orc_solver_setup(&s, &qp) != 0) { termination[p] = -1; continue; }
        have_solver = 1;
      } else {
        s.qp = qp;
        memset(s.H, 0, sizeof(double) * (size_t)s.P * s.P);   /* Setup re-zeroes H_ (qp.cc:47) */
      }
      memcpy(s.variables, vars_io + (size_t)p * V, sizeof(double) * (size_t)V);
      int nit = 0;
      termination[p] = orc_solve(&s, params, its, &nit);
      num_iterations[p] = nit;
      memcpy(vars_io + (size_t)p * V, s.variables, sizeof(double) * (size_t)V);
    }
    if (have_solver) orc_solver_free(&s);
    free(its);
    free(Gl);
    free(cl);
  }
  return used;
}
