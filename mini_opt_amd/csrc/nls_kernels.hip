// nls_kernels.hip -- the small per-problem pieces of the SQP outer loop around the QP:
//   shift_constraints_kernel   tail of LinearizeAndFillQP            nonlinear.cc:192-214, qp.hpp:57-65
//   nonlinear_errors_kernel    EvaluateNonlinearErrors               nonlinear.cc:279-293
//   cost_derivative_kernel     ComputeQPCostDerivative (+ dx^T G dx) nonlinear.cc:452-483, 496-498
// One wavefront per problem (four problems per 256-thread workgroup); all of it is HBM-bound streaming with a handful of
// flops per byte, so the only design rule is coalesced rows and a single pass over J / G.
#include "mo_kernels.h"

namespace mo {
namespace {

template <typename T> __device__ inline T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
template <typename T> __device__ inline T sign_of(T x) { return x > (T)0 ? (T)1 : (x < (T)0 ? (T)-1 : (T)0); }  // nonlinear.cc:440-450

// cons_b_out = a * x[var] + b ; out2 = {f (left to the linearisation kernel), |b_eq|_1}
template <typename T>
__global__ __launch_bounds__(256) void shift_constraints_kernel(const AuxArgs a) {
  const int lane = threadIdx.x & 63;
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= a.batch) return;
  const T* x = (const T*)a.x + p * a.x_stride;
  bool bad = false;
  for (int i = lane; i < a.m; i += 64) {
    const int v = a.cons_var[p * a.cons_stride + i];
    const T ca = ((const T*)a.cons_a)[p * a.cons_stride + i], cb = ((const T*)a.cons_b)[p * a.cons_stride + i];
    const bool ok = v >= 0 && v < a.n;
    bad = bad || !ok;
    ((T*)a.cons_b_out)[p * a.cons_b_out_stride + i] = ok ? ca * x[v] + cb : (T)__builtin_nan("");  // qp.hpp:57-59
    if (a.cons_var_out) a.cons_var_out[p * a.cons_b_out_stride + i] = v;
    if (a.cons_a_out) ((T*)a.cons_a_out)[p * a.cons_b_out_stride + i] = ca;
  }
  T l1 = 0;
  if (a.out2 && a.b) {  // mo_nls_solve takes both error terms from nonlinear_errors_kernel and passes neither pointer
    for (int i = lane; i < a.k; i += 64) l1 += fabs(((const T*)a.b)[p * a.b_stride + i]);                // nonlinear.cc:203
    l1 = wave_sum(l1);
  }
  const bool any_bad = __any(bad);
  if (lane == 0) {
    if (a.out2) ((T*)a.out2)[2 * p + 1] = l1;
    if (a.status) a.status[p] = any_bad ? MO_STATUS_BAD_INDEX : MO_STATUS_OK;
  }
}

// out2 = {0.5 |r|^2, |r_eq|_1}
template <typename T>
__global__ __launch_bounds__(256) void nonlinear_errors_kernel(const AuxArgs a) {
  const int lane = threadIdx.x & 63;
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= a.batch) return;
  T sq = 0, l1 = 0;
  const T* r = (const T*)a.r + p * a.r_stride;
  for (int i = lane; i < a.m_r; i += 64) sq += r[i] * r[i];
  for (int i = lane; i < a.k; i += 64) l1 += fabs(((const T*)a.b)[p * a.b_stride + i]);
  sq = wave_sum(sq); l1 = wave_sum(l1);
  if (lane == 0) { ((T*)a.out2)[2 * p] = (T)0.5 * sq; ((T*)a.out2)[2 * p + 1] = l1; }
}

// out2 = {c^T dx, sum_i sign(b_i) (A dx)_i} ; quad_out = dx^T G dx.
// J-level: c^T dx = r^T (J dx) and dx^T G dx = |J dx|^2 + lambda |dx|^2, one pass over J (row-major rows are coalesced;
// a column-major J is walked column by column with the row sums kept per lane).
template <typename T>
__global__ __launch_bounds__(256) void cost_derivative_kernel(const AuxArgs a) {
  const int lane = threadIdx.x & 63;
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= a.batch) return;
  const int n = a.n;
  const T* dx = (const T*)a.x + p * a.x_stride;
  T d_f = 0, quad = 0;
  if (a.J) {
    const T* J = (const T*)a.J + p * a.J_stride;
    const T* r = (const T*)a.r + p * a.r_stride;
    if (a.J_row_major) {
      for (int q = 0; q < a.m_r; ++q) {
        T t = 0;
        for (int i = lane; i < n; i += 64) t += J[(size_t)q * a.J_ld + i] * dx[i];
        t = wave_sum(t);
        d_f += r[q] * t; quad += t * t;   // uniform accumulators
      }
    } else {
      for (int q0 = 0; q0 < a.m_r; q0 += 64) {
        const int q = q0 + lane;
        T t = 0;
        if (q < a.m_r)
          for (int i = 0; i < n; ++i) t += J[(size_t)i * a.J_ld + q] * dx[i];
        d_f += wave_sum(q < a.m_r ? r[q] * t : (T)0); quad += wave_sum(t * t);
      }
    }
    T dd = 0;
    for (int i = lane; i < n; i += 64) dd += dx[i] * dx[i];
    dd = wave_sum(dd);
    const T lam = a.lambda_vec ? ((const T*)a.lambda_vec)[p * a.lambda_vec_stride] : (T)a.lambda;
    if (lam > (T)0) quad += lam * dd;                                                                  // nonlinear.cc:187-189
  } else {
    const T* G = (const T*)a.G + p * a.G_stride;
    const T* c = (const T*)a.c + p * a.c_stride;
    T t = 0;
    for (int i = lane; i < n; i += 64) t += c[i] * dx[i];                                              // nonlinear.cc:471
    d_f = wave_sum(t);
    T qd = 0;  // dx^T sym(G) dx from the lower triangle (selfadjointView<Lower>, nonlinear.cc:497)
    for (int col = 0; col < n; ++col) {
      T s = 0;
      for (int i = col + lane; i < n; i += 64) s += G[(size_t)col * a.G_ld + i] * dx[i] * (i == col ? (T)1 : (T)2);
      qd += s * dx[col];
    }
    quad = wave_sum(qd);
  }
  T d_eq = 0;
  for (int i = lane; i < a.k; i += 64) {  // one equality row per lane: A is k x n column-major, reads along i are contiguous
    const T* A = (const T*)a.A + p * a.A_stride;
    T t = 0;
    for (int jn = 0; jn < n; ++jn) t += A[(size_t)jn * a.A_ld + i] * dx[jn];
    d_eq += sign_of(((const T*)a.b)[p * a.b_stride + i]) * t;                                          // nonlinear.cc:478-481
  }
  d_eq = wave_sum(d_eq);
  if (lane == 0) {
    ((T*)a.out2)[2 * p] = d_f; ((T*)a.out2)[2 * p + 1] = d_eq;
    if (a.quad_out) ((T*)a.quad_out)[p] = quad;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The per-problem state machine of ConstrainedNonlinearLeastSquares::Solve (nonlinear.cc:75-158) for mo_nls_solve.
// One wavefront per problem: lane 0 takes the scalar decisions, all lanes move the n-vectors.
__device__ inline double total_of(const double* e, double penalty) { return e[0] + penalty * e[1]; }  // Errors::Total, structs.hpp:177

__global__ __launch_bounds__(256) void nls_init_kernel(const NlsArgs a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.batch) return;
  double* sd = a.sd + p * NLS_SD; int* si = a.si + p * NLS_SI;
  sd[NLS_SD_LAMBDA] = a.prm.lambda_initial;              // nonlinear.cc:92
  sd[NLS_SD_PENALTY] = a.prm.equality_penalty_initial;   // nonlinear.cc:93
  si[NLS_SI_TERM] = -1; si[NLS_SI_STATE] = 0; si[NLS_SI_LS_RESULT] = -1; si[NLS_SI_NSTEPS] = 0; si[NLS_SI_NITER] = 0;
  if (a.status) a.status[p] = MO_STATUS_OK;
}

// RetractCandidateVars (nonlinear.cc:160-168) for the built-in retractions; MO_RETRACT_CALLBACK hands (dx, alpha) to the caller instead.
__device__ inline double mod_pi(double v) {  // math::ModPi: wrap into [-pi, pi)
  const double two_pi = 6.283185307179586476925286766559, pi = 3.141592653589793238462643383279;
  return v - two_pi * floor((v + pi) / two_pi);
}
// LANES = 64: one wavefront per problem (lane 0 takes the scalar decisions, all lanes move the n-vectors); LANES = 1: one THREAD per
// problem for small n -- 64 times fewer waves, and the "still searching / still active" counters are bumped once per wave instead of
// once per problem (one word sustains ~88 M atomics/s: 65 536 single increments were 0.75 ms of a 0.75 ms launch).
template <int LANES> __device__ inline void count_in(int* counter, bool inc) {
  if (LANES == 64) { if (inc) atomicAdd(counter, 1); return; }
  const unsigned long long mask = __ballot(inc);
  if (inc && (int)(threadIdx.x & 63) == __builtin_ctzll(mask)) atomicAdd(counter, __builtin_popcountll(mask));
}
template <int LANES>
__device__ inline void retract_candidate(const NlsArgs& a, long long p, double alpha, bool first, int lane) {
  const double* x = a.vars + p * a.vars_stride; const double* dx = a.qp_vars + p * a.qp_vars_stride;
  double* c = a.cand + p * a.cand_stride;
  if (a.prm.retraction == MO_RETRACT_CALLBACK) {
    if (first) {
      double* st = a.step + p * a.step_stride;
      for (int i = lane; i < a.n; i += LANES) st[i] = dx[i];
    }
    if (lane == 0) a.step_alpha[p] = alpha;
    return;
  }
  const bool wrap = a.prm.retraction == MO_RETRACT_WRAP_PI;
  for (int i = lane; i < a.n; i += LANES) {
    const double v = x[i] + dx[i] * alpha;
    c[i] = wrap ? mod_pi(v) : v;
  }
}

// After the QP: penalty (nonlinear.cc:108-115, 485-500), directional derivative, first trial point alpha = 1 (:363, :160-168)
template <int LANES>
__global__ __launch_bounds__(256) void nls_begin_search_kernel(const NlsArgs a) {
  const int lane = LANES == 64 ? (int)(threadIdx.x & 63) : 0;
  const long long p = LANES == 64 ? (long long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.batch) return;
  double* sd = a.sd + p * NLS_SD; int* si = a.si + p * NLS_SI;
  if (si[NLS_SI_TERM] >= 0) return;  // wave-uniform
  const int qp_status = a.qp_status[p];
  double* rec = a.iterations ? a.iterations + ((size_t)p * a.prm.max_iterations + a.iter) * a.rec : nullptr;
  const double* e0 = a.errors_pre + 2 * p;
  const double d_f = a.deriv[2 * p], d_eq = a.deriv[2 * p + 1];
  double penalty = sd[NLS_SD_PENALTY];
  if (qp_status == MO_STATUS_OK && a.k > 0) {
    double new_penalty;
    if (a.m == 0) {  // the reference runs QPNullSpaceSolver here and has no multipliers: inequality (18.36), nonlinear.cc:491-499
      const double l1 = fmax(e0[1], 2.220446049250313e-16);
      const double q = d_f + 0.5 * fmax(0.0, a.quad[p]);
      new_penalty = q / ((1.0 - a.prm.equality_penalty_rho) * l1);
    } else {
      new_penalty = a.lagrange[2 * p + 1];                 // l_infinity of y, nonlinear.cc:488-490
    }
    if (new_penalty > penalty) penalty = new_penalty * a.prm.equality_penalty_scale_factor;
  }
  bool searching = false;
  if (lane == 0) {
    if (rec) {
      for (int i = 0; i < a.rec; ++i) rec[i] = __builtin_nan("");
      rec[1] = sd[NLS_SD_LAMBDA]; rec[2] = e0[0]; rec[3] = e0[1]; rec[4] = d_f; rec[5] = d_eq; rec[6] = penalty;
      rec[8] = 0; rec[9] = a.qp_term[p]; rec[10] = a.qp_nit[p]; rec[11] = qp_status;
    }
    if (qp_status != MO_STATUS_OK) {                       // the reference throws here (qp.cc:285, 303-307)
      // with equalities and no inequalities the reference's null-space solver reports a reduced Hessian it cannot factorise and
      // Solve returns QP_INDEFINITE without logging the iteration (nonlinear.cc:103-105)
      const bool nullspace_path = a.m == 0 && a.k > 0;
      si[NLS_SI_TERM] = nullspace_path ? MO_NLS_QP_INDEFINITE : MO_NLS_QP_FAILURE;
      si[NLS_SI_NITER] = nullspace_path ? a.iter : a.iter + 1;
      if (a.status) a.status[p] = qp_status;
      if (rec) rec[0] = si[NLS_SI_STATE];
    } else {
      sd[NLS_SD_PENALTY] = penalty;
      sd[NLS_SD_DIRECTIONAL] = d_f + penalty * d_eq;       // DirectionalDerivatives::Total, structs.hpp:197
      sd[NLS_SD_ALPHA] = 1.0;
      si[NLS_SI_LS_RESULT] = -1; si[NLS_SI_NSTEPS] = 0;
      searching = true;
    }
  }
  if (qp_status != MO_STATUS_OK) return;
  retract_candidate<LANES>(a, p, 1.0, true, lane);                                                 // alpha = 1, :363
  count_in<LANES>(a.counters, searching);
}

// One evaluation of the line search (nonlinear.cc:378-407) and, if it goes on, the next alpha (:364-376, 414-438)
template <int LANES>
__global__ __launch_bounds__(256) void nls_search_step_kernel(const NlsArgs a) {
  const int lane = LANES == 64 ? (int)(threadIdx.x & 63) : 0;
  const long long p = LANES == 64 ? (long long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.batch) return;
  double* sd = a.sd + p * NLS_SD; int* si = a.si + p * NLS_SI;
  if (si[NLS_SI_TERM] >= 0 || si[NLS_SI_LS_RESULT] >= 0) return;
  const double* e0 = a.errors_pre + 2 * p; const double* e1 = a.errors_step + 2 * p;
  const double penalty = sd[NLS_SD_PENALTY], directional = sd[NLS_SD_DIRECTIONAL], alpha = sd[NLS_SD_ALPHA];
  const double d_f = a.deriv[2 * p], d_eq = a.deriv[2 * p + 1];
  const double phi0 = total_of(e0, penalty), phi1 = total_of(e1, penalty);
  int result = -1;
  double next_alpha = alpha;
  const bool finite = isfinite(e1[0]) && isfinite(e1[1]);
  if (!finite) result = MO_LS_FAILURE_NON_FINITE_COST;                                            // :383-385
  else if (fmax(fabs(d_f), fabs(d_eq)) < a.prm.absolute_first_derivative_tol) result = MO_LS_FIRST_ORDER_SATISFIED;  // :387-389
  else if (directional > 0) result = MO_LS_POSITIVE_DERIVATIVE;                                   // :390-393
  else if (phi1 <= phi0 + directional * alpha * 1.0e-4) result = MO_LS_SUCCESS;                   // armijo_c1, :118, :396-400
  else if (a.ls >= a.prm.max_line_search_iterations) result = MO_LS_MAX_ITERATIONS;               // :403
  else if (a.prm.line_search_strategy == MO_POLYNOMIAL_APPROXIMATION) {
    bool valid = true;
    if (a.ls == 0) {                                       // QuadraticApproxMinimum, :524-531
      const double numerator = phi1 - directional * alpha - phi0;
      if (directional > 0 || numerator <= 0) valid = false;
      else next_alpha = -directional * alpha * alpha / (2.0 * numerator);
    } else {                                               // CubicApproxCoeffs + CubicApproxMinimum, :558-603
      const double a0 = sd[NLS_SD_A1], t0 = sd[NLS_SD_T1];  // second last step
      const double a1 = alpha, t1 = phi1;                    // last step
      const double m00 = a0 * a0 * a0, m01 = a0 * a0, m10 = a1 * a1 * a1, m11 = a1 * a1;
      const double r0 = t0 - phi0 - directional * a0, r1 = t1 - phi0 - directional * a1;
      const double det = m00 * m11 - m01 * m10;
      const double ca = (m11 * r0 - m01 * r1) / det, cb = (-m10 * r0 + m00 * r1) / det;
      const double arg = cb * cb - 3 * ca * directional;
      if (ca == 0.0 || arg < -1.0e-12) valid = false;
      else next_alpha = (-cb + sqrt(fmax(arg, 0.0))) / (3 * ca);
    }
    if (!valid || !isfinite(next_alpha) || next_alpha <= 0.0 || next_alpha >= alpha) result = MO_LS_FAILURE_INVALID_ALPHA;  // :369-372
  } else {
    next_alpha = alpha * a.prm.armijo_search_tau;          // :377-380
  }
  bool searching = false;
  if (lane == 0) {
    const int ns = si[NLS_SI_NSTEPS];
    if (a.iterations) {
      double* rec = a.iterations + ((size_t)p * a.prm.max_iterations + a.iter) * a.rec;
      rec[MO_NLS_ITER_HEADER + 3 * ns] = alpha; rec[MO_NLS_ITER_HEADER + 3 * ns + 1] = e1[0]; rec[MO_NLS_ITER_HEADER + 3 * ns + 2] = e1[1];
    }
    si[NLS_SI_NSTEPS] = ns + 1;
    sd[NLS_SD_A2] = sd[NLS_SD_A1]; sd[NLS_SD_T2] = sd[NLS_SD_T1];
    sd[NLS_SD_A1] = alpha; sd[NLS_SD_T1] = phi1;
    if (result >= 0) si[NLS_SI_LS_RESULT] = result;
    else { sd[NLS_SD_ALPHA] = next_alpha; searching = true; }
  }
  if (result >= 0) return;
  retract_candidate<LANES>(a, p, next_alpha, false, lane);                                        // RetractCandidateVars, :160-168
  count_in<LANES>(a.counters, searching);
}

// UpdateLambdaAndCheckExitConditions (nonlinear.cc:296-339) + the bookkeeping of the outer loop (:121-157)
template <int LANES>
__global__ __launch_bounds__(256) void nls_update_kernel(const NlsArgs a) {
  const int lane = LANES == 64 ? (int)(threadIdx.x & 63) : 0;
  const long long p = LANES == 64 ? (long long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.batch) return;
  double* sd = a.sd + p * NLS_SD; int* si = a.si + p * NLS_SI;
  const bool was_active = si[NLS_SI_TERM] < 0;
  bool still_active = false;
  if (was_active) {
    const int result = si[NLS_SI_LS_RESULT];
    const double* e0 = a.errors_pre + 2 * p;
    const double penalty = sd[NLS_SD_PENALTY];
    double lambda = sd[NLS_SD_LAMBDA];
    int state = si[NLS_SI_STATE], term = -1;
    if (result == MO_LS_SUCCESS) {
      double* x = a.vars + p * a.vars_stride; const double* c = a.cand + p * a.cand_stride;
      for (int i = lane; i < a.n; i += LANES) x[i] = c[i];                                         // variables_.swap(candidate_vars_), :303
      lambda = fmax(lambda * (state == 1 ? a.prm.lambda_decrease_on_restore : a.prm.lambda_decrease_on_success), a.prm.min_lambda);
      state = 0;
      const double* e1 = a.errors_step + 2 * p;            // the accepted step is the last one evaluated
      if (fmax(e1[0], e1[1]) < a.prm.absolute_exit_tol) term = MO_NLS_SATISFIED_ABSOLUTE_TOL;      // :314-316
      else if (total_of(e1, penalty) > total_of(e0, penalty) * (1 - a.prm.relative_exit_tol)) term = MO_NLS_SATISFIED_RELATIVE_TOL;
    } else if (result == MO_LS_FIRST_ORDER_SATISFIED) {
      term = MO_NLS_SATISFIED_FIRST_ORDER_TOL;                                                    // :321-323
    } else if (result == MO_LS_MAX_ITERATIONS || result == MO_LS_POSITIVE_DERIVATIVE) {
      if (state == 0) { lambda = fmax(a.prm.lambda_failure_init, lambda * 10.0); state = 1; }     // :326-329
      else lambda *= 10.0;                                                                        // :332
      if (lambda > a.prm.max_lambda) term = MO_NLS_MAX_LAMBDA;                                    // :334-337
    }
    if (lane == 0) {
      sd[NLS_SD_LAMBDA] = lambda; si[NLS_SI_STATE] = state; si[NLS_SI_NITER] = a.iter + 1;
      if (a.iterations) {
        double* rec = a.iterations + ((size_t)p * a.prm.max_iterations + a.iter) * a.rec;
        rec[0] = state; rec[7] = result; rec[8] = si[NLS_SI_NSTEPS];
      }
      if (term >= 0) si[NLS_SI_TERM] = term;
      else still_active = true;
    }
  }
  if (lane == 0) {
    const int t = si[NLS_SI_TERM];
    if (a.termination) a.termination[p] = t >= 0 ? t : MO_NLS_MAX_ITERATIONS;                     // :157
    if (a.num_iterations) a.num_iterations[p] = si[NLS_SI_NITER];
  }
  count_in<LANES>(a.counters + 1, still_active);
}

// SetUserExitCallback (nonlinear.cc:142-149): a problem that is still active and whose flag the callback set ends with USER_CALLBACK
__global__ __launch_bounds__(256) void nls_user_exit_kernel(const NlsArgs a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.batch) return;
  int* si = a.si + p * NLS_SI;
  if (si[NLS_SI_TERM] < 0 && a.user_exit[p] != 0) {
    si[NLS_SI_TERM] = MO_NLS_USER_CALLBACK;
    if (a.termination) a.termination[p] = MO_NLS_USER_CALLBACK;
    atomicSub(a.counters + 1, 1);
  }
}

__global__ __launch_bounds__(256) void nullspace_termination_kernel(const AuxArgs a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p < a.batch) a.status[p] = a.status[p] != MO_STATUS_OK ? MO_NULLSPACE_NOT_POSITIVE_DEFINITE : MO_NULLSPACE_SUCCESS;
}

template <typename K64, typename K32>
hipError_t launch_aux(K64 k64, K32 k32, const AuxArgs& a, int dtype, hipStream_t stream) {
  if (a.batch <= 0) return hipSuccess;
  const dim3 gd((unsigned)((a.batch + 3) / 4)), bd(256);
  if (dtype == MO_F64) hipLaunchKernelGGL(k64, gd, bd, 0, stream, a);
  else hipLaunchKernelGGL(k32, gd, bd, 0, stream, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_shift_constraints(const AuxArgs& a, int dtype, hipStream_t stream) {
  return launch_aux(shift_constraints_kernel<double>, shift_constraints_kernel<float>, a, dtype, stream);
}
hipError_t launch_nonlinear_errors(const AuxArgs& a, int dtype, hipStream_t stream) {
  return launch_aux(nonlinear_errors_kernel<double>, nonlinear_errors_kernel<float>, a, dtype, stream);
}
hipError_t launch_cost_derivative(const AuxArgs& a, int dtype, hipStream_t stream) {
  return launch_aux(cost_derivative_kernel<double>, cost_derivative_kernel<float>, a, dtype, stream);
}

hipError_t launch_nullspace_termination(const AuxArgs& a, hipStream_t stream) {
  if (a.batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(nullspace_termination_kernel, dim3((unsigned)((a.batch + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}
hipError_t launch_nls_init(const NlsArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(nls_init_kernel, dim3((unsigned)((a.batch + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}
// one thread per problem up to 32 variables, one wavefront per problem beyond
#define MO_NLS_LAUNCH(KERNEL)                                                                                              \
  do {                                                                                                                     \
    if (a.n <= 32) hipLaunchKernelGGL(KERNEL<1>, dim3((unsigned)((a.batch + 255) / 256)), dim3(256), 0, stream, a);        \
    else hipLaunchKernelGGL(KERNEL<64>, dim3((unsigned)((a.batch + 3) / 4)), dim3(256), 0, stream, a);                     \
  } while (0)
hipError_t launch_nls_begin_search(const NlsArgs& a, hipStream_t stream) {
  MO_NLS_LAUNCH(nls_begin_search_kernel);
  return hipGetLastError();
}
hipError_t launch_nls_search_step(const NlsArgs& a, hipStream_t stream) {
  MO_NLS_LAUNCH(nls_search_step_kernel);
  return hipGetLastError();
}
hipError_t launch_nls_update(const NlsArgs& a, hipStream_t stream) {
  MO_NLS_LAUNCH(nls_update_kernel);
  return hipGetLastError();
}
#undef MO_NLS_LAUNCH
hipError_t launch_nls_user_exit(const NlsArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(nls_user_exit_kernel, dim3((unsigned)((a.batch + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mo

// ---------------------------------------------------------------------------------------------------------------------
// Device residual families: the residual functions of the reference's own NLS tests as kernels, so that a
// caller's mo_nls_solve callback can be two launches instead of a PCIe round trip.  One thread per (problem, residual row);
// J is written dense (row-major m_r x n, or column-major rows x n for equality stacks = QP::A_eq), zeros included.
namespace mo {
namespace {

template <typename T> __device__ inline void put(T* J, int row, int col, int rows, int ld, bool row_major, T v) {
  J[row_major ? (size_t)row * ld + col : (size_t)col * ld + row] = v;
  (void)rows;
}

template <typename T>
__global__ __launch_bounds__(256) void residual_family_kernel(int family, int n, int rows, long long batch, const T* prm, const T* x,
                                                              long long x_stride, T* r, long long r_stride, T* J, long long J_stride,
                                                              int J_ld, int row_major) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= batch * rows) return;
  const long long p = idx / rows;
  const int q = (int)(idx - p * rows);
  const T* xp = x + p * x_stride;
  T* Jp = J ? J + p * J_stride : nullptr;
  if (Jp)
    for (int c = 0; c < n; ++c) put(Jp, q, c, rows, J_ld, row_major != 0, (T)0);
  T val = 0;
  switch (family) {
    case MO_RESIDUAL_ROSENBROCK: {  // nonlinear_test.cc:375-386 (n = 2), 502-521 (chain): rows 2i, 2i+1 for i < n-1
      const int i = q >> 1;
      if (q & 1) {
        val = (T)10 * (xp[i + 1] - xp[i] * xp[i]);
        if (Jp) { put(Jp, q, i, rows, J_ld, row_major != 0, (T)-20 * xp[i]); put(Jp, q, i + 1, rows, J_ld, row_major != 0, (T)10); }
      } else {
        val = (T)1 - xp[i];
        if (Jp) put(Jp, q, i, rows, J_ld, row_major != 0, (T)-1);
      }
      break;
    }
    case MO_RESIDUAL_HIMMELBLAU: {  // nonlinear_test.cc:578-593
      if (q == 0) {
        val = xp[0] * xp[0] + xp[1] - (T)11;
        if (Jp) { put(Jp, 0, 0, rows, J_ld, row_major != 0, (T)2 * xp[0]); put(Jp, 0, 1, rows, J_ld, row_major != 0, (T)1); }
      } else {
        val = xp[0] + xp[1] * xp[1] - (T)7;
        if (Jp) { put(Jp, 1, 0, rows, J_ld, row_major != 0, (T)1); put(Jp, 1, 1, rows, J_ld, row_major != 0, (T)2 * xp[1]); }
      }
      break;
    }
    case MO_RESIDUAL_SPHERE: {      // nonlinear_test.cc:722-730: h(x) = x
      val = xp[q];
      if (Jp) put(Jp, q, q, rows, J_ld, row_major != 0, (T)1);
      break;
    }
    case MO_RESIDUAL_PRODUCT_PAIRS: {  // nonlinear_test.cc:737-743: x_{2q} x_{2q+1} - v_q
      val = xp[2 * q] * xp[2 * q + 1] - prm[q];
      if (Jp) { put(Jp, q, 2 * q, rows, J_ld, row_major != 0, xp[2 * q + 1]); put(Jp, q, 2 * q + 1, rows, J_ld, row_major != 0, xp[2 * q]); }
      break;
    }
    default: break;
  }
  r[p * r_stride + q] = val;
}


// MO_RESIDUAL_ACTUATOR_CHAIN: the reference's kinematic chains (test/transform_chains.cc) as a device residual family.  One thread
// per problem walks the parameter block (wave-uniform scalar loads): per chain the link poses (ActuatorLink::Compute, :125-158), the
// chain products and their derivatives (ComputeChain, :23-82) and the chain rule onto the active parameters (ActuatorChain::Update,
// :202-243); then every residual row  const + sum lin_i x_i + sum_c w_c . t_c  with its Jacobian.
constexpr int kChainMaxLinks = 8;

template <typename T> __device__ inline void mat3_mul(const T (&A)[9], const T (&B)[9], T (&C)[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

template <typename T>
__global__ __launch_bounds__(64) void actuator_chain_kernel(int n, int rows, long long batch, const T* prm, const T* x, long long x_stride, T* r,
                                                            long long r_stride, T* J, long long J_stride, int J_ld, int row_major) {
  const long long p = (long long)blockIdx.x * 64 + threadIdx.x;
  if (p >= batch) return;
  const T* xp = x + p * x_stride;
  T* rp = r + p * r_stride;
  T* Jp = J ? J + p * J_stride : nullptr;
  if (Jp)
    for (int q = 0; q < rows; ++q)
      for (int c = 0; c < n; ++c) put(Jp, q, c, rows, J_ld, row_major != 0, (T)0);
  const int C = (int)prm[0];
  // the row table starts behind the chains
  int row_table = 1;
  for (int c = 0; c < C; ++c) row_table += 1 + 12 * (int)prm[row_table];
  {  // constant and linear parts
    int pos = row_table;
    for (int q = 0; q < rows; ++q) {
      T val = prm[pos++];
      const int nl = (int)prm[pos++];
      for (int l = 0; l < nl; ++l) {
        const int ix = (int)prm[pos]; const T cf = prm[pos + 1]; pos += 2;
        val += cf * xp[ix];
        if (Jp) { const size_t o = row_major ? (size_t)q * J_ld + ix : (size_t)ix * J_ld + q; Jp[o] += cf; }
      }
      const int nt = (int)prm[pos++];
      pos += 4 * nt;
      rp[q] = val;
    }
  }
  int cpos = 1;
  for (int c = 0; c < C; ++c) {
    const int Ln = (int)prm[cpos];
    const T* lk = prm + cpos + 1;
    cpos += 1 + 12 * Ln;
    T R[kChainMaxLinks][9], t[kChainMaxLinks][3], D[kChainMaxLinks][9], tend[kChainMaxLinks + 1][3];
    for (int i = 0; i < Ln && i < kChainMaxLinks; ++i) {  // ActuatorLink::Compute: substitute the active parameters, R = Rx Ry Rz
      T v[6];
#pragma unroll
      for (int d = 0; d < 6; ++d) { const int ix = (int)lk[12 * i + 6 + d]; v[d] = ix >= 0 ? xp[ix] : lk[12 * i + d]; }
      const T cx = cos(v[0]), sx = sin(v[0]), cy = cos(v[1]), sy = sin(v[1]), cz = cos(v[2]), sz = sin(v[2]);
      const T Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
      T Ryz[9];
      mat3_mul(Ry, Rz, Ryz);
      mat3_mul(Rx, Ryz, R[i]);
      // right-tangent derivative of R wrt (x, y, z): (Ry Rz)^T e_x | Rz^T e_y | e_z   (rotation_D_angles)
      D[i][0] = Ryz[0]; D[i][3] = Ryz[1]; D[i][6] = Ryz[2];
      D[i][1] = Rz[3];  D[i][4] = Rz[4];  D[i][7] = Rz[5];
      D[i][2] = 0;      D[i][5] = 0;      D[i][8] = 1;
      t[i][0] = v[3]; t[i][1] = v[4]; t[i][2] = v[5];
    }
    tend[Ln][0] = tend[Ln][1] = tend[Ln][2] = 0;           // i_t_end, transform_chains.cc:47-52
    for (int i = Ln - 1; i >= 0; --i)
#pragma unroll
      for (int a = 0; a < 3; ++a) tend[i][a] = R[i][3 * a] * tend[i + 1][0] + R[i][3 * a + 1] * tend[i + 1][1] + R[i][3 * a + 2] * tend[i + 1][2] + t[i][a];
    // rows that look at this chain
    T S[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};                  // start_R_i, :56-61
    for (int i = 0; i < Ln; ++i) {
      T Sn[9];
      mat3_mul(S, R[i], Sn);
      for (int d = 0; d < 6; ++d) {
        const int ix = (int)lk[12 * i + 6 + d];
        if (ix < 0) continue;
        T dt[3];                                           // d(effector translation) / d(parameter)
        if (d >= 3) {                                      // translation_D_translation column, :56-61, 231-241
          dt[0] = S[d - 3]; dt[1] = S[3 + d - 3]; dt[2] = S[6 + d - 3];
        } else if (i < Ln - 1) {                           // start_R_[i+1] [-[i+1]_t_N]_x rot_D_angles[:, d], :66-73, 221-222
          const T v0 = D[i][d], v1 = D[i][3 + d], v2 = D[i][6 + d];
          const T* te = tend[i + 1];
          const T w0 = v1 * te[2] - v2 * te[1], w1 = v2 * te[0] - v0 * te[2], w2 = v0 * te[1] - v1 * te[0];   // v x t = [-t]_x v
          dt[0] = Sn[0] * w0 + Sn[1] * w1 + Sn[2] * w2; dt[1] = Sn[3] * w0 + Sn[4] * w1 + Sn[5] * w2; dt[2] = Sn[6] * w0 + Sn[7] * w1 + Sn[8] * w2;
        } else {
          dt[0] = dt[1] = dt[2] = 0;                       // the last link's rotation does not move the effector, :73
        }
        if (!Jp) continue;
        int pos = row_table;
        for (int q = 0; q < rows; ++q) {
          pos += 1; const int nl = (int)prm[pos++]; pos += 2 * nl;
          const int nt = (int)prm[pos++];
          for (int e = 0; e < nt; ++e, pos += 4) {
            if ((int)prm[pos] != c) continue;
            const T g = prm[pos + 1] * dt[0] + prm[pos + 2] * dt[1] + prm[pos + 3] * dt[2];
            const size_t o = row_major ? (size_t)q * J_ld + ix : (size_t)ix * J_ld + q;
            Jp[o] += g;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 9; ++e) S[e] = Sn[e];
    }
    int pos = row_table;                                   // values: w . t_effector with t_effector = 0_t_N
    for (int q = 0; q < rows; ++q) {
      pos += 1; const int nl = (int)prm[pos++]; pos += 2 * nl;
      const int nt = (int)prm[pos++];
      for (int e = 0; e < nt; ++e, pos += 4)
        if ((int)prm[pos] == c) rp[q] += prm[pos + 1] * tend[0][0] + prm[pos + 2] * tend[0][1] + prm[pos + 3] * tend[0][2];
    }
  }
}

}  // namespace

int residual_family_rows(int family, int n, int rows_hint) {
  switch (family) {
    case MO_RESIDUAL_ROSENBROCK: return n >= 2 ? 2 * (n - 1) : -1;
    case MO_RESIDUAL_HIMMELBLAU: return n == 2 ? 2 : -1;
    case MO_RESIDUAL_SPHERE: return n;
    case MO_RESIDUAL_PRODUCT_PAIRS: return (rows_hint >= 1 && 2 * rows_hint <= n) ? rows_hint : -1;
    case MO_RESIDUAL_ACTUATOR_CHAIN: return rows_hint >= 1 ? rows_hint : -1;   // the rows are described by the parameter block
    default: return -1;
  }
}

hipError_t launch_residual_family(int family, int n, int rows, long long batch, int dtype, const void* prm, const void* x,
                                  long long x_stride, void* r, long long r_stride, void* J, long long J_stride, int J_ld,
                                  int row_major, hipStream_t stream) {
  if (batch <= 0 || rows <= 0) return hipSuccess;
  if (family == MO_RESIDUAL_ACTUATOR_CHAIN) {
    const dim3 cg((unsigned)((batch + 63) / 64)), cb(64);
    if (dtype == MO_F64)
      hipLaunchKernelGGL(actuator_chain_kernel<double>, cg, cb, 0, stream, n, rows, batch, (const double*)prm, (const double*)x, x_stride,
                         (double*)r, r_stride, (double*)J, J_stride, J_ld, row_major);
    else
      hipLaunchKernelGGL(actuator_chain_kernel<float>, cg, cb, 0, stream, n, rows, batch, (const float*)prm, (const float*)x, x_stride,
                         (float*)r, r_stride, (float*)J, J_stride, J_ld, row_major);
    return hipGetLastError();
  }
  const long long total = batch * rows;
  const dim3 gd((unsigned)((total + 255) / 256)), bd(256);
  if (dtype == MO_F64)
    hipLaunchKernelGGL(residual_family_kernel<double>, gd, bd, 0, stream, family, n, rows, batch, (const double*)prm, (const double*)x,
                       x_stride, (double*)r, r_stride, (double*)J, J_stride, J_ld, row_major);
  else
    hipLaunchKernelGGL(residual_family_kernel<float>, gd, bd, 0, stream, family, n, rows, batch, (const float*)prm, (const float*)x,
                       x_stride, (float*)r, r_stride, (float*)J, J_stride, J_ld, row_major);
  return hipGetLastError();
}

}  // namespace mo
