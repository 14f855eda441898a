// mo_api.hip -- implementation of the C ABI declared in include/mini_opt_hip.h.
// Validation mirrors QPInteriorPointSolver::Setup (qp.cc:20-73) and CheckParams (qp.cc:76-82); everything numeric
// happens in the gfx950 kernels (kkt_generic.hip, kkt_fused.hip).  There is NO CPU fallback: without a usable HIP
// device every entry point fails with MO_ERR_NO_DEVICE / MO_ERR_HIP.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "mo_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define MO_HIP_CHECK(expr)                                                                     \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess) return fail(MO_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

}  // namespace

struct mo_plan {
  mo_plan_desc desc;
  int num_cus;
  int elem;
  size_t generic_lds;
  // scratch for mo_qp_solve with J-level input: G [max_batch][n*n], c [max_batch][n]
  void* G_scratch;
  void* c_scratch;
  unsigned long long* ticket;  // device work counter of the fused kernels (zeroed on the stream before each launch)
};

namespace {

int check_plan(const mo_plan* plan) {
  if (!plan) return fail(MO_ERR_INVALID_ARGUMENT, "plan is NULL");
  return MO_OK;
}

// Dimension / pointer checks shared by every batched entry point (the F_ASSERTs of qp.cc:21-34).
int fill_problem(const mo_plan* plan, const mo_problem* prob, int64_t batch, bool need_cost, bool need_constraints,
                 mo::KernelArgs* a) {
  if (!prob) return fail(MO_ERR_INVALID_ARGUMENT, "Must pass a non-null problem");
  if (batch < 0) return fail(MO_ERR_INVALID_ARGUMENT, "batch must be >= 0");
  const mo_plan_desc& d = plan->desc;
  memset(a, 0, sizeof(*a));
  a->n = d.n; a->k = d.k; a->m = d.m; a->m_r = 0;
  a->batch = batch;
  if (need_cost) {
    if (prob->J) {
      if (d.m_r <= 0) return fail(MO_ERR_DIMENSION, "J given but the plan was created with m_r = 0");
      if (!prob->r) return fail(MO_ERR_INVALID_ARGUMENT, "J given without r");
      if (prob->J_layout != MO_ROW_MAJOR && prob->J_layout != MO_COL_MAJOR)
        return fail(MO_ERR_INVALID_ARGUMENT, "bad J_layout");
      const int min_ld = prob->J_layout == MO_ROW_MAJOR ? d.n : d.m_r;
      if (prob->J_ld < min_ld) return fail(MO_ERR_DIMENSION, "J_ld %d < %d", prob->J_ld, min_ld);
      a->J = prob->J; a->J_stride = prob->J_stride; a->J_ld = prob->J_ld; a->J_row_major = prob->J_layout == MO_ROW_MAJOR;
      a->r = prob->r; a->r_stride = prob->r_stride; a->lambda = prob->lambda; a->m_r = d.m_r;
      a->lambda_vec = prob->lambda_vec; a->lambda_vec_stride = prob->lambda_stride;
    } else {
      if (!prob->G || !prob->c) return fail(MO_ERR_INVALID_ARGUMENT, "need either (J, r) or (G, c)");
      if (prob->G_ld < d.n) return fail(MO_ERR_DIMENSION, "G must be square: G_ld %d < n %d", prob->G_ld, d.n);
      a->G = prob->G; a->G_stride = prob->G_stride; a->G_ld = prob->G_ld;
      a->c = prob->c; a->c_stride = prob->c_stride;
    }
  }
  if (d.k > 0) {
    if (!prob->A_eq || !prob->b_eq) return fail(MO_ERR_DIMENSION, "Rows of A_e and b_e must match (k = %d but NULL given)", d.k);
    if (prob->A_ld < d.k) return fail(MO_ERR_DIMENSION, "A_ld %d < k %d", prob->A_ld, d.k);
    a->A = prob->A_eq; a->A_stride = prob->A_stride; a->A_ld = prob->A_ld;
    a->b = prob->b_eq; a->b_stride = prob->b_stride;
  }
  if (d.m > 0 && need_constraints) {
    if (!prob->cons_var || !prob->cons_a || !prob->cons_b)
      return fail(MO_ERR_INVALID_ARGUMENT, "m = %d but constraint arrays are NULL", d.m);
    a->cons_var = prob->cons_var; a->cons_a = prob->cons_a; a->cons_b = prob->cons_b; a->cons_stride = prob->cons_stride;
  }
  return MO_OK;
}

int launch(const mo_plan* plan, const mo::KernelArgs& a_in, void* stream) {
  if (a_in.batch == 0) return MO_OK;
  mo::KernelArgs a = a_in;
  a.ticket = plan->ticket;
  MO_HIP_CHECK(hipSetDevice(plan->desc.device));
  hipStream_t s = (hipStream_t)stream;
  const bool force_generic = (plan->desc.flags & MO_PLAN_FORCE_GENERIC) != 0;
  if (!force_generic && mo::fused_supported(a, plan->desc.dtype)) {
    MO_HIP_CHECK(mo::launch_fused(a, plan->desc.dtype, plan->num_cus, s));
  } else {
    MO_HIP_CHECK(mo::launch_generic(a, plan->desc.dtype, plan->num_cus, s));
  }
  return MO_OK;
}

}  // namespace

extern "C" {

const char* mo_version_string(void) { return "mini_opt_hip 0.1 (gfx950)"; }

const char* mo_status_string(int32_t status) {
  switch (status) {
    case MO_STATUS_OK: return "OK";
    case MO_STATUS_NONPOSITIVE_SLACK: return "NONPOSITIVE_SLACK";
    case MO_STATUS_FACTORIZATION_FAILED: return "FACTORIZATION_FAILED";
    case MO_STATUS_NONFINITE: return "NONFINITE";
    case MO_STATUS_BAD_INDEX: return "BAD_INDEX";
    default: return "UNKNOWN";
  }
}

const char* mo_last_error(void) { return g_err; }

void mo_default_solve_params(mo_solve_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->initial_mu = 1.0;  // qp.hpp:134-164
  p->sigma = 0.5;
  p->termination_kkt_tol = 1.0e-9;
  p->termination_complementarity_tol = 1.0e-6;
  p->max_iterations = 10;
  p->barrier_strategy = MO_COMPLEMENTARITY;
  p->decrease_mu_only_on_small_error = 0;
  p->initial_guess_method = MO_GUESS_NAIVE;
  p->initialize_mu_with_complementarity = 0;
}

int mo_plan_create(const mo_plan_desc* desc, mo_plan** out) {
  g_err[0] = 0;
  if (!desc || !out) return fail(MO_ERR_INVALID_ARGUMENT, "desc/out is NULL");
  *out = nullptr;
  if (desc->n <= 0 || desc->k < 0 || desc->m < 0 || desc->m_r < 0)
    return fail(MO_ERR_DIMENSION, "bad dimensions n=%d k=%d m=%d m_r=%d", desc->n, desc->k, desc->m, desc->m_r);
  if (desc->dtype != MO_F64 && desc->dtype != MO_F32) return fail(MO_ERR_UNSUPPORTED, "unknown dtype %d", desc->dtype);
  if (desc->n + desc->k > 192) return fail(MO_ERR_UNSUPPORTED, "n + k = %d exceeds the LDS-resident limit of 192", desc->n + desc->k);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(MO_ERR_NO_DEVICE, "no HIP device available (the HIP path has no CPU fallback)");
  if (desc->device < 0 || desc->device >= ndev) return fail(MO_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", desc->device, ndev);
  MO_HIP_CHECK(hipSetDevice(desc->device));
  hipDeviceProp_t prop;
  MO_HIP_CHECK(hipGetDeviceProperties(&prop, desc->device));
  mo_plan* p = new (std::nothrow) mo_plan();
  if (!p) return fail(MO_ERR_HIP, "out of host memory");
  p->desc = *desc;
  p->num_cus = prop.multiProcessorCount;
  p->elem = desc->dtype == MO_F64 ? 8 : 4;
  p->G_scratch = nullptr;
  p->c_scratch = nullptr;
  p->ticket = nullptr;
  if (hipMalloc((void**)&p->ticket, 256) != hipSuccess) {
    delete p;
    return fail(MO_ERR_HIP, "hipMalloc of the work counter failed");
  }
  mo::KernelArgs a;
  memset(&a, 0, sizeof(a));
  a.n = desc->n; a.k = desc->k; a.m = desc->m; a.m_r = desc->m_r;
  p->generic_lds = mo::generic_lds_bytes(a, p->elem);
  if (p->generic_lds > 160 * 1024) {
    const size_t need = p->generic_lds;
    (void)hipFree(p->ticket);
    delete p;
    return fail(MO_ERR_UNSUPPORTED, "problem needs %zu B of LDS (> 160 KiB)", need);
  }
  *out = p;
  return MO_OK;
}

int mo_plan_destroy(mo_plan* plan) {
  if (!plan) return MO_OK;
  if (plan->G_scratch) (void)hipFree(plan->G_scratch);
  if (plan->c_scratch) (void)hipFree(plan->c_scratch);
  if (plan->ticket) (void)hipFree(plan->ticket);
  delete plan;
  return MO_OK;
}

const char* mo_plan_step_kernel(const mo_plan* plan, const mo_problem* prob) {
  if (!plan || !prob) return "invalid";
  mo::KernelArgs a;
  if (fill_problem(plan, prob, 1, true, true, &a) != MO_OK) return "invalid";
  a.mode = mo::MODE_STEP;
  a.vars = a.delta = reinterpret_cast<void*>(16);  // layout query only: assume 16-byte aligned state / output
  a.ticket = plan->ticket;
  if (!(plan->desc.flags & MO_PLAN_FORCE_GENERIC) && mo::fused_supported(a, plan->desc.dtype)) return mo::fused_name(a, plan->desc.dtype);
  return "generic";
}

int mo_linearize(mo_plan* plan, const mo_problem* prob, int64_t batch, void* G_out, int64_t G_stride, int32_t G_ld,
                 void* c_out, int64_t c_stride, void* half_sq_out, void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  mo::KernelArgs a;
  if (prob && !prob->J) return fail(MO_ERR_INVALID_ARGUMENT, "mo_linearize needs J-level input");
  // A_eq / constraints are not touched by the linearisation of the cost
  mo_plan tmp = *plan;
  tmp.desc.k = 0; tmp.desc.m = 0;
  if (int rc = fill_problem(&tmp, prob, batch, true, false, &a)) return rc;
  if (!G_out || !c_out) return fail(MO_ERR_INVALID_ARGUMENT, "G_out / c_out is NULL");
  if (G_ld < plan->desc.n) return fail(MO_ERR_DIMENSION, "G_ld %d < n", G_ld);
  a.mode = mo::MODE_LINEARIZE;
  a.G_out = G_out; a.G_out_stride = G_stride; a.G_out_ld = G_ld;
  a.c_out = c_out; a.c_out_stride = c_stride; a.half_sq_out = half_sq_out;
  return launch(&tmp, a, stream);
}

int mo_fill_qp(mo_plan* plan, const mo_problem* prob, int64_t batch, const void* x, int64_t x_stride, void* G_out,
               int64_t G_stride, int32_t G_ld, void* c_out, int64_t c_stride, void* cons_b_out, int64_t cons_b_stride,
               void* errors_out, int32_t* status, void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  if (!prob || !prob->J) return fail(MO_ERR_INVALID_ARGUMENT, "mo_fill_qp needs the cost residual stack (J, r)");
  if (!errors_out) return fail(MO_ERR_INVALID_ARGUMENT, "errors_out is NULL");
  const mo_plan_desc& d = plan->desc;
  if (d.m > 0 && (!x || !cons_b_out)) return fail(MO_ERR_INVALID_ARGUMENT, "x / cons_b_out is NULL");
  mo::KernelArgs ka;
  if (int rc = fill_problem(plan, prob, batch, true, true, &ka)) return rc;  // validates A_eq / b_eq / constraints too
  // cost part: G = J^T J + lambda I, c = J^T r, f = 0.5 |r|^2 (nonlinear.cc:182-189) -> errors_out[2 p]
  mo_plan tmp = *plan;
  tmp.desc.k = 0; tmp.desc.m = 0;
  mo::KernelArgs a;
  if (int rc = fill_problem(&tmp, prob, batch, true, false, &a)) return rc;
  if (!G_out || !c_out) return fail(MO_ERR_INVALID_ARGUMENT, "G_out / c_out is NULL");
  if (G_ld < d.n) return fail(MO_ERR_DIMENSION, "G_ld %d < n", G_ld);
  a.mode = mo::MODE_LINEARIZE;
  a.G_out = G_out; a.G_out_stride = G_stride; a.G_out_ld = G_ld;
  a.c_out = c_out; a.c_out_stride = c_stride; a.half_sq_out = errors_out; a.half_sq_stride = 2;
  if (int rc = launch(&tmp, a, stream)) return rc;
  if (batch == 0) return MO_OK;
  // tail: shifted constraints and the equality L1 norm (nonlinear.cc:192-212)
  mo::AuxArgs x_args;
  memset(&x_args, 0, sizeof(x_args));
  x_args.n = d.n; x_args.k = d.k; x_args.m = d.m; x_args.batch = batch;
  x_args.x = x; x_args.x_stride = x_stride;
  x_args.b = ka.b; x_args.b_stride = ka.b_stride;
  x_args.cons_var = ka.cons_var; x_args.cons_a = ka.cons_a; x_args.cons_b = ka.cons_b; x_args.cons_stride = ka.cons_stride;
  x_args.cons_b_out = cons_b_out; x_args.cons_b_out_stride = cons_b_stride;
  x_args.out2 = errors_out; x_args.status = status;
  MO_HIP_CHECK(mo::launch_shift_constraints(x_args, d.dtype, (hipStream_t)stream));
  return MO_OK;
}

int mo_nonlinear_errors(mo_plan* plan, const void* r, int64_t r_stride, const void* r_eq, int64_t r_eq_stride,
                        int64_t batch, void* errors_out, void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  const mo_plan_desc& d = plan->desc;
  if (batch < 0) return fail(MO_ERR_INVALID_ARGUMENT, "batch must be >= 0");
  if (!errors_out) return fail(MO_ERR_INVALID_ARGUMENT, "errors_out is NULL");
  if (d.m_r > 0 && !r) return fail(MO_ERR_INVALID_ARGUMENT, "r is NULL");
  if (d.k > 0 && !r_eq) return fail(MO_ERR_DIMENSION, "k = %d but r_eq is NULL", d.k);
  mo::AuxArgs a;
  memset(&a, 0, sizeof(a));
  a.n = d.n; a.k = d.k; a.m = d.m; a.m_r = d.m_r; a.batch = batch;
  a.r = r; a.r_stride = r_stride; a.b = r_eq; a.b_stride = r_eq_stride; a.out2 = errors_out;
  MO_HIP_CHECK(hipSetDevice(d.device));
  MO_HIP_CHECK(mo::launch_nonlinear_errors(a, d.dtype, (hipStream_t)stream));
  return MO_OK;
}

int mo_qp_cost_derivative(mo_plan* plan, const mo_problem* prob, int64_t batch, const void* dx, int64_t dx_stride,
                          void* deriv_out, void* quad_out, void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  const mo_plan_desc& d = plan->desc;
  mo::KernelArgs ka;
  if (int rc = fill_problem(plan, prob, batch, true, false, &ka)) return rc;
  if (!dx || !deriv_out) return fail(MO_ERR_INVALID_ARGUMENT, "dx / deriv_out is NULL");  // F_ASSERT_EQ(qp.c.rows(), dx.rows())
  mo::AuxArgs a;
  memset(&a, 0, sizeof(a));
  a.n = d.n; a.k = d.k; a.m = d.m; a.m_r = ka.m_r; a.batch = batch;
  a.x = dx; a.x_stride = dx_stride;
  a.J = ka.J; a.J_stride = ka.J_stride; a.J_ld = ka.J_ld; a.J_row_major = ka.J_row_major; a.r = ka.r; a.r_stride = ka.r_stride;
  a.G = ka.G; a.G_stride = ka.G_stride; a.G_ld = ka.G_ld; a.c = ka.c; a.c_stride = ka.c_stride;
  a.A = ka.A; a.A_stride = ka.A_stride; a.A_ld = ka.A_ld; a.b = ka.b; a.b_stride = ka.b_stride;
  a.lambda = ka.lambda; a.lambda_vec = ka.lambda_vec; a.lambda_vec_stride = ka.lambda_vec_stride;
  a.out2 = deriv_out; a.quad_out = quad_out;
  MO_HIP_CHECK(hipSetDevice(d.device));
  MO_HIP_CHECK(mo::launch_cost_derivative(a, d.dtype, (hipStream_t)stream));
  return MO_OK;
}

int mo_kkt_residual(mo_plan* plan, const mo_problem* prob, int64_t batch, const void* vars, int64_t vars_stride,
                    const void* mu, int64_t mu_stride, uint32_t flags, void* r_out, int64_t r_stride, void* kkt_out,
                    void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  mo::KernelArgs a;
  if (int rc = fill_problem(plan, prob, batch, true, true, &a)) return rc;
  if (!vars || !r_out) return fail(MO_ERR_INVALID_ARGUMENT, "vars / r_out is NULL");
  a.mode = mo::MODE_RESIDUAL;
  a.flags = flags;
  a.vars = const_cast<void*>(vars); a.vars_stride = vars_stride;
  a.mu = mu; a.mu_stride = mu_stride;
  a.r_out = r_out; a.r_out_stride = r_stride; a.kkt_out = kkt_out;
  return launch(plan, a, stream);
}

int mo_newton_step(mo_plan* plan, const mo_problem* prob, int64_t batch, const void* vars, int64_t vars_stride,
                   const void* mu, int64_t mu_stride, double tau, uint32_t flags, void* delta, int64_t delta_stride,
                   void* alpha, int32_t* status, void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  mo::KernelArgs a;
  if (int rc = fill_problem(plan, prob, batch, true, true, &a)) return rc;
  if (!vars || !delta) return fail(MO_ERR_INVALID_ARGUMENT, "vars / delta is NULL");
  if (flags & ~MO_STEP_NO_INEQUALITIES) return fail(MO_ERR_INVALID_ARGUMENT, "unsupported flags 0x%x for mo_newton_step", flags);
  if (!(tau > 0) || !(tau <= 1)) return fail(MO_ERR_INVALID_ARGUMENT, "tau must be in (0, 1]");  // qp.cc:494-495
  if (plan->desc.m > 0 && !mu && !(flags & MO_STEP_NO_INEQUALITIES)) return fail(MO_ERR_INVALID_ARGUMENT, "mu is NULL");
  a.mode = mo::MODE_STEP;
  a.flags = flags;
  a.vars = const_cast<void*>(vars); a.vars_stride = vars_stride;
  a.mu = mu; a.mu_stride = mu_stride; a.tau = tau;
  a.barrier_strategy = MO_COMPLEMENTARITY;
  a.delta = delta; a.delta_stride = delta_stride; a.alpha = alpha; a.status = status;
  return launch(plan, a, stream);
}

int mo_iterate(mo_plan* plan, const mo_problem* prob, int64_t batch, void* vars, int64_t vars_stride, const void* mu,
               int64_t mu_stride, int32_t barrier_strategy, void* delta, int64_t delta_stride, void* ip_out,
               int32_t* status, void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  mo::KernelArgs a;
  if (int rc = fill_problem(plan, prob, batch, true, true, &a)) return rc;
  if (!vars) return fail(MO_ERR_INVALID_ARGUMENT, "vars is NULL");
  if (barrier_strategy < MO_COMPLEMENTARITY || barrier_strategy > MO_PREDICTOR_CORRECTOR)
    return fail(MO_ERR_INVALID_ARGUMENT, "bad barrier_strategy %d", barrier_strategy);
  if (plan->desc.m > 0 && !mu) return fail(MO_ERR_INVALID_ARGUMENT, "mu is NULL");
  a.mode = mo::MODE_ITERATE;
  a.vars = vars; a.vars_stride = vars_stride;
  a.mu = mu; a.mu_stride = mu_stride; a.tau = 0.995;  // qp.cc:192
  a.barrier_strategy = barrier_strategy;
  a.delta = delta; a.delta_stride = delta_stride; a.ip_out = ip_out; a.status = status;
  return launch(plan, a, stream);
}

int mo_qp_solve(mo_plan* plan, const mo_problem* prob, int64_t batch, const mo_solve_params* params, void* vars,
                int64_t vars_stride, int32_t* termination, int32_t* num_iterations, void* iterations, void* lagrange,
                int32_t* status, void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  if (!params) return fail(MO_ERR_INVALID_ARGUMENT, "params is NULL");
  // CheckParams, qp.cc:76-82
  if (!(params->initial_mu > 0)) return fail(MO_ERR_INVALID_ARGUMENT, "initial_mu must be > 0");
  if (!(params->sigma > 0) || !(params->sigma <= 1.0)) return fail(MO_ERR_INVALID_ARGUMENT, "sigma must be in (0, 1]");
  if (!(params->termination_kkt_tol > 0)) return fail(MO_ERR_INVALID_ARGUMENT, "termination_kkt_tol must be > 0");
  if (!(params->max_iterations > 0)) return fail(MO_ERR_INVALID_ARGUMENT, "max_iterations must be > 0");
  if (params->barrier_strategy < MO_COMPLEMENTARITY || params->barrier_strategy > MO_PREDICTOR_CORRECTOR)
    return fail(MO_ERR_INVALID_ARGUMENT, "bad barrier_strategy");
  if (params->initial_guess_method < MO_GUESS_NAIVE || params->initial_guess_method > MO_GUESS_USER_PROVIDED)
    return fail(MO_ERR_INVALID_ARGUMENT, "bad initial_guess_method");
  mo::KernelArgs a;
  if (int rc = fill_problem(plan, prob, batch, true, true, &a)) return rc;
  if (!vars) return fail(MO_ERR_INVALID_ARGUMENT, "vars is NULL");
  a.mode = mo::MODE_SOLVE;
  a.vars = vars; a.vars_stride = vars_stride;
  a.sp = *params;
  a.termination = termination; a.num_iterations = num_iterations; a.iterations = iterations; a.lagrange = lagrange;
  a.status = status;
  a.ticket = plan->ticket;
  const bool use_fused = !(plan->desc.flags & MO_PLAN_FORCE_GENERIC) && mo::fused_supported(a, plan->desc.dtype);
  if (a.J && !use_fused) {  // the generic loop re-reads G after every factorisation: keep the linearised G, c in plan scratch
    if (batch > plan->desc.max_batch) return fail(MO_ERR_INVALID_ARGUMENT, "batch %lld > plan max_batch %lld", (long long)batch, (long long)plan->desc.max_batch);
    const size_t n = (size_t)plan->desc.n;
    MO_HIP_CHECK(hipSetDevice(plan->desc.device));
    if (!plan->G_scratch) {
      MO_HIP_CHECK(hipMalloc(&plan->G_scratch, (size_t)plan->desc.max_batch * n * n * plan->elem));
      MO_HIP_CHECK(hipMalloc(&plan->c_scratch, (size_t)plan->desc.max_batch * n * plan->elem));
    }
    a.G_out = plan->G_scratch; a.G_out_stride = (long long)(n * n); a.G_out_ld = (int)n;
    a.c_out = plan->c_scratch; a.c_out_stride = (long long)n;
  }
  return launch(plan, a, stream);  // fused Solve kernel for J-level n = 32 / 64 fp64 problems, generic kernel otherwise
}

}  // extern "C"
