// mo_api.hip -- implementation of the C ABI declared in include/mini_opt_hip.h.
// Validation mirrors QPInteriorPointSolver::Setup (qp.cc:20-73) and CheckParams (qp.cc:76-82); everything numeric
// happens in the gfx950 kernels (kkt_generic.hip, kkt_fused.hip).  There is NO CPU fallback: without a usable HIP
// device every entry point fails with MO_ERR_NO_DEVICE / MO_ERR_HIP.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
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
  // generic kernel beyond its LDS-resident range: one H workspace per workgroup of the persistent grid.  Sized and allocated ONCE, in
  // mo_plan_create, for every launch shape the plan can produce (the full system and the k = m = 0 system of mo_linearize / mo_fill_qp);
  // launches only read these three fields -- no allocation, no synchronisation, nothing plan-owned changes on the launch path.
  void* H_work;
  size_t H_work_slot_bytes; // bytes per workgroup slot (also the eigenvalue kernel's n x n fp64 matrix beyond the LDS)
  long long H_work_slots;   // workgroup slots allocated: a launch's grid is clamped to it
  void* tile_scratch;  // fused Solve: per wave slot of the persistent grid, the G tiles a wave cannot park in LDS between passes
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

// Which kernel serves a call is decided ONCE, here, from the plan flags, the process-wide MO_FORCE_GENERIC knob (read a single
// time: the scratch a call prepares and the kernel that consumes it must come from the same decision) and the kernels' own
// shape predicates.
enum KernelChoice { KERNEL_FUSED_F64, KERNEL_FUSED_F32, KERNEL_GENERIC };

// The product library reads NO environment variable (the reference has none): kernel selection and scheduling follow the
// plan flags of include/mini_opt_hip.h only.  The getenv knobs of the A/B scripts exist in builds with -DMO_TUNING (tools/ab_build.sh).
bool generic_forced(const mo_plan* plan) {
#ifdef MO_TUNING
  static const bool env_force_generic = getenv("MO_FORCE_GENERIC") != nullptr;  // A/B and bisection knob
  if (env_force_generic) return true;
#endif
  return (plan->desc.flags & MO_PLAN_FORCE_GENERIC) != 0;
}

// Launches of at most this many problems per wave of the persistent grid are split statically (KernelArgs::static_rounds): -1 = the launcher's
// own choice per kernel family (launch_fused / launch_fused_f32); MO_PLAN_TICKETS_ALWAYS -> 0, MO_PLAN_STATIC_ROUNDS_ALWAYS -> every launch
int fused_static_rounds(const mo_plan* plan) {
#ifdef MO_TUNING
  static const int v = [] { const char* e = getenv("MO_FUSED_STATIC_ROUNDS"); return e ? atoi(e) : -1; }();
  if (v >= 0) return v;
#endif
  if (plan->desc.flags & MO_PLAN_TICKETS_ALWAYS) return 0;
  if (plan->desc.flags & MO_PLAN_STATIC_ROUNDS_ALWAYS) return 1 << 30;
  return -1;
}

KernelChoice choose_kernel(const mo_plan* plan, const mo::KernelArgs& a) {
  if (!generic_forced(plan)) {
    if (mo::fused_supported(a, plan->desc.dtype)) return KERNEL_FUSED_F64;
    if (mo::fused_f32_supported(a, plan->desc.dtype)) return KERNEL_FUSED_F32;
  }
  return KERNEL_GENERIC;
}

int launch_chosen(const mo_plan* plan, const mo::KernelArgs& a_in, KernelChoice choice, void* stream) {
  if (a_in.batch == 0) return MO_OK;
  mo::KernelArgs a = a_in;
  a.ticket = plan->ticket; a.static_rounds = fused_static_rounds(plan); a.no_tiny = (plan->desc.flags & MO_PLAN_NO_TINY) != 0;
  MO_HIP_CHECK(hipSetDevice(plan->desc.device));
  hipStream_t s = (hipStream_t)stream;
  if (choice == KERNEL_FUSED_F64) {
    MO_HIP_CHECK(mo::launch_fused(a, plan->desc.dtype, plan->num_cus, s));
  } else if (choice == KERNEL_FUSED_F32) {
    MO_HIP_CHECK(mo::launch_fused_f32(a, plan->num_cus, s));
  } else {
    if (mo::generic_needs_large(a, plan->elem)) {  // H in a global workspace per workgroup: plan-owned, allocated by mo_plan_create
      const size_t need = mo::generic_large_lds_bytes(a, plan->elem);
      if (need > 160 * 1024)
        return fail(MO_ERR_UNSUPPORTED, "n = %d, k = %d, m = %d: not even the state / residual vectors of one problem fit the 160 KiB of LDS", a.n, a.k, a.m);
      if (!plan->H_work || mo::generic_large_workspace_elems(a) * plan->elem > plan->H_work_slot_bytes)
        return fail(MO_ERR_UNSUPPORTED, "the plan owns no H workspace for n = %d, k = %d (created for n = %d, k = %d)", a.n, a.k, plan->desc.n, plan->desc.k);
      a.H_work = plan->H_work; a.H_work_stride = (long long)(plan->H_work_slot_bytes / plan->elem); a.H_work_slots = plan->H_work_slots;
    }
    MO_HIP_CHECK(mo::launch_generic(a, plan->desc.dtype, plan->num_cus, s));
  }
  return MO_OK;
}

int launch(const mo_plan* plan, const mo::KernelArgs& a_in, void* stream) {
  mo::KernelArgs a = a_in;
  a.ticket = plan->ticket; a.static_rounds = fused_static_rounds(plan); a.no_tiny = (plan->desc.flags & MO_PLAN_NO_TINY) != 0;
  return launch_chosen(plan, a, choose_kernel(plan, a), stream);
}

}  // namespace

namespace {
// scratch of one mo_nls_solve call, released on every exit path
struct NlsScratch {  // one allocation carved into 256-byte aligned pieces
  char* base = nullptr;
  size_t used = 0, capacity = 0;
  ~NlsScratch() { if (base) (void)hipFree(base); }
  static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
  template <typename T> void reserve(size_t elems) { capacity += padded((elems ? elems : 1) * sizeof(T)); }
  hipError_t commit() { return hipMalloc((void**)&base, capacity ? capacity : 256); }
  template <typename T> hipError_t alloc(T** out, size_t elems) {
    const size_t bytes = padded((elems ? elems : 1) * sizeof(T));
    if (!base || used + bytes > capacity) return hipErrorOutOfMemory;
    *out = reinterpret_cast<T*>(base + used);
    used += bytes;
    return hipSuccess;
  }
};
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
    case MO_STATUS_NOT_POSITIVE_DEFINITE: return "NOT_POSITIVE_DEFINITE";
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
  if (desc->max_batch < 0) return fail(MO_ERR_INVALID_ARGUMENT, "max_batch must be >= 0 (got %lld)", (long long)desc->max_batch);
  {
    // Any size Setup accepts (qp.cc:36-48 resizes to any N, K): beyond the fused kernels and the LDS-resident generic kernel (n + k <= 192,
    // H in LDS) the generic kernel keeps H in a global workspace; only a problem whose state / residual vectors alone exceed the LDS is refused.
    mo::KernelArgs sz;
    memset(&sz, 0, sizeof(sz));
    sz.n = desc->n; sz.k = desc->k; sz.m = desc->m; sz.m_r = desc->m_r;
    const int elem = desc->dtype == MO_F64 ? 8 : 4;
    if (mo::generic_needs_large(sz, elem) && mo::generic_large_lds_bytes(sz, elem) > 160 * 1024)
      return fail(MO_ERR_UNSUPPORTED, "n = %d, k = %d, m = %d: the state / residual vectors of one problem exceed the 160 KiB of LDS", desc->n, desc->k, desc->m);
  }
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
  p->tile_scratch = nullptr;
  p->H_work = nullptr;
  p->H_work_slot_bytes = 0;
  p->H_work_slots = 0;
  p->ticket = nullptr;
  if (hipMalloc((void**)&p->ticket, 256) != hipSuccess) {
    delete p;
    return fail(MO_ERR_HIP, "hipMalloc of the work counter failed");
  }
  mo::KernelArgs a;
  memset(&a, 0, sizeof(a));
  a.n = desc->n; a.k = desc->k; a.m = desc->m; a.m_r = desc->m_r;
  p->generic_lds = mo::generic_lds_bytes(a, p->elem);
  // fp64 systems up to n = 128 (k <= 63, m <= 256) run on the fused kernels even when the LDS-resident generic kernel cannot hold them.
  // Beyond the LDS-resident range the generic kernel keeps H in a global workspace per workgroup of its persistent grid: everything that
  // sizes it is known here (shape, dtype, CU count), so it is allocated now -- for the larger of the two systems a plan launches there, the
  // full one and the k = m = 0 one of mo_linearize / mo_fill_qp (fewer LDS bytes per workgroup, hence possibly MORE workgroups per CU) --
  // and never touched again.  max_batch > 0 bounds the number of slots (a launch's grid never exceeds its batch).
  // (mo_qp_eigenvalue_stats keeps its n x n fp64 matrix in the same slots once it exceeds the LDS, n > ~139)
  if (mo::generic_needs_large(a, p->elem) || mo::eig_needs_global(desc->n)) {
    mo::KernelArgs lin = a;
    lin.k = 0; lin.m = 0;
    long long slots = 0;
    size_t slot_bytes = 0;
    if (mo::generic_needs_large(a, p->elem)) {
      slots = mo::generic_large_grid(a, p->elem, p->num_cus);
      if (desc->m_r > 0 && mo::generic_needs_large(lin, p->elem)) {
        const long long ls = mo::generic_large_grid(lin, p->elem, p->num_cus);
        if (ls > slots) slots = ls;
      }
      slot_bytes = mo::generic_large_workspace_elems(a) * p->elem;   // (n + k) x ld: covers the n x ld(n) of the k = 0 system
    }
    if (mo::eig_needs_global(desc->n)) {
      const long long es = mo::eig_grid(desc->n, p->num_cus);
      if (es > slots) slots = es;
      if (mo::eig_workspace_bytes(desc->n) > slot_bytes) slot_bytes = mo::eig_workspace_bytes(desc->n);
    }
    if (desc->max_batch > 0 && slots > desc->max_batch) slots = desc->max_batch;
    slot_bytes = (slot_bytes + 255) & ~(size_t)255;
    const size_t bytes = (size_t)slots * slot_bytes;
    if (hipMalloc(&p->H_work, bytes) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(p->ticket);
      delete p;
      return fail(MO_ERR_HIP, "hipMalloc of the %zu B workspace of H (n = %d, k = %d: %lld workgroup slots) failed", bytes, desc->n, desc->k, slots);
    }
    p->H_work_slot_bytes = slot_bytes;
    p->H_work_slots = slots;
  }
  // Tile park of the fused fp64 Solve kernels beyond the 32 grid (qp_solve_impl): one slot per wave of the persistent grid (one workgroup
  // per CU, at most twelve waves each), 22.5 KB at n = 64, 78 KB at n = 128.  Optional -- without it the kernels rebuild the tiles every
  // pass -- so a failed allocation is not an error.
  if (desc->dtype == MO_F64 && desc->n > 32 && desc->n <= 128 && desc->k <= 63 && desc->m <= 256) {
    const int nt = desc->n > 96 ? 8 : desc->n > 64 ? 6 : 4;
    const size_t per_slot = (size_t)(nt * (nt + 1) / 2) * 256 + (size_t)nt * 64;
    const size_t slots = (size_t)p->num_cus * 12;
    if (hipMalloc(&p->tile_scratch, slots * per_slot * p->elem) != hipSuccess) {
      p->tile_scratch = nullptr;
      (void)hipGetLastError();
    }
  }

  *out = p;
  return MO_OK;
}

int mo_plan_destroy(mo_plan* plan) {
  if (!plan) return MO_OK;
  if (plan->G_scratch) (void)hipFree(plan->G_scratch);
  if (plan->c_scratch) (void)hipFree(plan->c_scratch);
  if (plan->tile_scratch) (void)hipFree(plan->tile_scratch);
  if (plan->H_work) (void)hipFree(plan->H_work);
  if (plan->ticket) (void)hipFree(plan->ticket);
  delete plan;
  return MO_OK;
}

const char* mo_plan_step_kernel(const mo_plan* plan, const mo_problem* prob) {
  if (!plan || !prob) return "invalid";
  mo::KernelArgs a;
  if (fill_problem(plan, prob, 1, true, true, &a) != MO_OK) return "invalid";
  a.mode = mo::MODE_STEP;
  a.vars = a.delta = reinterpret_cast<void*>(16);  // layout query only: assume 16-byte aligned, densely packed state / output
  a.vars_stride = a.delta_stride = plan->desc.n + 2 * plan->desc.m + plan->desc.k;
  a.ticket = plan->ticket; a.static_rounds = fused_static_rounds(plan); a.no_tiny = (plan->desc.flags & MO_PLAN_NO_TINY) != 0;
  switch (choose_kernel(plan, a)) {
    case KERNEL_FUSED_F64: return mo::fused_name(a, plan->desc.dtype);
    case KERNEL_FUSED_F32: return mo::fused_f32_name(a);
    default: return "generic";
  }
}

const char* mo_plan_solve_kernel(const mo_plan* plan, const mo_problem* prob) {
  if (!plan || !prob) return "invalid";
  mo::KernelArgs a;
  if (fill_problem(plan, prob, 1, true, true, &a) != MO_OK) return "invalid";
  a.mode = mo::MODE_SOLVE;
  a.vars = reinterpret_cast<void*>(16);  // layout query only
  a.vars_stride = plan->desc.n + 2 * plan->desc.m + plan->desc.k;
  a.ticket = plan->ticket; a.static_rounds = fused_static_rounds(plan); a.no_tiny = (plan->desc.flags & MO_PLAN_NO_TINY) != 0;
  switch (choose_kernel(plan, a)) {
    case KERNEL_FUSED_F64: return mo::fused_name(a, plan->desc.dtype);
    case KERNEL_FUSED_F32: return mo::fused_f32_name(a);
    default: return "generic";
  }
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

int mo_qp_eigenvalue_stats(mo_plan* plan, const mo_problem* prob, int64_t batch, void* out, void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  if (!out) return fail(MO_ERR_INVALID_ARGUMENT, "out is NULL");
  mo_plan tmp = *plan;
  tmp.desc.k = 0; tmp.desc.m = 0;   // only the cost (G, or J with its damping) takes part
  mo::KernelArgs a;
  if (int rc = fill_problem(&tmp, prob, batch, true, false, &a)) return rc;
  if (batch == 0) return MO_OK;
  if (mo::eig_lds_bytes(plan->desc.n, false) > 160 * 1024)
    return fail(MO_ERR_UNSUPPORTED, "n = %d: not even the vectors of the tridiagonal eigenvalue problem fit the 160 KiB of LDS", plan->desc.n);
  MO_HIP_CHECK(hipSetDevice(plan->desc.device));
  MO_HIP_CHECK(mo::launch_qp_eig(a, plan->desc.dtype, plan->num_cus, out, 3, plan->H_work, plan->H_work_slot_bytes, plan->H_work_slots,
                                 (hipStream_t)stream));
  return MO_OK;
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

}  // extern "C"

namespace {
int qp_solve_impl(mo_plan* plan, const mo_problem* prob, int64_t batch, const mo_solve_params* params, void* vars,
                  int64_t vars_stride, int32_t* termination, int32_t* num_iterations, void* iterations, void* lagrange,
                  int32_t* status, const int* skip, long long skip_stride, long long skip_active, void* stream) {
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
  a.skip = skip; a.skip_stride = skip_stride; a.skip_active = skip_active;
  a.ticket = plan->ticket; a.static_rounds = fused_static_rounds(plan); a.no_tiny = (plan->desc.flags & MO_PLAN_NO_TINY) != 0;
  const KernelChoice choice = choose_kernel(plan, a);
  const bool use_fused = choice != KERNEL_GENERIC;
  if (a.J && !use_fused) {  // the generic loop re-reads G after every factorisation: keep the linearised G, c in plan scratch
    if (batch > plan->desc.max_batch) return fail(MO_ERR_INVALID_ARGUMENT, "batch %lld > plan max_batch %lld", (long long)batch, (long long)plan->desc.max_batch);
    const size_t n = (size_t)plan->desc.n;
    MO_HIP_CHECK(hipSetDevice(plan->desc.device));
    if (!plan->G_scratch || !plan->c_scratch) {  // both or neither: a half-made pair would hand the kernel a NULL c_out
      void *g = nullptr, *c = nullptr;
      if (hipMalloc(&g, (size_t)plan->desc.max_batch * n * n * plan->elem) != hipSuccess ||
          hipMalloc(&c, (size_t)plan->desc.max_batch * n * plan->elem) != hipSuccess) {
        if (g) (void)hipFree(g);
        (void)hipGetLastError();
        return fail(MO_ERR_HIP, "hipMalloc of the linearisation scratch failed (max_batch %lld, n %zu)", (long long)plan->desc.max_batch, n);
      }
      if (plan->G_scratch) (void)hipFree(plan->G_scratch);
      if (plan->c_scratch) (void)hipFree(plan->c_scratch);
      plan->G_scratch = g; plan->c_scratch = c;
    }
    a.G_out = plan->G_scratch; a.G_out_stride = (long long)(n * n); a.G_out_ld = (int)n;
    a.c_out = plan->c_scratch; a.c_out_stride = (long long)n;
  }
  if (use_fused && plan->desc.dtype == MO_F64 && plan->desc.n > 32) {   // (fp32 and the 32 grid park every tile in LDS)
    // Tile park of the fused Solve kernels: the G tiles a wave cannot keep in LDS between the passes go to a scratch indexed by the
    // wave's slot in the persistent grid (one workgroup per CU, at most twelve waves each), so its size does not depend on the batch
    // and the lines a wave re-reads every pass stay in its XCD's L2.  Optional: without it the kernel rebuilds the tiles every pass.
    // Plan-owned: mo_plan_create allocates it (round 4; it used to be allocated by the first Solve -- an allocation in the launch path).
    const int nt = plan->desc.n > 96 ? 8 : plan->desc.n > 64 ? 6 : plan->desc.n > 32 ? 4 : 2;
    const size_t per_slot = (size_t)(nt * (nt + 1) / 2) * 256 + (size_t)nt * 64;
    a.G_out = plan->tile_scratch; a.G_out_stride = (long long)per_slot;
  }
  return launch_chosen(plan, a, choice, stream);  // fused Solve kernel (fp64: n <= 128; fp32: n = 64 / 128), generic kernel otherwise
}
}  // namespace

extern "C" {

int mo_qp_solve(mo_plan* plan, const mo_problem* prob, int64_t batch, const mo_solve_params* params, void* vars,
                int64_t vars_stride, int32_t* termination, int32_t* num_iterations, void* iterations, void* lagrange,
                int32_t* status, void* stream) {
  return qp_solve_impl(plan, prob, batch, params, vars, vars_stride, termination, num_iterations, iterations, lagrange, status, nullptr, 0, -1,
                       stream);
}

int mo_nullspace_solve(mo_plan* plan, const mo_problem* prob, int64_t batch, void* x_out, int64_t x_stride,
                       int32_t* termination, void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  if (plan->desc.k <= 0) return fail(MO_ERR_DIMENSION, "Problem must have at least one equality constraint");  // F_ASSERT_GT qp.cc:680
  if (plan->desc.k > plan->desc.n) return fail(MO_ERR_DIMENSION, "k = %d equality rows for n = %d variables", plan->desc.k, plan->desc.n);
  if (!x_out || !termination) return fail(MO_ERR_INVALID_ARGUMENT, "x_out / termination is NULL");
  mo_plan tmp = *plan;
  tmp.desc.m = 0;  // no inequalities on this path
  mo::KernelArgs a;
  if (int rc = fill_problem(&tmp, prob, batch, true, false, &a)) return rc;
  if (batch == 0) return MO_OK;
  a.delta = x_out; a.delta_stride = x_stride;
  a.status = termination;  // MO_STATUS_* first; translated to QPNullSpaceTerminationState below
  const size_t need = mo::nullspace_lds_bytes(plan->desc.n, plan->desc.k, a.m_r, plan->elem);
  if (need > 160 * 1024) return fail(MO_ERR_UNSUPPORTED, "the null-space solver keeps G and A_eq^T in LDS: %zu B needed (> 160 KiB)", need);
  MO_HIP_CHECK(hipSetDevice(plan->desc.device));
  MO_HIP_CHECK(mo::launch_nullspace(a, plan->desc.dtype, plan->num_cus, (hipStream_t)stream));
  mo::AuxArgs t;
  memset(&t, 0, sizeof(t));
  t.batch = batch; t.status = termination;
  MO_HIP_CHECK(mo::launch_nullspace_termination(t, (hipStream_t)stream));
  return MO_OK;
}

void mo_default_nls_params(mo_nls_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->max_iterations = 10;  // nonlinear.hpp:64-124
  p->max_qp_iterations = 10;
  p->termination_kkt_tolerance = 1.0e-6;
  p->absolute_exit_tol = 1.0e-12;
  p->relative_exit_tol = 1.0e-5;
  p->absolute_first_derivative_tol = 1.0e-6;
  p->max_line_search_iterations = 2;
  p->line_search_strategy = MO_POLYNOMIAL_APPROXIMATION;
  p->armijo_search_tau = 0.8;
  p->equality_penalty_initial = 1.0;
  p->equality_penalty_scale_factor = 1.01;
  p->equality_penalty_rho = 0.1;
  p->lambda_initial = 0.0;
  p->lambda_failure_init = 1.0e-2;
  p->lambda_decrease_on_success = 0.1;
  p->lambda_decrease_on_restore = 0.8;
  p->max_lambda = 1.0;
  p->min_lambda = 0.0;
  p->retraction = MO_RETRACT_EUCLIDEAN;
}


static bool nls_takes_nullspace_path(const mo_plan* plan) {
  const mo_plan_desc& d = plan->desc;
  return d.m == 0 && d.k > 0 && d.k <= d.n && mo::nullspace_lds_bytes(d.n, d.k, d.m_r, plan->elem) <= 160 * 1024;
}

int mo_plan_nls_uses_nullspace(const mo_plan* plan) { return plan && nls_takes_nullspace_path(plan) ? 1 : 0; }

int mo_nls_solve(mo_plan* plan, const mo_nls_problem* np, int64_t batch, const mo_nls_params* prm, mo_nls_eval_fn eval,
                 void* user, int32_t* termination, int32_t* num_iterations, void* iterations, int32_t* status, void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  const mo_plan_desc& d = plan->desc;
  if (d.dtype != MO_F64) return fail(MO_ERR_UNSUPPORTED, "mo_nls_solve needs an fp64 plan");
  if (!np || !prm || !eval) return fail(MO_ERR_INVALID_ARGUMENT, "problem / params / eval is NULL");
  if (batch < 0) return fail(MO_ERR_INVALID_ARGUMENT, "batch must be >= 0");
  if (d.m_r <= 0) return fail(MO_ERR_DIMENSION, "the plan must be created with m_r > 0 (the cost residual stack)");
  // CheckParams, nonlinear.cc:48-73
  if (prm->max_iterations < 0) return fail(MO_ERR_INVALID_ARGUMENT, "max_iterations must be >= 0");
  if (prm->max_qp_iterations < 1) return fail(MO_ERR_INVALID_ARGUMENT, "max_qp_iterations must be >= 1");
  if (!(prm->termination_kkt_tolerance > 0)) return fail(MO_ERR_INVALID_ARGUMENT, "termination_kkt_tolerance must be > 0");
  if (!(prm->absolute_exit_tol > 0)) return fail(MO_ERR_INVALID_ARGUMENT, "absolute_exit_tol must be > 0");
  if (prm->max_line_search_iterations < 0) return fail(MO_ERR_INVALID_ARGUMENT, "max_line_search_iterations must be >= 0");
  if (!(prm->relative_exit_tol >= 0) || !(prm->relative_exit_tol <= 1)) return fail(MO_ERR_INVALID_ARGUMENT, "relative_exit_tol must be in [0, 1]");
  if (!(prm->absolute_first_derivative_tol >= 0)) return fail(MO_ERR_INVALID_ARGUMENT, "absolute_first_derivative_tol must be >= 0");
  if (!(prm->armijo_search_tau > 0) || !(prm->armijo_search_tau < 1)) return fail(MO_ERR_INVALID_ARGUMENT, "armijo_search_tau must be in (0, 1)");
  if (!(prm->equality_penalty_initial >= 0)) return fail(MO_ERR_INVALID_ARGUMENT, "equality_penalty_initial must be >= 0");
  if (!(prm->equality_penalty_scale_factor >= 1.0)) return fail(MO_ERR_INVALID_ARGUMENT, "equality_penalty_scale_factor must be >= 1");
  if (!(prm->equality_penalty_rho >= 0) || !(prm->equality_penalty_rho < 1)) return fail(MO_ERR_INVALID_ARGUMENT, "equality_penalty_rho must be in [0, 1)");
  if (!(prm->max_lambda >= 0) || !(prm->min_lambda <= prm->max_lambda)) return fail(MO_ERR_INVALID_ARGUMENT, "need 0 <= min_lambda <= max_lambda");
  if (!(prm->lambda_initial >= prm->min_lambda) || !(prm->lambda_initial <= prm->max_lambda)) return fail(MO_ERR_INVALID_ARGUMENT, "lambda_initial outside [min_lambda, max_lambda]");
  if (!(prm->lambda_failure_init >= 0)) return fail(MO_ERR_INVALID_ARGUMENT, "lambda_failure_init must be >= 0");
  if (!(prm->lambda_decrease_on_success >= 0) || !(prm->lambda_decrease_on_success < 1.0)) return fail(MO_ERR_INVALID_ARGUMENT, "lambda_decrease_on_success must be in [0, 1)");
  if (!(prm->lambda_decrease_on_restore >= 0) || !(prm->lambda_decrease_on_restore < 1.0)) return fail(MO_ERR_INVALID_ARGUMENT, "lambda_decrease_on_restore must be in [0, 1)");
  if (prm->line_search_strategy != MO_ARMIJO_BACKTRACK && prm->line_search_strategy != MO_POLYNOMIAL_APPROXIMATION)
    return fail(MO_ERR_INVALID_ARGUMENT, "bad line_search_strategy");
  if (prm->retraction < MO_RETRACT_EUCLIDEAN || prm->retraction > MO_RETRACT_CALLBACK) return fail(MO_ERR_INVALID_ARGUMENT, "bad retraction");
  if (prm->retraction == MO_RETRACT_CALLBACK && (!np->step || !np->step_alpha))
    return fail(MO_ERR_INVALID_ARGUMENT, "MO_RETRACT_CALLBACK needs the step / step_alpha buffers");
  if (!np->vars || !np->candidate || !np->J || !np->r || !np->r_cand) return fail(MO_ERR_INVALID_ARGUMENT, "vars / candidate / J / r / r_cand is NULL");
  if (d.k > 0 && (!np->J_eq || !np->r_eq || !np->r_eq_cand)) return fail(MO_ERR_DIMENSION, "k = %d but J_eq / r_eq / r_eq_cand is NULL", d.k);
  if (d.m > 0 && (!np->cons_var || !np->cons_a || !np->cons_b)) return fail(MO_ERR_INVALID_ARGUMENT, "m = %d but constraint arrays are NULL", d.m);
  if (prm->log_qp_eigenvalues && !np->qp_eigenvalues) return fail(MO_ERR_INVALID_ARGUMENT, "log_qp_eigenvalues needs the qp_eigenvalues buffer");
  if (!termination) return fail(MO_ERR_INVALID_ARGUMENT, "termination is NULL");
  if (batch == 0) return MO_OK;
  MO_HIP_CHECK(hipSetDevice(d.device));
  hipStream_t s = (hipStream_t)stream;
  const int n = d.n, k = d.k, m = d.m;
  const long long V = n + 2 * m + k, Vs = (V + 1) & ~1ll;  // even stride: the fused kernels want 16-byte aligned states

  NlsScratch scratch;
  double *qp_vars, *cons_b, *cons_a, *errors_pre, *errors_step, *deriv, *quad, *lagrange, *sd;
  int *qp_status, *qp_term, *qp_nit, *si, *counters, *cons_var;
  scratch.reserve<double>((size_t)batch * Vs);
  for (int i = 0; i < 2; ++i) scratch.reserve<double>((size_t)batch * m);
  scratch.reserve<int>((size_t)batch * m);
  for (int i = 0; i < 3; ++i) scratch.reserve<double>((size_t)batch * 2);
  scratch.reserve<double>((size_t)batch);
  scratch.reserve<double>((size_t)batch * 2);
  scratch.reserve<double>((size_t)batch * mo::NLS_SD);
  for (int i = 0; i < 3; ++i) scratch.reserve<int>((size_t)batch);
  scratch.reserve<int>((size_t)batch * mo::NLS_SI);
  scratch.reserve<int>(2);
  MO_HIP_CHECK(scratch.commit());
  MO_HIP_CHECK(scratch.alloc(&qp_vars, (size_t)batch * Vs));
  MO_HIP_CHECK(scratch.alloc(&cons_b, (size_t)batch * m));  // the QP's constraints: one stride for (variable, a, shifted b)
  MO_HIP_CHECK(scratch.alloc(&cons_a, (size_t)batch * m));
  MO_HIP_CHECK(scratch.alloc(&cons_var, (size_t)batch * m));
  MO_HIP_CHECK(scratch.alloc(&errors_pre, (size_t)batch * 2));
  MO_HIP_CHECK(scratch.alloc(&errors_step, (size_t)batch * 2));
  MO_HIP_CHECK(scratch.alloc(&deriv, (size_t)batch * 2));
  MO_HIP_CHECK(scratch.alloc(&quad, (size_t)batch));
  MO_HIP_CHECK(scratch.alloc(&lagrange, (size_t)batch * 2));
  MO_HIP_CHECK(scratch.alloc(&sd, (size_t)batch * mo::NLS_SD));
  MO_HIP_CHECK(scratch.alloc(&qp_status, (size_t)batch));
  MO_HIP_CHECK(scratch.alloc(&qp_term, (size_t)batch));
  MO_HIP_CHECK(scratch.alloc(&qp_nit, (size_t)batch));
  MO_HIP_CHECK(scratch.alloc(&si, (size_t)batch * mo::NLS_SI));
  MO_HIP_CHECK(scratch.alloc(&counters, 2));

  mo::NlsArgs na;
  memset(&na, 0, sizeof(na));
  na.n = n; na.k = k; na.m = m; na.batch = batch; na.prm = *prm;
  na.vars = (double*)np->vars; na.vars_stride = np->vars_stride;
  na.cand = (double*)np->candidate; na.cand_stride = np->candidate_stride;
  na.qp_vars = qp_vars; na.qp_vars_stride = Vs;
  na.errors_pre = errors_pre; na.errors_step = errors_step; na.deriv = deriv; na.quad = quad; na.lagrange = lagrange;
  na.qp_status = qp_status; na.qp_term = qp_term; na.qp_nit = qp_nit;
  na.sd = sd; na.si = si;
  na.iterations = (double*)iterations; na.rec = MO_NLS_ITER_RECORD(prm->max_line_search_iterations);
  na.termination = termination; na.num_iterations = num_iterations; na.status = status;
  na.counters = counters;
  na.step = (double*)np->step; na.step_stride = np->step_stride; na.step_alpha = (double*)np->step_alpha;
  na.user_exit = np->user_exit;
  const bool retract_cb = prm->retraction == MO_RETRACT_CALLBACK;
  if (retract_cb) MO_HIP_CHECK(hipMemsetAsync(np->step_alpha, 0, sizeof(double) * (size_t)batch, s));
  if (np->user_exit) MO_HIP_CHECK(hipMemsetAsync(np->user_exit, 0, sizeof(int32_t) * (size_t)batch, s));
  MO_HIP_CHECK(mo::launch_nls_init(na, s));
  if (prm->max_iterations == 0) {  // nonlinear.cc:97, 157: no iteration at all
    MO_HIP_CHECK(hipMemsetAsync(termination, 0, sizeof(int32_t) * (size_t)batch, s));
    if (num_iterations) MO_HIP_CHECK(hipMemsetAsync(num_iterations, 0, sizeof(int32_t) * (size_t)batch, s));
    return MO_OK;
  }

  // the QP of this outer iteration (LinearizeAndFillQP's output, nonlinear.cc:98) in J-level form
  mo_problem qp;
  memset(&qp, 0, sizeof(qp));
  qp.J = np->J; qp.J_stride = np->J_stride; qp.J_ld = np->J_ld; qp.J_layout = np->J_layout;
  qp.r = np->r; qp.r_stride = np->r_stride;
  qp.lambda_vec = sd + mo::NLS_SD_LAMBDA; qp.lambda_stride = mo::NLS_SD;
  qp.A_eq = np->J_eq; qp.A_stride = np->J_eq_stride; qp.A_ld = np->J_eq_ld;
  qp.b_eq = np->r_eq; qp.b_stride = np->r_eq_stride;
  qp.cons_var = cons_var; qp.cons_a = cons_a; qp.cons_b = cons_b; qp.cons_stride = m;
  mo_solve_params sp;
  mo_default_solve_params(&sp);       // ComputeStepDirection, nonlinear.cc:226-241
  sp.max_iterations = prm->max_qp_iterations;
  sp.termination_kkt_tol = prm->termination_kkt_tolerance;
  sp.initial_mu = 1.0;
  sp.sigma = 0.1;
  sp.initialize_mu_with_complementarity = 0;
  sp.initial_guess_method = k > 0 ? MO_GUESS_SOLVE_EQUALITY_CONSTRAINED : MO_GUESS_NAIVE;

  mo::AuxArgs ea;  // errors at the linearisation point / at the candidate
  memset(&ea, 0, sizeof(ea));
  ea.n = n; ea.k = k; ea.m = m; ea.m_r = d.m_r; ea.batch = batch;
  mo::AuxArgs sa = ea;  // ShiftTo
  sa.x = np->vars; sa.x_stride = np->vars_stride;
  sa.cons_var = np->cons_var; sa.cons_a = np->cons_a; sa.cons_b = np->cons_b; sa.cons_stride = np->cons_stride;
  sa.cons_b_out = cons_b; sa.cons_b_out_stride = m; sa.cons_var_out = cons_var; sa.cons_a_out = cons_a;
  mo::AuxArgs da = ea;  // ComputeQPCostDerivative
  da.x = qp_vars; da.x_stride = Vs;
  da.J = np->J; da.J_stride = np->J_stride; da.J_ld = np->J_ld; da.J_row_major = np->J_layout == MO_ROW_MAJOR;
  da.r = np->r; da.r_stride = np->r_stride;
  da.A = np->J_eq; da.A_stride = np->J_eq_stride; da.A_ld = np->J_eq_ld; da.b = np->r_eq; da.b_stride = np->r_eq_stride;
  da.lambda_vec = sd + mo::NLS_SD_LAMBDA; da.lambda_vec_stride = mo::NLS_SD;
  da.out2 = deriv; da.quad_out = quad;

  int host_counters[2];
  long long still_active = batch;  // problems the previous outer iteration left active (a hint for the QP kernel's ticket size)
  for (int iter = 0; iter < prm->max_iterations; ++iter) {
    na.iter = iter;
    if (eval(user, MO_NLS_EVAL_LINEARIZE, stream) != 0) return fail(MO_ERR_CALLBACK, "eval(LINEARIZE) failed at iteration %d", iter);
    // errors_pre (nonlinear.cc:184-186, 203) and the shifted constraints (:209-212)
    ea.r = np->r; ea.r_stride = np->r_stride; ea.b = np->r_eq; ea.b_stride = np->r_eq_stride; ea.out2 = errors_pre;
    MO_HIP_CHECK(mo::launch_nonlinear_errors(ea, d.dtype, s));
    if (m > 0) MO_HIP_CHECK(mo::launch_shift_constraints(sa, d.dtype, s));
    // ComputeStepDirection (nonlinear.cc:216-247): the interior-point QP on device
    if (nls_takes_nullspace_path(plan)) {
      // equality constraints only: QPNullSpaceSolver (nonlinear.cc:83-86, 249-258) -- singular G = J^T J is fine as long as
      // the reduced Hessian is positive definite; NOT_POSITIVE_DEFINITE ends the problem with QP_INDEFINITE (:103-105)
      mo_plan tmp = *plan;
      mo::KernelArgs ka;
      if (int rc = fill_problem(&tmp, &qp, batch, true, false, &ka)) return rc;
      ka.delta = qp_vars; ka.delta_stride = Vs; ka.status = qp_status;
      MO_HIP_CHECK(hipMemsetAsync(qp_term, 0, sizeof(int) * (size_t)batch, s));
      MO_HIP_CHECK(hipMemsetAsync(qp_nit, 0, sizeof(int) * (size_t)batch, s));
      MO_HIP_CHECK(mo::launch_nullspace(ka, d.dtype, plan->num_cus, s));
    } else {
      void* qp_records = np->qp_iterations ? (void*)((double*)np->qp_iterations + (size_t)iter * (size_t)batch * sp.max_iterations * MO_ITER_RECORD) : nullptr;
      if (int rc = qp_solve_impl(plan, &qp, batch, &sp, qp_vars, Vs, qp_term, qp_nit, qp_records, lagrange, qp_status,
                                 si + mo::NLS_SI_TERM, mo::NLS_SI, still_active, stream)) return rc;  // terminated problems are skipped
      if (np->qp_lagrange && k > 0)
        MO_HIP_CHECK(hipMemcpyAsync((double*)np->qp_lagrange + (size_t)iter * (size_t)batch * 2, lagrange, sizeof(double) * 2 * (size_t)batch,
                                    hipMemcpyDeviceToDevice, s));
    }
    if (prm->log_qp_eigenvalues) {   // qp_.ComputeEigenvalueStats() of this iteration's QP (nonlinear.cc:138), still-active problems only
      mo_plan tmp = *plan;
      tmp.desc.k = 0; tmp.desc.m = 0;
      mo::KernelArgs ka;
      if (int rc = fill_problem(&tmp, &qp, batch, true, false, &ka)) return rc;
      ka.skip = si + mo::NLS_SI_TERM; ka.skip_stride = mo::NLS_SI;
      MO_HIP_CHECK(mo::launch_qp_eig(ka, d.dtype, plan->num_cus, (double*)np->qp_eigenvalues + (size_t)iter * (size_t)batch * 3, 3, plan->H_work,
                                     plan->H_work_slot_bytes, plan->H_work_slots, s));
    }
    MO_HIP_CHECK(mo::launch_cost_derivative(da, d.dtype, s));
    MO_HIP_CHECK(hipMemsetAsync(counters, 0, 2 * sizeof(int), s));
    MO_HIP_CHECK(mo::launch_nls_begin_search(na, s));
    if (retract_cb && eval(user, MO_NLS_EVAL_RETRACT, stream) != 0) return fail(MO_ERR_CALLBACK, "eval(RETRACT) failed at iteration %d", iter);
    // SelectStepSize (nonlinear.cc:346-412)
    for (int ls = 0; ls <= prm->max_line_search_iterations; ++ls) {
      MO_HIP_CHECK(hipMemcpyAsync(host_counters, counters, sizeof(int), hipMemcpyDeviceToHost, s));
      MO_HIP_CHECK(hipStreamSynchronize(s));
      if (host_counters[0] == 0) break;  // nobody is searching any more
      na.ls = ls;
      if (eval(user, MO_NLS_EVAL_ERRORS, stream) != 0) return fail(MO_ERR_CALLBACK, "eval(ERRORS) failed at iteration %d", iter);
      ea.r = np->r_cand; ea.r_stride = np->r_cand_stride; ea.b = np->r_eq_cand; ea.b_stride = np->r_eq_cand_stride; ea.out2 = errors_step;
      MO_HIP_CHECK(mo::launch_nonlinear_errors(ea, d.dtype, s));
      MO_HIP_CHECK(hipMemsetAsync(counters, 0, sizeof(int), s));
      MO_HIP_CHECK(mo::launch_nls_search_step(na, s));
      if (retract_cb && eval(user, MO_NLS_EVAL_RETRACT, stream) != 0) return fail(MO_ERR_CALLBACK, "eval(RETRACT) failed at iteration %d", iter);
    }
    MO_HIP_CHECK(mo::launch_nls_update(na, s));
    if (np->user_exit) {  // SetUserExitCallback, nonlinear.cc:142-149
      if (eval(user, MO_NLS_EVAL_ITERATION_DONE, stream) != 0) return fail(MO_ERR_CALLBACK, "eval(ITERATION_DONE) failed at iteration %d", iter);
      MO_HIP_CHECK(mo::launch_nls_user_exit(na, s));
    }
    MO_HIP_CHECK(hipMemcpyAsync(host_counters, counters, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    MO_HIP_CHECK(hipStreamSynchronize(s));
    if (host_counters[1] == 0) break;  // every problem has terminated
    still_active = host_counters[1];
  }
  MO_HIP_CHECK(hipStreamSynchronize(s));  // the scratch is released on return
  return MO_OK;
}

int mo_residual_eval(mo_plan* plan, int32_t family, int32_t rows, const void* params, const void* x, int64_t x_stride,
                     int64_t batch, void* r, int64_t r_stride, void* J, int64_t J_stride, int32_t J_ld, int32_t J_layout,
                     void* stream) {
  g_err[0] = 0;
  if (int rc = check_plan(plan)) return rc;
  const mo_plan_desc& d = plan->desc;
  if (batch < 0) return fail(MO_ERR_INVALID_ARGUMENT, "batch must be >= 0");
  if (!x || !r) return fail(MO_ERR_INVALID_ARGUMENT, "x / r is NULL");
  const int want = mo::residual_family_rows(family, d.n, rows);
  if (want < 0 || want != rows) return fail(MO_ERR_DIMENSION, "residual family %d with n = %d has %d rows, not %d", family, d.n, want, rows);
  if (family == MO_RESIDUAL_PRODUCT_PAIRS && !params) return fail(MO_ERR_INVALID_ARGUMENT, "PRODUCT_PAIRS needs params");
  if (family == MO_RESIDUAL_ACTUATOR_CHAIN && !params) return fail(MO_ERR_INVALID_ARGUMENT, "ACTUATOR_CHAIN needs its parameter block");
  if (J) {
    if (J_layout != MO_ROW_MAJOR && J_layout != MO_COL_MAJOR) return fail(MO_ERR_INVALID_ARGUMENT, "bad J_layout");
    const int min_ld = J_layout == MO_ROW_MAJOR ? d.n : rows;
    if (J_ld < min_ld) return fail(MO_ERR_DIMENSION, "J_ld %d < %d", J_ld, min_ld);
  }
  MO_HIP_CHECK(hipSetDevice(d.device));
  MO_HIP_CHECK(mo::launch_residual_family(family, d.n, rows, batch, d.dtype, params, x, x_stride, r, r_stride, J, J_stride, J_ld,
                                          J_layout == MO_ROW_MAJOR, (hipStream_t)stream));
  return MO_OK;
}

}  // extern "C"
