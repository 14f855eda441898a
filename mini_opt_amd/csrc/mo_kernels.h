// mo_kernels.h -- internal launch interface between the C ABI (mo_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mini_opt_hip.h"

namespace mo {

enum Mode : int {
  MODE_LINEARIZE = 0,  // nonlinear.cc:182-189 (J^T J, J^T r)
  MODE_RESIDUAL = 1,   // qp.cc:391-437
  MODE_STEP = 2,       // qp.cc:391-420, 275-364, 485-507 on the caller's state
  MODE_ITERATE = 3,    // qp.cc:153-201
  MODE_SOLVE = 4,      // qp.cc:100-151
};

// Everything a kernel needs, passed by value (kernarg segment).
struct KernelArgs {
  int n, k, m, m_r;
  int mode;
  unsigned flags;  // MO_STEP_*
  long long batch;
  // problem
  const void* J; long long J_stride; int J_ld; int J_row_major;
  const void* r; long long r_stride;
  double lambda;
  const void* lambda_vec; long long lambda_vec_stride;  // per-problem damping, overrides lambda when non-NULL
  const void* G; long long G_stride; int G_ld;
  const void* c; long long c_stride;
  const void* A; long long A_stride; int A_ld;
  const void* b; long long b_stride;
  const int* cons_var; const void* cons_a; const void* cons_b; long long cons_stride;
  // state
  void* vars; long long vars_stride;
  const void* mu; long long mu_stride;
  double tau;
  int barrier_strategy;
  // outputs
  void* delta; long long delta_stride;
  void* alpha;     // [batch][2]
  int* status;     // [batch]
  void* ip_out;    // [batch][6]
  void* r_out; long long r_out_stride;  // MODE_RESIDUAL
  void* kkt_out;   // [batch][4]
  void* G_out; long long G_out_stride; int G_out_ld;  // MODE_LINEARIZE (also SOLVE scratch for J-level input)
  void* c_out; long long c_out_stride;
  void* half_sq_out; long long half_sq_stride;  // stride in elements (0 means 1)
  // MODE_SOLVE
  mo_solve_params sp;
  int* termination; int* num_iterations; void* iterations; void* lagrange;
  // MODE_SOLVE inside mo_nls_solve: problems whose word skip[p * skip_stride] is >= 0 have terminated and are left untouched
  const int* skip; long long skip_stride;
  long long skip_active;  // how many of them are still active (host-side count from the previous outer iteration; -1: unknown)
  // fused kernel: device work counter (plan-owned, zeroed on the launch stream before every launch)
  unsigned long long* ticket;
  int stagger;     // fused step kernel: start offset between the waves that share a SIMD, in units of 127 x 64 cycles (set by launch_fused)
  int chain_prio;  // fused step kernel: s_setprio 1 while a wave is in the elimination / substitution chains
  int static_rounds;  // fused kernels: launches of at most this many problems per wave are split statically, round by round, with no ticket
                      // (mo_api.hip hands over -1 = "the launcher decides" or MO_FUSED_STATIC_ROUNDS; 0 = tickets always).  The grids are
                      // min(CUs, ceil(batch / 4)) workgroups, so that a small batch spreads one wave per SIMD over the CUs before any SIMD
                      // gets a second wave.  Measured: the 32-variable grid wins with static rounds at every batch up to
                      // 65 536 (21 rounds), the 64 grid and the fp32 128 grid up to ~8 rounds; beyond, tickets in guided chunks balance better.
  int no_tiny;     // MO_PLAN_NO_TINY: keep n + k <= 15 on the 32-variable tile grid (set by mo_api.hip from the plan flags)
  // generic kernel beyond its LDS-resident range: P x ldh workspace of H per workgroup of the persistent grid (plan-owned)
  void* H_work; long long H_work_stride;
  long long H_work_slots;   // workgroup slots behind H_work: the launch clamps its grid to it
  // diagnostics only (tools/phase_timer.hip builds kkt_fused.hip with MO_FUSED_STAMPS); NULL in the product
  unsigned long long* debug;
};

// shape-generic LDS kernel (any n,k,m,m_r that fits LDS), kkt_generic.hip
size_t generic_lds_bytes(const KernelArgs& a, int elem_size);
hipError_t launch_generic(const KernelArgs& a, int dtype, int num_cus, hipStream_t stream);
// beyond n + k = 192 or the 160 KiB of LDS the generic kernel keeps H in a global workspace (blocked LDL^T, kkt_generic.hip "LARGE")
bool generic_needs_large(const KernelArgs& a, int elem_size);
size_t generic_large_lds_bytes(const KernelArgs& a, int elem_size);   // > 160 KiB: not even the vectors fit
size_t generic_large_workspace_elems(const KernelArgs& a);            // per workgroup
int generic_large_grid(const KernelArgs& a, int elem_size, int num_cus);
// QPNullSpaceSolver::Solve (qp.cc:679-729): pivoted Householder QR of A_eq^T, reduced Hessian, LLT; x -> a.delta, status -> a.status
hipError_t launch_nullspace(const KernelArgs& a, int dtype, int num_cus, hipStream_t stream);
size_t nullspace_lds_bytes(int n, int k, int m_r, int elem_size);

// QP::ComputeEigenvalueStats (qp.cc:12-16): {min, max, min |.|} of the eigenvalues of sym(G) per problem, eig_kernels.hip.  `work`: the plan's
// global workspace (one slot of work_slot_bytes per workgroup) for matrices beyond the LDS; out [batch][3] in the plan's dtype.
bool eig_needs_global(int n);
size_t eig_workspace_bytes(int n);
size_t eig_lds_bytes(int n, bool matrix_in_lds);
int eig_grid(int n, int num_cus);
hipError_t launch_qp_eig(const KernelArgs& a, int dtype, int num_cus, void* out, long long out_stride, void* work, size_t work_slot_bytes,
                         long long work_slots, hipStream_t stream);

// fused single-wave MFMA kernels for fixed shapes, kkt_fused.hip.  Returns false if (shape, layout) is unsupported.
bool fused_supported(const KernelArgs& a, int dtype);
const char* fused_name(const KernelArgs& a, int dtype);
hipError_t launch_fused(const KernelArgs& a, int dtype, int num_cus, hipStream_t stream);
bool fused_needs_gather(const KernelArgs& a);  // J-level input in a layout only the per-lane gather stream takes
hipError_t launch_fused_gather(const KernelArgs& a, int num_cus, hipStream_t stream);  // kkt_fused_gather.hip
hipError_t launch_fused_ny2(const KernelArgs& a, int num_cus, hipStream_t stream);     // kkt_fused_ny2.hip: 16 <= k <= 31
hipError_t launch_fused_ny34(const KernelArgs& a, int num_cus, hipStream_t stream);    // kkt_fused_ny34.hip: 32 <= k <= 63 on the 32 / 64 grids
hipError_t launch_fused_mc4(const KernelArgs& a, int num_cus, hipStream_t stream);     // kkt_fused_mc4.hip: 128 < m <= 256; Solve with m > 64 on the 96 / 128 grids
bool fused_tiny_supported(const KernelArgs& a);                                          // kkt_fused_tiny.hip: n + k <= 15, m <= 64 (a subset of fused_supported)
hipError_t launch_fused_tiny(const KernelArgs& a, int num_cus, hipStream_t stream);


// fused single-wave fp32 step kernel for n = 64 / 128 (J-level input), kkt_fused_f32.hip
bool fused_f32_supported(const KernelArgs& a, int dtype);
const char* fused_f32_name(const KernelArgs& a);
hipError_t launch_fused_f32(const KernelArgs& a, int num_cus, hipStream_t stream);

// small per-problem kernels around the QP (LinearizeAndFillQP tail, EvaluateNonlinearErrors, ComputeQPCostDerivative), nls_kernels.hip
struct AuxArgs {
  int n, k, m, m_r;
  long long batch;
  const void* x; long long x_stride;                  // linearisation point / direction dx
  const void* J; long long J_stride; int J_ld; int J_row_major;
  const void* r; long long r_stride;
  const void* G; long long G_stride; int G_ld;
  const void* c; long long c_stride;
  const void* A; long long A_stride; int A_ld;
  const void* b; long long b_stride;                  // b_eq or r_eq
  const int* cons_var; const void* cons_a; const void* cons_b; long long cons_stride;
  double lambda; const void* lambda_vec; long long lambda_vec_stride;
  void* cons_b_out; long long cons_b_out_stride;
  int* cons_var_out; void* cons_a_out;                // optional per-problem copies of (variable, a), same stride as cons_b_out
  void* out2;                                         // [batch][2]
  void* quad_out;                                     // [batch]
  int* status;
};
hipError_t launch_shift_constraints(const AuxArgs& a, int dtype, hipStream_t stream);
hipError_t launch_nonlinear_errors(const AuxArgs& a, int dtype, hipStream_t stream);
hipError_t launch_cost_derivative(const AuxArgs& a, int dtype, hipStream_t stream);
hipError_t launch_nullspace_termination(const AuxArgs& a, hipStream_t stream);  // status[p] = status[p] != 0

// per-problem state machine of the batched SQP loop (mo_nls_solve), nls_kernels.hip; fp64 only
enum { NLS_SD_LAMBDA = 0, NLS_SD_PENALTY, NLS_SD_ALPHA, NLS_SD_DIRECTIONAL, NLS_SD_A2, NLS_SD_T2, NLS_SD_A1, NLS_SD_T1, NLS_SD = 8 };
enum { NLS_SI_TERM = 0, NLS_SI_STATE, NLS_SI_LS_RESULT, NLS_SI_NSTEPS, NLS_SI_NITER, NLS_SI = 8 };
struct NlsArgs {
  int n, k, m;
  long long batch;
  mo_nls_params prm;
  int iter, ls;                                    // outer / line-search iteration this launch belongs to
  double* vars; long long vars_stride;
  double* cand; long long cand_stride;
  const double* qp_vars; long long qp_vars_stride;  // [x | s | y | z] of the QP; dx = its x block
  const double* errors_pre;                         // [batch][2]
  const double* errors_step;                        // [batch][2]
  const double* deriv;                              // [batch][2]
  const double* quad;                               // [batch]
  const double* lagrange;                           // [batch][2]
  const int* qp_status; const int* qp_term; const int* qp_nit;
  double* sd; int* si;                              // [batch][NLS_SD], [batch][NLS_SI]
  double* iterations; int rec;                      // [batch][max_iterations][rec] or NULL
  int* termination; int* num_iterations; int* status;
  int* counters;                                    // [0] problems still in the line search, [1] problems still active
  double* step; long long step_stride; double* step_alpha;  // MO_RETRACT_CALLBACK: dx and alpha handed to the caller's retraction
  const int* user_exit;                             // SetUserExitCallback flags (NULL: none)
};
// device residual families (nls_kernels.hip); rows = -1 if (family, n, rows_hint) is not a valid combination
int residual_family_rows(int family, int n, int rows_hint);
hipError_t launch_residual_family(int family, int n, int rows, long long batch, int dtype, const void* prm, const void* x,
                                  long long x_stride, void* r, long long r_stride, void* J, long long J_stride, int J_ld,
                                  int row_major, hipStream_t stream);
hipError_t launch_nls_init(const NlsArgs& a, hipStream_t stream);
hipError_t launch_nls_begin_search(const NlsArgs& a, hipStream_t stream);
hipError_t launch_nls_search_step(const NlsArgs& a, hipStream_t stream);
hipError_t launch_nls_update(const NlsArgs& a, hipStream_t stream);
hipError_t launch_nls_user_exit(const NlsArgs& a, hipStream_t stream);

}  // namespace mo
