// eig_kernels.hip -- QP::ComputeEigenvalueStats (qp.cc:12-16): {min, max, min |.|} of the eigenvalues of the QP Hessian G, per problem of a
// batch.  The reference runs Eigen's SelfAdjointEigenSolver on G (lower triangle) and only LOGS the three numbers (Params::log_qp_eigenvalues,
// nonlinear.cc:138; NLSIteration::qp_eigenvalues, structs.hpp:267-310), so this is not a hot path: correctness and any n first.
//
// One 256-thread workgroup per problem, always in fp64 (an fp32 plan's G is widened on load):
//   1. A = sym(G) from the lower triangle of the caller's column-major G -- or G = J^T J + lambda I from the stacked Jacobian (any layout of J),
//      as LinearizeAndFillQP forms it (nonlinear.cc:182-189) -- in LDS (n x (n | 1) doubles); beyond ~139 variables in the plan's global workspace
//      slot of the workgroup (the H workspace of the generic kernel's LARGE path, allocated by mo_plan_create).
//   2. Householder tridiagonalisation T = Q^T A Q (n - 2 reflections; the matrix-vector product and the rank-2 update are spread over the workgroup).
//   3. Sturm counts on T: the number of eigenvalues below x is the number of negative terms of q_0 = d_0 - x, q_i = d_i - x - e_{i-1}^2 / q_{i-1}.
//      Four eigenvalues are wanted -- index 0 (min), n - 1 (max) and the two around zero, indices c - 1 and c with c = count(0) -- and each is
//      located by 64-way multisection: 64 lanes evaluate 64 counts per round, ten rounds shrink the Gershgorin interval by 65^10 > 2^60.
//      All 256 threads work: four targets x 64 section points.  Accuracy: a few ulp of |T| absolute, like LAPACK's bisection.
#include <math.h>

#include "mo_kernels.h"

namespace mo {
namespace {

constexpr int kEigThreads = 256;

struct EigArgs {
  int n, m_r;
  long long batch;
  const void* G; long long G_stride; int G_ld;
  const void* J; long long J_stride; int J_ld; int J_row_major;
  double lambda; const void* lambda_vec; long long lambda_vec_stride;
  const int* skip; long long skip_stride;   // mo_nls_solve: problems whose word is >= 0 have terminated
  double* work; long long work_stride;      // global A slots (doubles), NULL: A lives in LDS
  void* out; long long out_stride;          // [batch][3] in the plan's dtype
};

__device__ inline double block_sum(double v, double* red, int tid) {   // sum over the workgroup, result in every thread
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ __launch_bounds__(kEigThreads) void qp_eig_kernel(const EigArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = a.n, tid = threadIdx.x;
  const int ld = n | 1;
  double* const d = reinterpret_cast<double*>(smem);   // diagonal of T
  double* const e = d + n;                             // sub-diagonal of T (n - 1 entries)
  double* const v = e + n;                             // Householder vector
  double* const pw = v + n;                            // p = beta A v, then w
  double* const red = pw + n;                          // 8 doubles of reduction scratch
  double* const sect = red + 8;                        // 4 targets x {lo, hi}
  int* const cnt = reinterpret_cast<int*>(sect + 8);   // 256 Sturm counts
  double* const A = a.work ? a.work + (size_t)blockIdx.x * (size_t)a.work_stride : reinterpret_cast<double*>(cnt + kEigThreads);

  for (long long p = blockIdx.x; p < a.batch; p += gridDim.x) {
    if (a.skip && a.skip[p * a.skip_stride] >= 0) continue;   // wave-uniform per workgroup: one problem per workgroup
    __syncthreads();
    // ---- 1. A = sym(G)
    if (a.J) {
      const T* Jp = (const T*)a.J + p * a.J_stride;
      const long long rs = a.J_row_major ? a.J_ld : 1, cs = a.J_row_major ? 1 : a.J_ld;
      const double lam_in = a.lambda_vec ? (double)((const T*)a.lambda_vec)[p * a.lambda_vec_stride] : a.lambda;
      const double lam = lam_in > 0.0 ? lam_in : 0.0;   // nonlinear.cc:187-189
      for (int idx = tid; idx < n * n; idx += kEigThreads) {
        const int i = idx % n, j = idx / n;
        if (i < j) continue;
        double s = 0.0;
        for (int r = 0; r < a.m_r; ++r) s = fma((double)Jp[r * rs + i * cs], (double)Jp[r * rs + j * cs], s);
        if (i == j) s += lam;
        A[i + (size_t)j * ld] = s;
        A[j + (size_t)i * ld] = s;
      }
    } else {
      const T* Gp = (const T*)a.G + p * a.G_stride;
      for (int idx = tid; idx < n * n; idx += kEigThreads) {
        const int i = idx % n, j = idx / n;
        if (i < j) continue;
        const double s = (double)Gp[i + (size_t)j * a.G_ld];   // lower triangle only (SelfAdjointEigenSolver reads Lower)
        A[i + (size_t)j * ld] = s;
        A[j + (size_t)i * ld] = s;
      }
    }
    __syncthreads();
    // ---- 2. Householder tridiagonalisation (column k: annihilate A[k + 2 .., k])
    for (int k = 0; k + 2 < n; ++k) {
      const int m = n - k - 1;                       // length of x = A[k + 1 .., k]
      double part = 0.0;
      for (int i = tid; i < m; i += kEigThreads) { const double x = A[(k + 1 + i) + (size_t)k * ld]; v[i] = x; if (i > 0) part = fma(x, x, part); }
      const double tail2 = block_sum(part, red, tid);   // |x[1 ..]|^2 (the barriers inside publish v)
      const double x0 = v[0];
      if (tail2 == 0.0) {                               // already tridiagonal in this column: no reflection
        if (tid == 0) e[k] = x0;
        __syncthreads();
        continue;
      }
      const double norm = sqrt(fma(x0, x0, tail2));
      const double alpha = x0 > 0.0 ? -norm : norm;
      const double v0 = x0 - alpha;
      const double beta = 2.0 / fma(v0, v0, tail2);     // H = I - beta v v^T maps x to alpha e_1
      __syncthreads();
      if (tid == 0) { v[0] = v0; e[k] = alpha; }
      __syncthreads();
      // p = beta A22 v
      for (int i = tid; i < m; i += kEigThreads) {
        double s = 0.0;
        const double* row = A + (k + 1 + i) + (size_t)(k + 1) * ld;   // A22[i][j] at row[j * ld] (symmetric: row i = column i)
        for (int j = 0; j < m; ++j) s = fma(row[(size_t)j * ld], v[j], s);
        pw[i] = beta * s;
      }
      double vp = 0.0;
      __syncthreads();
      for (int i = tid; i < m; i += kEigThreads) vp = fma(v[i], pw[i], vp);
      const double K = 0.5 * beta * block_sum(vp, red, tid);
      for (int i = tid; i < m; i += kEigThreads) pw[i] = fma(-K, v[i], pw[i]);   // w = p - K v
      __syncthreads();
      // A22 -= v w^T + w v^T
      for (int idx = tid; idx < m * m; idx += kEigThreads) {
        const int i = idx % m, j = idx / m;
        double* el = A + (k + 1 + i) + (size_t)(k + 1 + j) * ld;
        *el = *el - v[i] * pw[j] - pw[i] * v[j];
      }
      __syncthreads();
    }
    for (int i = tid; i < n; i += kEigThreads) d[i] = A[i + (size_t)i * ld];
    if (n >= 2 && tid == 0) e[n - 2] = A[(n - 1) + (size_t)(n - 2) * ld];
    __syncthreads();
    // ---- 3. Sturm counts + 64-way multisection
    double glo = INFINITY, ghi = -INFINITY, emax = 0.0;
    for (int i = 0; i < n; ++i) {   // Gershgorin interval (every thread: n is small next to the rest)
      const double r = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < n ? fabs(e[i]) : 0.0);
      glo = fmin(glo, d[i] - r); ghi = fmax(ghi, d[i] + r);
      emax = fmax(emax, fmax(fabs(d[i]), r));
    }
    const double pivmin = fmax(emax * emax, 1.0) * 2.2250738585072014e-308 * 4.0;   // the smallest pivot the recurrence divides by
    auto count_below = [&](double x) -> int {   // number of eigenvalues of T that are < x
      int c = 0;
      double q = d[0] - x;
      if (fabs(q) < pivmin) q = -pivmin;
      c += q < 0.0;
      for (int i = 1; i < n; ++i) {
        q = d[i] - x - e[i - 1] * e[i - 1] / q;
        if (fabs(q) < pivmin) q = -pivmin;
        c += q < 0.0;
      }
      return c;
    };
    const double span = fmax(ghi - glo, 0.0), pad = 4.0 * 2.220446049250313e-16 * fmax(fabs(glo), fabs(ghi)) + pivmin;
    const int target = tid >> 6, sl = tid & 63;   // target 0: index 0, 1: index n - 1, 2: index c - 1, 3: index c  (c = count(0))
    const int c0 = count_below(0.0);
    int want = target == 0 ? 0 : target == 1 ? n - 1 : target == 2 ? c0 - 1 : c0;
    const bool valid = want >= 0 && want < n;
    want = valid ? want : 0;
    double lo = glo - pad, hi = ghi + pad;
    (void)span;
    for (int round = 0; round < 11; ++round) {
      const double x = lo + (hi - lo) * ((double)(sl + 1) / 65.0);
      cnt[tid] = count_below(x);
      __syncthreads();
      // the eigenvalue of index `want` lies in [x_i, x_{i + 1}) with count(x_i) <= want < count(x_{i + 1})
      double nlo = lo, nhi = hi;
      for (int i = 0; i < 64; ++i) {
        const double xi = lo + (hi - lo) * ((double)(i + 1) / 65.0);
        if (cnt[64 * target + i] <= want) nlo = fmax(nlo, xi); else nhi = fmin(nhi, xi);
      }
      __syncthreads();
      if (nlo < nhi) { lo = nlo; hi = nhi; }
    }
    if (sl == 0) { sect[2 * target] = valid ? 0.5 * (lo + hi) : __builtin_nan(""); }
    __syncthreads();
    if (tid == 0) {
      const double emin_ = sect[0], emaxv = sect[2], below = sect[4], above = sect[6];   // below: largest eigenvalue < 0 (NaN: none)
      double amin = INFINITY;
      if (below == below) amin = fmin(amin, fabs(below));
      if (above == above) amin = fmin(amin, fabs(above));
      T* o = (T*)a.out + p * a.out_stride;
      o[0] = (T)emin_; o[1] = (T)emaxv; o[2] = (T)amin;
    }
  }
}

}  // namespace

size_t eig_lds_bytes(int n, bool matrix_in_lds) {
  size_t b = (size_t)(4 * n + 16) * 8 + kEigThreads * sizeof(int);
  if (matrix_in_lds) b += (size_t)n * (size_t)(n | 1) * 8;
  return (b + 15) & ~(size_t)15;
}
bool eig_needs_global(int n) { return eig_lds_bytes(n, true) > 160 * 1024; }
size_t eig_workspace_bytes(int n) { return (size_t)n * (size_t)(n | 1) * 8; }   // per workgroup slot
int eig_grid(int n, int num_cus) {
  const size_t lds = eig_lds_bytes(n, !eig_needs_global(n));
  int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;   // 256-thread workgroups: 8 of them are the CU's 32 waves
  return num_cus * per_cu;
}

hipError_t launch_qp_eig(const KernelArgs& a, int dtype, int num_cus, void* out, long long out_stride, void* work, size_t work_slot_bytes,
                         long long work_slots, hipStream_t stream) {
  EigArgs e;
  e.n = a.n; e.m_r = a.m_r; e.batch = a.batch;
  e.G = a.G; e.G_stride = a.G_stride; e.G_ld = a.G_ld;
  e.J = a.J; e.J_stride = a.J_stride; e.J_ld = a.J_ld; e.J_row_major = a.J_row_major;
  e.lambda = a.lambda; e.lambda_vec = a.lambda_vec; e.lambda_vec_stride = a.lambda_vec_stride;
  e.skip = a.skip; e.skip_stride = a.skip_stride;
  e.out = out; e.out_stride = out_stride;
  const bool global = eig_needs_global(a.n);
  long long grid = eig_grid(a.n, num_cus);
  if (grid > a.batch) grid = a.batch;
  if (global) {
    if (!work || work_slot_bytes < eig_workspace_bytes(a.n) || work_slots < 1) return hipErrorInvalidValue;
    if (grid > work_slots) grid = work_slots;
    e.work = (double*)work; e.work_stride = (long long)(work_slot_bytes / 8);
  } else {
    e.work = nullptr; e.work_stride = 0;
  }
  if (grid < 1) grid = 1;
  const size_t lds = eig_lds_bytes(a.n, !global);
  if (lds > 160 * 1024) return hipErrorInvalidValue;   // not even the vectors fit (n in the thousands)
  hipError_t err;
  if (dtype == MO_F64) {
    err = hipFuncSetAttribute((const void*)qp_eig_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL((qp_eig_kernel<double>), dim3((unsigned)grid), dim3(kEigThreads), lds, stream, e);
  } else {
    err = hipFuncSetAttribute((const void*)qp_eig_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL((qp_eig_kernel<float>), dim3((unsigned)grid), dim3(kEigThreads), lds, stream, e);
  }
  return hipGetLastError();
}

}  // namespace mo
