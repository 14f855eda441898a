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
  }
  T l1 = 0;
  for (int i = lane; i < a.k; i += 64) l1 += fabs(((const T*)a.b)[p * a.b_stride + i]);                  // nonlinear.cc:203
  l1 = wave_sum(l1);
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

}  // namespace mo
