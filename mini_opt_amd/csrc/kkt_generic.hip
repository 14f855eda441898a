// kkt_generic.hip -- shape-generic LDS-resident KKT kernel for gfx950 (one problem per 256-thread workgroup).
//
// Handles any (n, k, m, m_r) whose reduced KKT matrix fits the CU's 160 KiB LDS, in f64 or f32, and every
// mode of the C ABI (linearise / residual / Newton step / Iterate / full Solve).  The fixed-shape fused kernels
// (kkt_fused.hip) are the fast path for BASELINE.json's configs; this kernel is the general path and the
// on-device reference they are A/B-tested against.
//
// Reference arithmetic reproduced (citations into /root/reference):
//   J^T J / J^T r / lambda      residual.hpp:206-224, nonlinear.cc:182-189
//   KKT residual                qp.cc:391-420           errors qp.cc:423-437
//   reduced-KKT assembly        qp.cc:281-298
//   LDL^T                       qp.cc:302 (Eigen LDLT<MatrixXd,Lower>): here a right-looking LDL^T in natural order.
//                               Eigen's "pivoting" only orders by the ORIGINAL |H_ii| (its update is left-looking), which
//                               changes rounding, not the solution; zero-pivot rules are kept (zero pivot tolerated iff
//                               the column below is zero, failure if a non-zero pivot follows).
//   solve + back-substitution   qp.cc:337-363 by a direct solve (the explicit inverse of qp.cc:310-311 is not formed)
//   alpha / mu / mu_affine      qp.cc:485-537       Iterate qp.cc:153-201       Solve qp.cc:100-151, 439-482
//
// LDS layout: H is column-major with an ODD leading dimension (conflict-free column AND row access with ds_read_b64),
// followed by the state, residual, direction, rhs and constraint vectors and a small row-chunk of J.
#include <math.h>

#include <stdlib.h>
#include <type_traits>

#include "mo_kernels.h"

// Phase stamps exist only in the diagnostic build of tools/phase_timer_generic.hip; the product kernel executes none.
#ifdef MO_GENERIC_STAMPS
#define MO_GSTAMP(i)                                                                              \
  do {                                                                                            \
    __syncthreads();                                                                              \
    if (threadIdx.x == 0) {                                                                       \
      const unsigned long long t__ = __builtin_amdgcn_s_memtime();                                \
      gstamp_acc[i] += t__ - gstamp_prev;                                                         \
      gstamp_prev = t__;                                                                          \
    }                                                                                             \
  } while (0)
#else
#define MO_GSTAMP(i) do { } while (0)
#endif

namespace mo {
namespace {

// Every stage is inlined into its kernel: with R = 9 / 12 hipcc otherwise keeps assemble_and_factor (two call sites in MODE_SOLVE) as a real
// function, and the kernel that called it returned alpha_dual = 1 for every problem -- the records a caller reads after the call were lost.
#define MO_INLINE __attribute__((always_inline))
// Diagnostic builds only (tools/alpha_dual_probe/): MO_GENERIC_PROBE_CALL keeps assemble_and_factor a real function again, MO_GENERIC_PROBE
// makes Iterate / Solve report what compute_alpha saw (z[0], dz[0], the dual step length in LDS) in the three record slots a
// COMPLEMENTARITY iteration leaves NaN.  The product build defines neither.
#if defined(MO_GENERIC_PROBE_CALL)
#define MO_INLINE_FACTOR __attribute__((noinline))
#elif defined(MO_GENERIC_PROBE_AUTO)
#define MO_INLINE_FACTOR            /* the inliner's own choice, as before commit a1d56f0 */
#else
#define MO_INLINE_FACTOR MO_INLINE
#endif
#ifdef MO_GENERIC_PROBE_BYVAL
#define MO_FACTOR_WS(T) const Ws<T>     /* the workspace descriptor by value (registers) instead of by reference (the caller's scratch) */
#else
#define MO_FACTOR_WS(T) const Ws<T>&
#endif

#ifdef MO_GENERIC_STAMPS
__device__ unsigned long long g_nd_stamps[16];   // diagnostic build only: where newton_direction spends its time
#define MO_NDSTAMP(i)                                                              \
  do {                                                                             \
    __syncthreads();                                                               \
    if (threadIdx.x == 0) {                                                        \
      const unsigned long long t__ = __builtin_amdgcn_s_memtime();                 \
      atomicAdd(&g_nd_stamps[i], t__ - nd_prev);                                   \
      nd_prev = t__;                                                               \
    }                                                                              \
  } while (0)
#define MO_FBSTAMP(i)                                                              \
  do {                                                                             \
    __syncthreads();                                                               \
    if (threadIdx.x == 0) {                                                        \
      const unsigned long long t__ = __builtin_amdgcn_s_memtime();                 \
      atomicAdd(&g_nd_stamps[i], t__ - fb_prev);                                   \
      fb_prev = t__;                                                               \
    }                                                                              \
  } while (0)
#else
#define MO_NDSTAMP(i) do { } while (0)
#define MO_FBSTAMP(i) do { } while (0)
#endif

// Workgroup size is chosen at launch: 256 threads (4 waves) for large systems, ONE wave for small ones (n + k <= 48), where a
// 256-thread workgroup would idle on P-long loops and up to 32 single-wave workgroups fit a CU instead of 8.
constexpr int kMaxThreads = 256;
#define kThreads ((int)blockDim.x)
#define kWaves ((int)(blockDim.x >> 6))

template <typename T> struct Ws {
  T* H; int ldh;
  T* vars; T* res; T* delta; T* daff;  // V each
  T* cvec;                             // n
  T* beq;                              // k
  T* rhs; T* invd;                     // P each
  T* ca; T* cb;                        // m each
  T* Jc; T* rc; int chunk_rows;        // J row chunk
  int region;                          // LARGE: elements of the panel region that starts at Jc
  T* red;                              // 16 scalars
  int* cv;                             // m
  int* iflag;                          // 8
};

__host__ __device__ inline int odd_ld(int P) { return P | 1; }

__host__ __device__ inline int chunk_rows_for(int n, int m_r, int elem) {
  if (m_r <= 0) return 0;
  int cr = 8192 / (n * elem);
  if (cr < 4) cr = 4;
  if (cr > m_r) cr = m_r;
  return cr;
}

template <typename T>
__host__ __device__ inline size_t ws_elems(int n, int k, int m, int m_r) {
  const int P = n + k, V = n + 2 * m + k;
  const int cr = chunk_rows_for(n, m_r, (int)sizeof(T));
  size_t e = (size_t)P * odd_ld(P) + 4 * (size_t)V + n + k + 2 * (size_t)P + 2 * (size_t)m + (size_t)cr * n + cr + 16;
  return e;
}

// ---- LARGE systems (P = n + k beyond the register-distributed factorisation's 192, or an H that does not fit the LDS): H lives in a global
// workspace of the workgroup (P x ldh, column-major, plan-owned, L2-resident: one per workgroup of the persistent grid), everything else stays
// in LDS.  The J row chunk and the factorisation's column panel share one LDS region (they are never live together).
// (leading dimension of the global H: a multiple of 16 elements -- every column starts on a 128-byte line (fp64), so the 16-row strips the
// factorisation and the solves read are whole lines; with the first version's even ld every other column straddled two: FETCH_SIZE 1.5 x)
__host__ __device__ inline int large_ld(int P) { return (P + 15) & ~15; }
#ifndef MO_LARGE_NB_MAX
#define MO_LARGE_NB_MAX 32
#endif
__host__ __device__ inline int panel_cols_for(int n, int k, int m, int m_r, int elem) {
  // widest panel (32 / 16 / 8 columns of P | 1 rows) that fits beside the vectors
  const int P = n + k, V = n + 2 * m + k;
  const size_t vec = (4 * (size_t)V + n + k + 2 * (size_t)P + 2 * (size_t)m + 16) * elem + (size_t)(m + 8) * sizeof(int) + 64;
  // two workgroups per CU count for more than a wide panel (n = 256, k = 40, m = 128: 32 columns and one workgroup per CU 19.9 ms for 2 048
  // steps, 16 columns and two 14.0 ms, 8 columns and three 22.4 ms): 32 or 16 columns if two workspaces fit the LDS, else the widest that fits
#ifndef MO_LARGE_NO_PAIR
  for (int nb = MO_LARGE_NB_MAX; nb >= 16; nb >>= 1)
    if (2 * (vec + (size_t)(P | 1) * nb * elem + 64) <= 160 * 1024) return nb;
#endif
  for (int nb = MO_LARGE_NB_MAX; nb >= 8; nb >>= 1)
    if (vec + (size_t)(P | 1) * nb * elem <= 160 * 1024) return nb;
  return 0;
}
template <typename T>
__host__ __device__ inline size_t ws_elems_large(int n, int k, int m, int m_r) {
  const int P = n + k, V = n + 2 * m + k;
  const int nb = panel_cols_for(n, k, m, m_r, (int)sizeof(T));
  return 4 * (size_t)V + n + k + 2 * (size_t)P + 2 * (size_t)m + (size_t)(P | 1) * nb + 16;
}
// rows of J per chunk: as many as the panel region holds (each chunk costs one read-modify-write of the whole lower triangle of G in the
// global workspace, so few large chunks: 37 rows instead of 4 at n = 256 took the kernel's HBM traffic from 96 GB to a quarter of it)
__host__ __device__ inline int chunk_rows_large(int n, int k, int m, int m_r, int elem) {
  if (m_r <= 0) return 0;
  const int P = n + k;
  int cr = (int)(((size_t)(P | 1) * panel_cols_for(n, k, m, m_r, elem)) / (size_t)(n + 1));
  return cr > m_r ? m_r : cr;
}
template <typename T>
__device__ inline void carve_large(Ws<T>& w, char* smem, T* H_global, int n, int k, int m, int m_r) {
  const int P = n + k, V = n + 2 * m + k;
  T* p = reinterpret_cast<T*>(smem);
  w.ldh = large_ld(P);
  w.H = H_global;
  w.vars = p; p += V; w.res = p; p += V; w.delta = p; p += V; w.daff = p; p += V;
  w.cvec = p; p += n; w.beq = p; p += k;
  w.rhs = p; p += P; w.invd = p; p += P;
  w.ca = p; p += m; w.cb = p; p += m;
  w.chunk_rows = chunk_rows_large(n, k, m, m_r, (int)sizeof(T));
  const int nb = panel_cols_for(n, k, m, m_r, (int)sizeof(T));
  w.region = (P | 1) * nb;
  w.Jc = p; w.rc = p + (size_t)w.chunk_rows * n; p += (size_t)(P | 1) * nb;   // (the panel overlays the J chunk: chunk_rows (n + 1) <= panel)
  w.red = p; p += 16;
  w.cv = reinterpret_cast<int*>(p);
  w.iflag = w.cv + m;
}

template <typename T>
__device__ inline void carve(Ws<T>& w, char* smem, int n, int k, int m, int m_r) {
  const int P = n + k, V = n + 2 * m + k;
  T* p = reinterpret_cast<T*>(smem);
  w.ldh = odd_ld(P);
  w.H = p; p += (size_t)P * w.ldh;
  w.vars = p; p += V; w.res = p; p += V; w.delta = p; p += V; w.daff = p; p += V;
  w.cvec = p; p += n; w.beq = p; p += k;
  w.rhs = p; p += P; w.invd = p; p += P;
  w.ca = p; p += m; w.cb = p; p += m;
  w.chunk_rows = chunk_rows_for(n, m_r, (int)sizeof(T));
  w.region = 0;
  w.Jc = p; p += (size_t)w.chunk_rows * n; w.rc = p; p += w.chunk_rows;
  w.red = p; p += 16;
  w.cv = reinterpret_cast<int*>(p);
  w.iflag = w.cv + m;
}

// ---- small helpers -------------------------------------------------------------------------------------------
__device__ inline double rl(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ inline float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// a value every lane holds alike (read from one LDS address), as a scalar: decisions on it become scalar branches instead of EXEC-masked regions
__device__ inline double uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ inline float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline double absT(double v) { return fabs(v); }
__device__ inline float absT(float v) { return fabsf(v); }
__device__ inline double sqrtT(double v) { return sqrt(v); }
__device__ inline float sqrtT(float v) { return sqrtf(v); }
__device__ inline double fmaT(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ inline float fmaT(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// A copy the optimiser cannot see through.  The LARGE kernel calls this on the thread index at the top of every phase: a predicate of
// the lane index (lane == j, lane > jj, tid < wd ...) is loop-invariant, so LLVM computes each of the few hundred in the unrolled phases ONCE,
// in front of the problem loop, and keeps all the masks live through the whole kernel -- 480 live SGPRs, 1 100 spills to VGPR lanes, and
// the VGPRs holding those lanes spilled to scratch in turn.  From an opaque copy each mask is one v_cmp at its use.
__device__ inline int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
template <typename T> __device__ inline T nanT() { return (T)__builtin_nan(""); }
template <typename T> __device__ inline bool finiteT(T v) { return __builtin_isfinite(v); }

// Orders one wave's own LDS traffic: values another lane of the SAME wave stored are visible to the loads that follow.
__device__ inline void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <typename T> __device__ inline T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T> __device__ inline T wave_min(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o, 64); v = u < v ? u : v; }
  return v;
}

// ---- phases --------------------------------------------------------------------------------------------------

// H <- 0 ; H.lower(n x n) <- G.lower ; H[n.., 0..n) <- A_eq ; cvec <- c ; beq <- b_eq      (qp.cc:47, 289-292)
template <typename T>
__device__ MO_INLINE void load_qp(const Ws<T>& w, int n, int k, const T* G, int G_ld, const T* c, const T* A, int A_ld,
                        const T* b, int tid) {
  const int P = n + k;
  if (w.region > 0) {   // LARGE (H in global memory): only the y-y block needs zeros -- G's lower triangle is written below or by accumulate_jtj
    for (int j = n; j < P; ++j)    // (plain stores there), the A rows below, and nothing reads above the diagonal
      for (int i = n + tid; i < P; i += kThreads) w.H[i + (size_t)j * w.ldh] = (T)0;
  } else {
    for (int idx = tid; idx < P * w.ldh; idx += kThreads) w.H[idx] = (T)0;
  }
  __syncthreads();
  // Copies into H go in groups of eight loads followed by eight stores: with H in a global workspace the compiler must assume that a store to H
  // may alias the next load, and a load-store-load-store chain exposes one memory latency per element.
  if (G) {
    for (int j0 = 0; j0 < n; j0 += 8)
      for (int i = j0 + tid; i < n; i += kThreads) {
        T v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (j0 + e < n && i >= j0 + e) ? G[i + (size_t)(j0 + e) * G_ld] : (T)0;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (j0 + e < n && i >= j0 + e) w.H[i + (size_t)(j0 + e) * w.ldh] = v[e];
      }
    for (int i = tid; i < n; i += kThreads) w.cvec[i] = c[i];
  }
  if (k > 0) {   // the k x n block of A_eq as one flat index (k is small: a loop over columns leaves most of the workgroup idle in every round trip)
    for (int base = 0; base < n * k; base += 8 * kThreads) {
      T v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = base + e * kThreads + tid, j = idx / k, q = idx - j * k;
        v[e] = idx < n * k ? A[q + (size_t)j * A_ld] : (T)0;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = base + e * kThreads + tid, j = idx / k, q = idx - j * k;
        if (idx < n * k) w.H[n + q + (size_t)j * w.ldh] = v[e];
      }
    }
  }
  for (int q = tid; q < k; q += kThreads) w.beq[q] = b[q];
  __syncthreads();
}

template <typename T, int TG, int RN>
__device__ inline void jtj_tile_rows_impl(const Ws<T>& w, int n, int rows, int tid) {
  const int ti = tid & (TG - 1), tj = tid / TG;
  T acc[RN * (RN + 1) / 2];
#pragma unroll
  for (int e = 0; e < RN * (RN + 1) / 2; ++e) acc[e] = (T)0;
  int ic[RN], jc[RN];
#pragma unroll
  for (int b = 0; b < RN; ++b) {   // clamped column indices: out-of-range operands are read (in bounds) and their results never stored
    ic[b] = ti + TG * b < n ? ti + TG * b : n - 1;
    jc[b] = tj + TG * b < n ? tj + TG * b : n - 1;
  }
  for (int q = 0; q < rows; ++q) {
    const T* row = w.Jc + (size_t)q * n;
    T a[RN], b[RN];
#pragma unroll
    for (int e = 0; e < RN; ++e) { a[e] = row[ic[e]]; b[e] = row[jc[e]]; }
#pragma unroll
    for (int bi = 0; bi < RN; ++bi)
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj) acc[bi * (bi + 1) / 2 + bj] += a[bi] * b[bj];
  }
#pragma unroll
  for (int bi = 0; bi < RN; ++bi)
#pragma unroll
    for (int bj = 0; bj <= bi; ++bj) {
      const int i = ti + TG * bi, j = tj + TG * bj;
      if (i < n && j <= i) w.H[i + (size_t)j * w.ldh] += acc[bi * (bi + 1) / 2 + bj];
    }
}
// LARGE: J^T J over 96 x 96 super-blocks of G (6 x 6 tiles of 16 x 16), lower block triangle only, ON THE MATRIX CORES.  G is in global
// memory, so the super-block is the OUTER loop: its accumulators stay in registers over all rows of J and are stored once.  Per super-block
// only the columns [I0, I0 + wi) and [J0, J0 + wj) of J are staged (wj = 0 on the diagonal), `rows` of them at a time, rows padded with zeros
// to a multiple of 4 (one v_mfma_*_16x16x4 consumes 4 rows of J).  Each of the four waves owns a 3 x 3 quadrant of the tiles.  A tile is
// computed TRANSPOSED, D[jl][il] = sum_q J[q][J0 + jl] J[q][I0 + il] (A operand from the J block, B operand from the I block): the lanes of
// a 16-lane row then hold 16 consecutive ROWS of G and the stores to the column-major workspace are whole 128-byte segments.  The staged rows
// have a stride of (columns + 16) elements: a multiple of 32 plus 16, so the four rows an operand read touches fall on disjoint LDS banks.
// (Round 3, first version: 16 x 16 threads x 6 x 6 register blocks on the VALU, 12 LDS reads per 36 FMAs per row of J and thread.)
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
template <typename T> struct Mfma16;
template <> struct Mfma16<double> {
  typedef v4f64 Acc;
  __device__ static inline Acc mac(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  __device__ static inline int row(int g, int t) { return g + 4 * t; }    // C/D fragment: lane (g, l), register t <-> (row g + 4 t, col l)
};
template <> struct Mfma16<float> {
  typedef v4f32 Acc;
  __device__ static inline Acc mac(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  __device__ static inline int row(int g, int t) { return 4 * g + t; }    // (row 4 g + t, col l)
};
__host__ __device__ inline int large_jtj_stride(int wc) { return wc + 16 + ((32 - (wc & 31)) & 31); }   // = 16 (mod 32), >= wc + 16
#ifndef MO_LARGE_TQ
#define MO_LARGE_TQ 4
#endif
#ifndef MO_LARGE_JPF
#define MO_LARGE_JPF 2   // 4-row steps of J whose loads are issued together on the direct path (16 loads a lane at 2)
#endif
// G = J^T J (+ lambda on the diagonal) to the lower triangle of H, c = J^T r to cvec, |r|^2 to red[8], on the matrix cores over 128-wide
// super-blocks.  Round 4: a PACKED or padded ROW-MAJOR J is read straight from global memory into the MFMA operands -- lane (l, g) of a
// step takes row q + g, columns 16 e + l: every load instruction is four whole 128-byte segments -- with the loads of MO_LARGE_JPF steps
// issued together; the LDS staging of the first version (a barrier and a round of latency per ~16-row chunk, three passes) is kept for the
// column-major layout only.  Both paths feed the same steps in the same order (row q + g to lane group g), so every layout gives the same
// bits.  c comes from the operands the diagonal blocks hold anyway (one FMA per loaded value, reduced over the four lane groups at the end).
template <typename T>
__device__ inline void accumulate_jtj_large(const Ws<T>& w, int n, int m_r, const T* J, int J_ld, int row_major, const T* r, T lambda, int tid) {
  typedef Mfma16<T> MF;
  typedef typename MF::Acc Acc;
  constexpr int TQ = MO_LARGE_TQ, SBW = 32 * TQ;   // a wave's quadrant is TQ x TQ tiles, the super-block 2 TQ x 2 TQ
  constexpr int PF = MO_LARGE_JPF;
  tid = opaque(tid);
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l = lane & 15;
  const int qa = (wave >> 1) & 1, qb = wave & 1;             // this wave's quadrant of the tiles: I tiles TQ qa .., J tiles TQ qb ..
  if (wave == 1) {                                           // |r|^2: the wave without tiles on the first diagonal block
    T racc = (T)0;
#pragma unroll 4
    for (int q = lane; q < m_r; q += 64) racc = fmaT(r[q], r[q], racc);
    racc = wave_sum(racc);
    if (lane == 0) w.red[8] = racc;
  }
  for (int I0 = 0; I0 < n; I0 += SBW) {
    const int wi = n - I0 < SBW ? n - I0 : SBW;
    for (int J0 = 0; J0 <= I0; J0 += SBW) {
      const bool diag = J0 == I0;
      const int wj = diag ? 0 : SBW, wc = wi + wj;          // (J0 < I0: a full block of columns)
      // which of the wave's TQ x TQ tiles exist (wave-uniform): inside the block's columns, and on / below the diagonal of a diagonal block
      const int jw = diag ? wi : wj, joff = diag ? 0 : wi;
      bool need[TQ][TQ];
      bool any = false;
#pragma unroll
      for (int a_ = 0; a_ < TQ; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < TQ; ++b_) {
          need[a_][b_] = wave < 4 && 16 * (TQ * qa + a_) < wi && 16 * (TQ * qb + b_) < jw && (!diag || TQ * qb + b_ <= TQ * qa + a_);
          any = any || need[a_][b_];
        }
      const bool do_c = diag && qb == 0 && wave < 4 && 16 * TQ * qa < wi;   // this wave also accumulates c for its I columns (wave-uniform)
      const bool same = diag && qa == qb;                                   // the J operands are the I operands
      int ic[TQ], jc[TQ];                                    // clamped local columns: out-of-range operands are read (in bounds), their results never stored
#pragma unroll
      for (int e = 0; e < TQ; ++e) {
        const int ci = 16 * (TQ * qa + e) + l, cj = 16 * (TQ * qb + e) + l;
        ic[e] = ci < wi ? ci : wi - 1;
        jc[e] = joff + (cj < jw ? cj : jw - 1);
      }
      Acc acc[TQ][TQ];
      T cacc[TQ];
#pragma unroll
      for (int a_ = 0; a_ < TQ; ++a_) {
        cacc[a_] = (T)0;
#pragma unroll
        for (int b_ = 0; b_ < TQ; ++b_) acc[a_][b_] = Acc{(T)0, (T)0, (T)0, (T)0};
      }
      // tile rows of this wave with any needed tile (wave-uniform); the tiles of a row are computed together: all four, or the ones on / below
      // the diagonal where the wave's quadrant sits on it (TRI, a compile-time flag: no per-tile branches in the loop).  Tiles beyond the
      // block's columns are computed from clamped operands and never stored.
      int na = 0;
#pragma unroll
      for (int a_ = 0; a_ < TQ; ++a_) {
        bool row_any = false;
#pragma unroll
        for (int b_ = 0; b_ < TQ; ++b_) row_any = row_any || need[a_][b_];
        if (row_any) na = a_ + 1;
      }
      // one step: four rows of J (row q + g in lane group g)
      auto step = [&](auto tri, const T (&bi)[TQ], const T (&aj)[TQ], T rv) {
        constexpr bool TRI = decltype(tri)::value;
#pragma unroll
        for (int a_ = 0; a_ < TQ; ++a_) {
          if (a_ >= na) continue;                            // (scalar)
#pragma unroll
          for (int b_ = 0; b_ < TQ; ++b_)
            if (!TRI || b_ <= a_) acc[a_][b_] = MF::mac(aj[b_], bi[a_], acc[a_][b_]);
        }
        if (do_c) {
#pragma unroll
          for (int e = 0; e < TQ; ++e) cacc[e] = fmaT(bi[e], rv, cacc[e]);
        }
      };
      auto run = [&](auto tri) {
        constexpr bool TRI = decltype(tri)::value;           // (TRI <=> same: the J operands are the I operands)
        if (row_major) {
          if (any || do_c) {
            int oi[TQ], oj[TQ];
#pragma unroll
            for (int e = 0; e < TQ; ++e) { oi[e] = I0 + ic[e]; oj[e] = (diag ? I0 : J0) + (jc[e] - joff); }
            for (int q0 = 0; q0 < m_r; q0 += 4 * PF) {
              T bi[PF][TQ], aj[PF][TQ], rv[PF];
              const bool whole = q0 + 4 * PF <= m_r;         // (scalar)
#pragma unroll
              for (int s_ = 0; s_ < PF; ++s_) {
                const int qq = q0 + 4 * s_ + g;
                const int qc = (whole || qq < m_r) ? qq : m_r - 1;
                const T* rowp = J + (size_t)qc * J_ld;
#pragma unroll
                for (int e = 0; e < TQ; ++e) bi[s_][e] = rowp[oi[e]];
                if (!TRI) {
#pragma unroll
                  for (int e = 0; e < TQ; ++e) aj[s_][e] = rowp[oj[e]];
                }
                rv[s_] = do_c ? r[qc] : (T)0;
              }
              if (!whole) {                                  // the last rows: lanes beyond m_r contribute zeros
#pragma unroll
                for (int s_ = 0; s_ < PF; ++s_) {
                  const bool ok = q0 + 4 * s_ + g < m_r;
#pragma unroll
                  for (int e = 0; e < TQ; ++e) { bi[s_][e] = ok ? bi[s_][e] : (T)0; if (!TRI) aj[s_][e] = ok ? aj[s_][e] : (T)0; }
                }
              }
#pragma unroll
              for (int s_ = 0; s_ < PF; ++s_) {
                if (q0 + 4 * s_ >= m_r) continue;            // (scalar)
                if (TRI) step(tri, bi[s_], bi[s_], rv[s_]);
                else step(tri, bi[s_], aj[s_], rv[s_]);
              }
            }
          }
        } else {
          const int ws = large_jtj_stride(wc);
          int CR = (w.region / (ws + 1)) & ~3;
          if (CR > ((m_r + 3) & ~3)) CR = (m_r + 3) & ~3;
          if (CR < 4) return;                                // (cannot happen: generic_large_lds_bytes refuses a region below four staged rows -- but never loop on a zero step)
          T* const rcs = w.Jc + (size_t)CR * ws;             // r of the staged rows
          for (int q0 = 0; q0 < m_r; q0 += CR) {
            const int rows = m_r - q0 < CR ? m_r - q0 : CR, rows4 = (rows + 3) & ~3;
            // staging: every thread issues a batch of loads before it stores any of them (one round of global-memory latency per batch)
            for (int base = 0; base < rows * wc; base += 8 * kThreads) {
              T v[8];
              int dst[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int idx = base + e * kThreads + tid;
                dst[e] = -1;
                v[e] = (T)0;
                if (idx < rows * wc) {
                  const int cx = idx / rows, q = idx - cx * rows;
                  const int col = cx < wi ? I0 + cx : J0 + cx - wi;
                  v[e] = J[(size_t)col * J_ld + q0 + q];
                  dst[e] = q * ws + cx;
                }
              }
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (dst[e] >= 0) w.Jc[dst[e]] = v[e];
            }
            for (int idx = tid; idx < (rows4 - rows) * wc; idx += kThreads) w.Jc[(size_t)(rows + idx / wc) * ws + idx % wc] = (T)0;
            if (diag)
              for (int idx = tid; idx < rows4; idx += kThreads) rcs[idx] = idx < rows ? r[q0 + idx] : (T)0;
            __syncthreads();
            if (any || do_c) {
              for (int q = 0; q < rows4; q += 4) {
                const T* row = w.Jc + (size_t)(q + g) * ws;
                T bi[TQ], aj[TQ];
#pragma unroll
                for (int e = 0; e < TQ; ++e) { bi[e] = row[ic[e]]; aj[e] = row[jc[e]]; }
                step(tri, bi, aj, do_c ? rcs[q + g] : (T)0);
              }
            }
            __syncthreads();
          }
        }
      };
      if (same) run(std::true_type{});                       // (wave-uniform: every wave runs the same number of barriers either way)
      else run(std::false_type{});
      if (do_c) {                                            // c: the four lane groups' partial sums, in one fixed order
#pragma unroll
        for (int e = 0; e < TQ; ++e) {
          T v = cacc[e];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          const int ci = 16 * (TQ * qa + e) + l;
          if (g == 0 && ci < wi) w.cvec[I0 + ci] = v;
        }
      }
#pragma unroll
      for (int a_ = 0; a_ < TQ; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < TQ; ++b_) {
          if (!need[a_][b_]) continue;
          const int i = I0 + 16 * (TQ * qa + a_) + l;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int j = J0 + 16 * (TQ * qb + b_) + MF::row(g, t);
            if (i < n && j < n && j <= i) {
              T v = acc[a_][b_][t];
              if (i == j && lambda > (T)0) v += lambda;      // (nonlinear.cc:182-189: the damping, in the same store)
              w.H[i + (size_t)j * w.ldh] = v;
            }
          }
        }
    }
  }
}
// H.lower(n x n) += J^T J, cvec = J^T r, diag += lambda; returns 0.5|r|^2 in w.red[8]   (residual.hpp:206-225,
// nonlinear.cc:182-189).  H must be zero in its n x n block and cvec is overwritten.
template <typename T, int TG, int R, bool LARGE = false>
__device__ MO_INLINE void accumulate_jtj(const Ws<T>& w, int n, int m_r, const T* J, int J_ld, int row_major, const T* r,
                               T lambda, int tid) {
  for (int i = tid; i < n; i += kThreads) w.cvec[i] = (T)0;
  if (tid == 0) w.red[8] = (T)0;
  __syncthreads();
  if constexpr (LARGE) {
    accumulate_jtj_large<T>(w, n, m_r, J, J_ld, row_major, r, lambda, tid);
    __threadfence_block();   // G is read by other threads than the ones that stored it
    __syncthreads();
    if (tid == 0) w.red[8] *= (T)0.5;
    __syncthreads();
    return;
  }
  const int CR = w.chunk_rows;
  for (int q0 = 0; q0 < m_r; q0 += CR) {
    const int rows = (m_r - q0 < CR) ? (m_r - q0) : CR;
    if (row_major) {
      for (int idx = tid; idx < rows * n; idx += kThreads) {
        const int q = idx / n, i = idx - q * n;
        w.Jc[idx] = J[(size_t)(q0 + q) * J_ld + i];
      }
    } else {
      for (int idx = tid; idx < rows * n; idx += kThreads) {
        const int i = idx / rows, q = idx - i * rows;
        w.Jc[q * n + i] = J[(size_t)i * J_ld + q0 + q];
      }
    }
    for (int idx = tid; idx < rows; idx += kThreads) w.rc[idx] = r[q0 + idx];
    __syncthreads();
    // 2-D register tiling: thread (ti, tj) of a TG x TG grid accumulates G(i, j) for i = ti + TG bi, j = tj + TG bj, bi >= bj: per row of
    // J it reads RN + RN operands (broadcasts) for RN (RN + 1) / 2 FMAs, instead of five LDS reads for four FMAs
    jtj_tile_rows_impl<T, TG, R>(w, n, rows, tid);   // n <= n + k <= TG R
    for (int i = tid; i < n; i += kThreads) {
      T acc = 0;
      for (int q = 0; q < rows; ++q) acc += w.Jc[(size_t)q * n + i] * w.rc[q];
      w.cvec[i] += acc;
    }
    if (tid == 0) {
      T acc = 0;
      for (int q = 0; q < rows; ++q) acc += w.rc[q] * w.rc[q];
      w.red[8] += acc;
    }
    __syncthreads();
  }
  if (lambda > (T)0)
    for (int i = tid; i < n; i += kThreads) w.H[i + (size_t)i * w.ldh] += lambda;
  if (tid == 0) w.red[8] *= (T)0.5;
  __syncthreads();
}

// EvaluateKKTConditions, qp.cc:391-420.  H must hold G (lower) and A_eq (no Sigma yet).
template <typename T> __device__ inline void eval_kkt_large(const Ws<T>& w, int n, int k, int m, bool include_ineq, int tid);
template <bool LARGE = false, typename T>
__device__ MO_INLINE void eval_kkt(const Ws<T>& w, int n, int k, int m, bool include_ineq, int tid) {
  if constexpr (LARGE) { eval_kkt_large<T>(w, n, k, m, include_ineq, tid); return; }
  const T* x = w.vars; const T* s = w.vars + n; const T* y = w.vars + n + m; const T* z = w.vars + n + m + k;
  T* r_d = w.res; T* r_comp = w.res + n; T* r_pe = w.res + n + m; T* r_pi = w.res + n + m + k;
  for (int i = tid; i < n; i += kThreads) {
    T acc = 0;
    // (unrolled by 8: with H in a global workspace the eight loads of a group are in flight together; the order of the sum is unchanged)
#pragma unroll 8
    for (int j = 0; j <= i; ++j) acc += w.H[i + (size_t)j * w.ldh] * x[j];       // selfadjointView<Lower>, :404
#pragma unroll 8
    for (int j = i + 1; j < n; ++j) acc += w.H[j + (size_t)i * w.ldh] * x[j];
    T rd = acc + w.cvec[i];
    if (k > 0) {
      T acc2 = 0;
#pragma unroll 8
      for (int q = 0; q < k; ++q) acc2 += w.H[n + q + (size_t)i * w.ldh] * y[q];  // :406
      rd -= acc2;
    }
    if (include_ineq) {
      for (int c = 0; c < m; ++c)                                                 // :413-415, reference order per variable
        if (w.cv[c] == i) rd -= w.ca[c] * z[c];
    }
    r_d[i] = rd;
  }
  for (int q = tid; q < k; q += kThreads) {                                       // :408
    T acc = 0;
#pragma unroll 8
    for (int j = 0; j < n; ++j) acc += w.H[n + q + (size_t)j * w.ldh] * x[j];
    r_pe[q] = acc + w.beq[q];
  }
  if (include_ineq) {
    for (int c = tid; c < m; c += kThreads) {                                     // :416-417
      r_pi[c] = w.ca[c] * x[w.cv[c]] + w.cb[c] - s[c];
      r_comp[c] = s[c] * z[c];
    }
  }
  __syncthreads();
}

// LARGE (H in global memory): the same residuals from ONE coalesced pass over the lower triangle of [G; A_eq].  The loop above reads the
// part of row i beyond the diagonal as column i -- a different 128-byte line per lane and load -- and gives r_pe to k threads with n
// sequential loads each: 10 % of the n = 256 step.  Here a wave owns every fourth group of four columns, its lanes run down the rows (one
// 512-byte segment per load), the loads of a group are issued together, and each element H[i][j] is used twice: times x[j] for row i's
// sum (r_d below the diagonal, r_pe in the rows of A_eq), and times x[i] (or -y[i - n]) for column j's sum, which one wave reduction per
// column finishes.  The sums run in a different (fixed) order than Eigen's; the rest of the function is the LDS version's.
template <typename T>
__device__ inline void eval_kkt_large(const Ws<T>& w, int n, int k, int m, bool include_ineq, int tid) {
  constexpr int RT = 8, CB = 4;       // rows per lane of a pass (64 RT rows), columns whose loads are in flight together
  const int P = n + k, ldh = w.ldh;
  const T* x = w.vars; const T* s = w.vars + n; const T* y = w.vars + n + m; const T* z = w.vars + n + m + k;
  T* r_d = w.res; T* r_comp = w.res + n; T* r_pe = w.res + n + m; T* r_pi = w.res + n + m + k;
  tid = opaque(tid);
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = kThreads >> 6;
  T* const rpart = w.Jc;                     // nw x P: the waves' partial row sums   (the panel region: nw P + n <= 8 (P | 1) elements)
  T* const csum = rpart + (size_t)nw * P;    // n column sums; a column belongs to one wave
  for (int i = tid; i < n; i += kThreads) csum[i] = (T)0;
  __syncthreads();
  for (int ib = 0; ib < P; ib += 64 * RT) {
    T racc[RT], xx[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int i = ib + lane + 64 * t;
      racc[t] = (T)0;
      xx[t] = i < n ? x[i] : (i < P ? -y[i - n] : (T)0);
    }
    const int jmax = n < ib + 64 * RT ? n : ib + 64 * RT;    // columns with a row of this pass on or below the diagonal
    for (int j0 = wave * CB; j0 < jmax; j0 += nw * CB) {
      T v[CB][RT];
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int j = j0 + c;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          v[c][t] = (T)0;
          if (j < jmax && ib + 64 * t + 63 >= j && ib + 64 * t < P) {   // (scalar)
            const int i = ib + lane + 64 * t;
            v[c][t] = w.H[(size_t)(i < P ? i : P - 1) + (size_t)j * ldh];
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int j = j0 + c;
        if (j >= jmax) continue;                                         // (scalar)
        const T xj = x[j];
        T cp = (T)0;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          if (ib + 64 * t + 63 >= j && ib + 64 * t < P) {                 // (scalar)
            const int i = ib + lane + 64 * t;
            const T vv = (i >= j && i < P) ? v[c][t] : (T)0;             // (above the diagonal the workspace holds nothing)
            racc[t] = fmaT(vv, xj, racc[t]);
            cp = fmaT(i > j ? vv : (T)0, xx[t], cp);
          }
        }
        cp = wave_sum(cp);
        if (lane == 0) csum[j] += cp;
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int i = ib + lane + 64 * t;
      if (i < P) rpart[(size_t)wave * P + i] = racc[t];
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += kThreads) {
    T rd = (T)0;
    for (int wv = 0; wv < nw; ++wv) rd += rpart[(size_t)wv * P + i];
    rd += csum[i];                                                      // (with -A_eq^T y, :406)
    rd += w.cvec[i];
    if (include_ineq) {
      for (int c = 0; c < m; ++c)                                                 // :413-415, reference order per variable
        if (w.cv[c] == i) rd -= w.ca[c] * z[c];
    }
    r_d[i] = rd;
  }
  for (int q = tid; q < k; q += kThreads) {                                       // :408
    T acc = (T)0;
    for (int wv = 0; wv < nw; ++wv) acc += rpart[(size_t)wv * P + n + q];
    r_pe[q] = acc + w.beq[q];
  }
  if (include_ineq) {
    for (int c = tid; c < m; c += kThreads) {                                     // :416-417
      r_pi[c] = w.ca[c] * x[w.cv[c]] + w.cb[c] - s[c];
      r_comp[c] = s[c] * z[c];
    }
  }
  __syncthreads();
}

// ComputeErrors, qp.cc:423-437 -> w.red[0..3]
template <typename T>
__device__ MO_INLINE void compute_errors(const Ws<T>& w, int n, int k, int m, T mu, int tid) {
  if (tid < 64) {
    const T* r_d = w.res; const T* r_comp = w.res + n; const T* r_pe = w.res + n + m; const T* r_pi = w.res + n + m + k;
    T a = 0, b = 0, c1 = 0, c2 = 0, d = 0;
    for (int i = tid; i < n; i += 64) a += r_d[i] * r_d[i];
    for (int i = tid; i < k; i += 64) b += r_pe[i] * r_pe[i];
    for (int i = tid; i < m; i += 64) { c1 += r_comp[i] * r_comp[i]; c2 += r_comp[i]; d += r_pi[i] * r_pi[i]; }
    a = wave_sum(a); b = wave_sum(b); c1 = wave_sum(c1); c2 = wave_sum(c2); d = wave_sum(d);
    if (tid == 0) {
      w.red[0] = sqrtT(a);
      w.red[2] = k > 0 ? sqrtT(b) : (T)0;
      if (m > 0) {
        const T corrected = c1 - 2 * (c2 * mu) + (mu * mu) * (T)m;                // :432
        w.red[1] = sqrtT(corrected > (T)0 ? corrected : (T)0);
        w.red[3] = sqrtT(d);
      } else {
        w.red[1] = 0; w.red[3] = 0;
      }
    }
  }
  __syncthreads();
}

// ComputeMu, qp.cc:509-516 -> w.red[4]
template <typename T>
__device__ MO_INLINE void compute_mu(const Ws<T>& w, int n, int k, int m, int tid) {
  if (tid < 64) {
    const T* s = w.vars + n; const T* z = w.vars + n + m + k;
    T a = 0;
    for (int i = tid; i < m; i += 64) a += s[i] * z[i];
    a = wave_sum(a);
    if (tid == 0) w.red[4] = m > 0 ? a / (T)m : (T)0;
  }
  __syncthreads();
}

// Right-looking LDL^T (natural order, Eigen's zero-pivot rules: a zero pivot is tolerated iff the column below it is exactly zero, a
// non-zero pivot after a zero one is a failure) with the lower triangle of H held in the REGISTERS of the workgroup: a TG x TG thread
// grid, element (i, j) with thread (i mod TG, j mod TG) at local block (i / TG, j / TG) -- R (R + 1) / 2 values per thread, R = ceil(P /
// TG).  Per pivot the 16 (8) threads that own column k publish it to LDS (entries at and above the diagonal as zeros), one barrier, and
// every thread applies the rank-1 update to ALL its registers with 2 R broadcast reads: rows / columns that are already finished meet
// the zeros and stay.  The LDS-resident loop this replaces read two operands and wrote one result through LDS for every FMA.
// Afterwards H holds W = L D below the diagonal and invd = 1 / D, as the triangular solves expect.
template <typename T, int TG, int R>
__device__ MO_INLINE int factor_in_registers(const Ws<T>& w, int P, int tid) {
  const int ti = tid & (TG - 1), tj = tid / TG;
  T h[R * (R + 1) / 2];
#pragma unroll
  for (int bi = 0; bi < R; ++bi)
#pragma unroll
    for (int bj = 0; bj <= bi; ++bj) {
      const int i = ti + TG * bi, j = tj + TG * bj;
      h[bi * (bi + 1) / 2 + bj] = (i < P && j <= i) ? w.H[i + (size_t)j * w.ldh] : (T)0;
    }
  // two column buffers (alternating: one barrier per pivot): rhs (P) and delta (V >= P) are both rewritten by the solve that follows
  T* const colbuf[2] = {w.rhs, w.delta};
  bool found_zero = false;
  int status = MO_STATUS_OK;
  for (int kk = 0; kk < P; ++kk) {
    T* const col = colbuf[kk & 1];
    const int bk = kk / TG, rk = kk - bk * TG;
    if (tj == rk) {  // the owners of column kk publish it: col[i] = H(i, kk) for i > kk, 0 above, the pivot itself at col[kk]
#pragma unroll
      for (int bi = 0; bi < R; ++bi) {
        const int i = ti + TG * bi;
        if (i < P) {
          T v = (T)0;
#pragma unroll
          for (int bj = 0; bj <= bi; ++bj)
            if (bj == bk) v = h[bi * (bi + 1) / 2 + bj];   // compile-time register index, runtime select
          col[i] = i >= kk ? v : (T)0;
          if (i == kk) {  // the owner of the pivot also publishes 1 / d: one reciprocal per pivot instead of one per thread
            T q = (T)1 / v;   // (a zero pivot gives inf here; the branch below never uses it)
            w.invd[kk] = q;
          }
        }
      }
    }
    __syncthreads();
    const T d = col[kk];
    const bool valid = absT(d) > (T)0;
    if (!valid) {  // uniform: zero (or NaN) pivot
      bool nz = false;
      for (int i = kk + 1 + tid; i < P; i += kThreads) nz |= !(col[i] == (T)0);
      if (nz || !(d == (T)0)) w.iflag[1] = 1;
      __syncthreads();
      if (tid == 0) w.invd[kk] = (T)0;
      if (w.iflag[1]) { status = MO_STATUS_FACTORIZATION_FAILED; break; }
      found_zero = true;
      continue;
    }
    if (found_zero) { status = MO_STATUS_FACTORIZATION_FAILED; break; }  // non-zero pivot after a zero pivot
    const T inv = w.invd[kk];
    T ci[R], wj[R];
#pragma unroll
    for (int b = 0; b < R; ++b) {
      const int i = ti + TG * b, j = tj + TG * b;
      ci[b] = (i < P && i > kk) ? col[i] : (T)0;
      wj[b] = (j < P && j > kk) ? col[j] * inv : (T)0;
    }
#pragma unroll
    for (int bi = 0; bi < R; ++bi)
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj) h[bi * (bi + 1) / 2 + bj] -= ci[bi] * wj[bj];
  }
  __syncthreads();
  if (status != MO_STATUS_OK) return status;
#pragma unroll
  for (int bi = 0; bi < R; ++bi)
#pragma unroll
    for (int bj = 0; bj <= bi; ++bj) {
      const int i = ti + TG * bi, j = tj + TG * bj;
      if (i < P && j <= i) w.H[i + (size_t)j * w.ldh] = h[bi * (bi + 1) / 2 + bj];
    }
  __syncthreads();
  return MO_STATUS_OK;
}

// One 16-column half panel of the blocked factorisation with the panel in REGISTERS: thread tid owns rows tid + 256 e (e < RPT) and keeps
// their wd <= NBR values in registers; the panel's current values -- H minus the update from every column in front of it -- were staged in
// the LDS panel by left_stage, and H itself is only WRITTEN here, once, with the finished W.
// Round 4: NO barrier per pivot, and nothing but arithmetic in the pivot loop.  Every wave also keeps the 16 x 16 diagonal block in lanes
// 0 .. 15 (t[]: lane = row) and factorises it redundantly; the pivot and the multipliers f = W[jj][j] / D[j] then come out of its own
// registers by v_readlane -- scalars -- where the previous version published column j through LDS and waited for a workgroup barrier 16
// times per half panel.  Same arithmetic (one IEEE division per pivot, the same products in the same order).  The zero-pivot rules are
// NOT here: a zero (or NaN) pivot only raises a flag -- every wave sees the same pivots, so the flag needs no exchange -- and the function
// then writes nothing but that flag (w.iflag[4]); the caller repeats the panel with the LDS loop that knows Eigen's rules (left_factor_half).
// (The first cut had the rules in the loop: their branches cost a copy of the register panel at every join, then scratch reloads of spilled
// SGPRs behind vmcnt(0) in every pivot -- 3 k cycles per pivot, more than the barriers it had replaced.)
// Leaves W = L D in H and in the LDS panel (for the second half of the outer block), 1 / D in invd.
// A wave without a row of the panel (rows <= 64 x its index: the later panels of every system, most panels of a mid-size one) skips the
// loop altogether; the verdict travels through w.iflag[4] (wave 0 always owns rows) and is read behind the caller's barrier.
template <typename T, int NBR, int RPT>
__device__ inline void factor_panel_regs(const Ws<T>& w, T* panel, int kb, int wd, int rows, int ldp, int tid) {
  T a[RPT][NBR], t[NBR];
  tid = opaque(tid);
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool owns = 64 * wave < rows;   // (wave-uniform; with RPT = 2 the second row of a thread lies 256 further down)
#ifdef MO_GENERIC_STAMPS
  const unsigned long long fp_ta = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
  for (int e = 0; e < RPT; ++e) {
    const int i = tid + 256 * e;
#pragma unroll
    for (int jj = 0; jj < NBR; ++jj) a[e][jj] = (owns && i < rows && jj < wd && i >= jj) ? panel[i + (size_t)jj * ldp] : (T)0;
  }
  // (a ragged last panel, wd < NBR: the missing columns are identity columns -- pivot 1, multipliers 0 -- so that the pivot loop has no
  // condition on wd)
#pragma unroll
  for (int jj = 0; jj < NBR; ++jj) t[jj] = (owns && lane < wd && jj <= lane) ? panel[lane + (size_t)jj * ldp] : ((lane == jj && lane < NBR) ? (T)1 : (T)0);
  __syncthreads();   // every wave has its copy of the diagonal block before any wave overwrites the LDS panel with W
  if (!owns) return; // (wave-uniform)
#ifdef MO_GENERIC_STAMPS
  const unsigned long long fp_t0 = __builtin_amdgcn_s_memtime();
  if (tid == 0) atomicAdd(&g_nd_stamps[9], fp_t0 - fp_ta);
#endif
  bool bad = false;  // (scalar: d is)
  T invs = (T)0;
#pragma unroll
  for (int j = 0; j < NBR; ++j) {
    const T d = rl(t[j], j);                                                      // the pivot, as a scalar
    bad = bad || !(absT(d) > (T)0);
    const T inv = (T)1 / d;
    invs = lane == j ? inv : invs;                                                // lane j keeps 1 / D[j]: one masked store behind the loop
    const T tj = t[j];
#pragma unroll
    for (int jj = j + 1; jj < NBR; ++jj) {
      const T f = rl(tj, jj) * inv;
      t[jj] -= tj * f;
#pragma unroll
      for (int e = 0; e < RPT; ++e) {
        a[e][jj] -= a[e][j] * f;                                                   // (entries above the diagonal are never read)
        asm volatile("" : "+v"(a[e][jj]));   // (pinned: LLVM otherwise sinks the panel's updates behind the `bad` exit and keeps all 120
                                             // multipliers for them -- in scratch)
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // (one scheduling region per pivot: across all 16 the scheduler's order spills the panel)
  }
#ifdef MO_GENERIC_STAMPS
  const unsigned long long fp_t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) atomicAdd(&g_nd_stamps[7], fp_t1 - fp_t0);
  if (tid == 0 && uni((int)bad)) atomicAdd(&g_nd_stamps[8], 1ull);
#endif
  if (tid == 0) w.iflag[4] = bad ? 1 : 0;
  if (uni((int)bad)) return;
  if (tid < wd) w.invd[kb + tid] = invs;
#pragma unroll
  for (int e = 0; e < RPT; ++e) {
    const int i = tid + 256 * e;
    if (i < rows) {
#pragma unroll
      for (int jj = 0; jj < NBR; ++jj) {
        if (jj >= wd) continue;
        panel[i + (size_t)jj * ldp] = a[e][jj];
        if (i >= jj) w.H[(size_t)(kb + i) + (size_t)(kb + jj) * w.ldh] = a[e][jj];
      }
    }
  }
#ifdef MO_GENERIC_STAMPS
  if (tid == 0) atomicAdd(&g_nd_stamps[10], __builtin_amdgcn_s_memtime() - fp_t1);
#endif
}
// LARGE: LEFT-LOOKING blocked LDL^T with H in global memory (round 4; same natural order, same zero-pivot rules, same result layout: W = L D
// below the diagonal, invd = 1 / D).  Round 3 was right-looking -- after every 16-column panel the whole trailing matrix was read, updated and
// written back: P^3 / (6 NB) elements read AND written (4.3 MB per factorisation at P = 296 for 0.35 MB of H), the kernel's largest traffic
// term.  Left-looking, a block column is brought up to date only when its turn comes:
//   acc = W[panel rows, 0 .. kb) D^-1 W[kb .. kb + 32, 0 .. kb)^T      on the matrix cores, operands straight from the finished columns of H
//   panel = H[panel] - acc                                             staged in LDS, 16 columns at a time
//   the panel is factorised in registers (factor_panel_regs, thread = row) and its W is written to H -- the ONLY write of those elements.
// Every element of H is written once and read P / 32 times on average: 1.1 MB of reads + 0.35 MB of writes at P = 296.  The outer block is 32
// columns = two tile columns of accumulators per 16-row strip (a wave owns every fourth strip: 5 strips x 2 tiles x 8 registers at 320 rows);
// its second half also needs the first half's columns, which it takes from the LDS copy factor_panel_regs leaves behind.  Panels of more than
// 320 rows fall back to 16 columns and accumulate in chunks of 320 rows.  Tiles are computed transposed (D[jl][il], as the J^T J and the old
// trailing update did): the lanes of a 16-lane row hold 16 consecutive rows, so every operand load is a whole 128-byte segment of a column.
constexpr int kLeftSPW = 5;      // 16-row strips per wave and chunk: 4 waves x 5 strips = 320 rows
// acc[sl][c] += sum over the columns [c_begin, c_end) of SRC of  W[row0 + i][col] inv[col] W[rowc + 16 c + jl][col]   for strip sl of this wave
// (rows i = 16 (4 sl + wave) + l of the chunk that starts at panel row r0) and tile column c.  SRC: H (global, leading dimension w.ldh, absolute
// rows) or the LDS copy of the half panel just factorised (leading dimension ldp, rows relative to that panel).
template <typename T, int C16, bool FROM_PANEL>
__device__ inline void left_accumulate(const Ws<T>& w, const T* src, int ld, int row_base, int colrow_base, int c_begin, int c_end, int inv_base,
                                       int rows_avail, int r0, int c_first, int lane, int wave, typename Mfma16<T>::Acc (&acc)[kLeftSPW][C16]) {
  typedef Mfma16<T> MF;
  const int g = lane >> 4, l = lane & 15;
  int irow[kLeftSPW];
  bool have[kLeftSPW];
#pragma unroll
  for (int sl = 0; sl < kLeftSPW; ++sl) {
    const int i = r0 + 16 * (4 * sl + wave) + l;
    have[sl] = r0 + 16 * (4 * sl + wave) < rows_avail;           // wave-uniform: the strip exists
    irow[sl] = row_base + (i < rows_avail ? i : rows_avail - 1);  // clamped (in bounds): results of rows beyond the panel are never stored
  }
  int jrow[C16];
#pragma unroll
  for (int c = 0; c < C16; ++c) {
    const int j = 16 * (c_first + c) + l;
    jrow[c] = colrow_base + (j < rows_avail ? j : rows_avail - 1);
  }
  // Software pipeline (round 4): the columns come from global memory (HBM / MALL latency: the per-workgroup H does not stay in L2) and the
  // first version waited for every 4-column step's loads before its MFMAs.  Now the loads of 16 columns (4 steps: 28 loads a lane) are
  // issued together and the scale by 1 / D happens at consumption, not at the load: one round of latency per 40 MFMAs, which the other
  // workgroup's waves on the SIMD fill.  (Two batches in flight, ping-pong, cost 128 registers beside the 80 accumulators: 1 KB of spills.)
  constexpr int PF = 4;
  struct Batch { T aj[PF][C16], bi[PF][kLeftSPW], inv[PF]; };
  auto load = [&](int q0, Batch& b) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int col = q0 + 4 * s + g;
      const T* colp = src + (size_t)col * ld;
      b.inv[s] = w.invd[inv_base + col];
#pragma unroll
      for (int c = 0; c < C16; ++c) b.aj[s][c] = colp[jrow[c]];
#pragma unroll
      for (int sl = 0; sl < kLeftSPW; ++sl) b.bi[s][sl] = have[sl] ? colp[irow[sl]] : (T)0;
    }
  };
  auto consume = [&](const Batch& b) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      T a[C16];
#pragma unroll
      for (int c = 0; c < C16; ++c) a[c] = b.aj[s][c] * b.inv[s];
#pragma unroll
      for (int sl = 0; sl < kLeftSPW; ++sl) {
        if (!have[sl]) continue;
#pragma unroll
        for (int c = 0; c < C16; ++c) acc[sl][c] = MF::mac(a[c], b.bi[s][sl], acc[sl][c]);
      }
    }
  };
  const int nfull = (c_end - c_begin) / (4 * PF);
  int q = c_begin;
  for (int it = 0; it < nfull; ++it, q += 4 * PF) {
    Batch b;
    load(q, b);
    consume(b);
  }
  for (; q < c_end; q += 4) {   // (a ragged tail only where the blocks are 8 columns wide: very large systems)
    const bool valid = q + g < c_end;
    const int col = valid ? q + g : c_begin;
    const T inv = valid ? w.invd[inv_base + col] : (T)0;
    const T* colp = src + (size_t)col * ld;
    T aj[C16], bi[kLeftSPW];
#pragma unroll
    for (int c = 0; c < C16; ++c) aj[c] = valid ? colp[jrow[c]] * inv : (T)0;
#pragma unroll
    for (int sl = 0; sl < kLeftSPW; ++sl) bi[sl] = (valid && have[sl]) ? colp[irow[sl]] : (T)0;
#pragma unroll
    for (int sl = 0; sl < kLeftSPW; ++sl) {
      if (!have[sl]) continue;
#pragma unroll
      for (int c = 0; c < C16; ++c) acc[sl][c] = MF::mac(aj[c], bi[sl], acc[sl][c]);
    }
  }
  (void)FROM_PANEL;
}
// acc = -H[kb + i][kb + 16 c + j] for the rows of this chunk: the block's own values enter the accumulators BEFORE the update products are
// added (their loads are in flight during left_accumulate instead of being waited for between the accumulation and the staging).
template <typename T, int C16>
__device__ inline void left_init(const Ws<T>& w, int kb, int rows, int wd, int r0, int lane, int wave, typename Mfma16<T>::Acc (&acc)[kLeftSPW][C16]) {
  typedef Mfma16<T> MF;
  const int g = lane >> 4, l = lane & 15;
#pragma unroll
  for (int sl = 0; sl < kLeftSPW; ++sl) {
    const int i = r0 + 16 * (4 * sl + wave) + l;                // panel row
    const bool strip = r0 + 16 * (4 * sl + wave) < rows;        // wave-uniform
    const int ic = i < rows ? i : rows - 1;
#pragma unroll
    for (int c = 0; c < C16; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = 16 * c + MF::row(g, t);                   // panel column
        const int jc = j < wd ? j : wd - 1;
        T v = (T)0;
        if (strip) v = w.H[(size_t)(kb + ic) + (size_t)(kb + jc) * w.ldh];   // (clamped: in bounds; a scalar branch around the loads)
        acc[sl][c][t] = (i < rows && j < wd && j <= i) ? -v : (T)0;
      }
  }
}
// Stage -acc = H[kb + i][kb + 16 c + j] - (update) for the rows of this chunk and tile column c into the LDS panel of the half that starts at
// panel column 16 c (rows relative to that half: i - 16 c), zeros above the diagonal.
template <typename T, int C16>
__device__ inline void left_stage(const Ws<T>& w, T* panel, int kb, int rows, int wd, int c, int r0, int lane, int wave,
                                  const typename Mfma16<T>::Acc (&acc)[kLeftSPW][C16]) {
  typedef Mfma16<T> MF;
  const int g = lane >> 4, l = lane & 15;
  const int ldp = (rows - 16 * c) | 1;
#pragma unroll
  for (int sl = 0; sl < kLeftSPW; ++sl) {
    const int i = r0 + 16 * (4 * sl + wave) + l;                // panel row
    if (r0 + 16 * (4 * sl + wave) >= rows) continue;            // wave-uniform
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = 16 * c + MF::row(g, t);                     // panel column
      if (i < rows && i >= 16 * c && j < wd) {
        const T v = j <= i ? -acc[sl][c][t] : (T)0;
        panel[(i - 16 * c) + (size_t)(j - 16 * c) * ldp] = v;
      }
    }
  }
}
// One 16-column half panel whose current values sit in the LDS panel: factorise (registers where the rows allow it, LDS otherwise), write W to H.
template <typename T>
__device__ inline int left_factor_half(const Ws<T>& w, T* panel, int kb, int wd, int rows, int tid, bool& found_zero) {
  const int ldp = rows | 1;
  int status = MO_STATUS_OK;
  tid = opaque(tid);
  if (uni((int)kThreads) == 256 && rows <= 512 && !uni((int)found_zero)) {
    if (rows <= 256) factor_panel_regs<T, 16, 1>(w, panel, kb, wd, rows, ldp, tid);
    else factor_panel_regs<T, 16, 2>(w, panel, kb, wd, rows, ldp, tid);
    __syncthreads();                                                             // the LDS copy of W is read by other threads
    if (!uni(w.iflag[4])) return status;
    // (a zero or NaN pivot in this half panel: nothing was written; the loop below applies the rules)
  }
  for (int j = 0; j < wd; ++j) {                                                 // any size: the panel stays in LDS
    const T d = uni(panel[j + (size_t)j * ldp]);                                  // the pivot, as a scalar
    if (!(absT(d) > (T)0)) {                                                      // zero (or NaN) pivot: Eigen's rules, as in factor_in_registers
      bool nz = false;
      for (int i = j + 1 + tid; i < rows; i += kThreads) nz |= !(panel[i + (size_t)j * ldp] == (T)0);
      if (nz || !(d == (T)0)) w.iflag[1] = 1;
      if (tid == 0) w.invd[kb + j] = (T)0;                                        // published by the barrier below, together with the flag
      __syncthreads();
      if (uni(w.iflag[1])) { status = MO_STATUS_FACTORIZATION_FAILED; break; }
      found_zero = true;
      continue;
    }
    if (uni((int)found_zero)) { status = MO_STATUS_FACTORIZATION_FAILED; break; } // non-zero pivot after a zero pivot
    const T inv = (T)1 / d;
    if (tid == 0) w.invd[kb + j] = inv;
    const int rem = wd - j - 1;
    for (int idx = tid; idx < rem * rows; idx += kThreads) {                      // the panel's remaining columns
      const int c = idx / rows, i = idx - c * rows, jj = j + 1 + c;
      if (i >= jj) panel[i + (size_t)jj * ldp] -= panel[i + (size_t)j * ldp] * (panel[jj + (size_t)j * ldp] * inv);
    }
    __syncthreads();
  }
  if (status != MO_STATUS_OK) return status;                                       // uniform
  for (int idx = tid; idx < rows * wd; idx += kThreads) {                          // W to H: the one write of these elements
    const int jj = idx / rows, i = idx - jj * rows;
    if (i >= jj) w.H[(size_t)(kb + i) + (size_t)(kb + jj) * w.ldh] = panel[i + (size_t)jj * ldp];
  }
  __syncthreads();
  return status;
}
template <typename T>
__device__ MO_INLINE int factor_blocked(const Ws<T>& w, int P, int NB, int tid) {
  typedef Mfma16<T> MF;
  typedef typename MF::Acc Acc;
  // (the LDS panel region holds NB columns of P | 1 rows, NB = 32 / 16 / 8 by what fits beside the vectors: one 16-column half panel, or
  // NB = 8 columns of a very large system -- the tiles then carry eight idle columns)
  T* const panel = w.Jc;                     // the J chunk is dead by now
  bool found_zero = false;
  int status = MO_STATUS_OK;
  // (the LARGE kernel is launched with exactly four waves: launch_generic_large.  The wave index is made scalar so that the per-strip
  // conditions below are scalar branches, not EXEC-masked regions around the accumulator tiles.)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int kb = 0;
#ifdef MO_GENERIC_STAMPS
  unsigned long long fb_prev = __builtin_amdgcn_s_memtime();
#endif
  while (kb < P && status == MO_STATUS_OK) {
    const int rows = P - kb;
    tid = opaque(tid);
    const int lane = tid & 63;
    if (rows <= 64 * kLeftSPW) {
      // ---- a 32-column outer block in one chunk of rows: both halves' accumulators live in registers
      const int wd = rows < 32 ? rows : 32;
      Acc acc[kLeftSPW][2];
      left_init<T, 2>(w, kb, rows, wd, 0, lane, wave, acc);
      MO_FBSTAMP(2);
      left_accumulate<T, 2, false>(w, w.H, w.ldh, kb, kb, 0, kb, 0, rows, 0, 0, lane, wave, acc);
      MO_FBSTAMP(3);
      left_stage<T, 2>(w, panel, kb, rows, wd, 0, 0, lane, wave, acc);
      __syncthreads();
      MO_FBSTAMP(4);
      status = uni(left_factor_half<T>(w, panel, kb, wd < 16 ? wd : 16, rows, tid, found_zero));
      MO_FBSTAMP(5);
      if (status != MO_STATUS_OK) break;
      if (wd > 16) {
        // second half: the columns of the first half come from its LDS copy (rows relative to the first half's panel)
        {
          // tile column 1 only: aj from panel rows 16 .. 31, bi from the strips' rows; columns 0 .. 15 of the LDS panel, inv at kb + column
          Acc one[kLeftSPW][1];
#pragma unroll
          for (int sl = 0; sl < kLeftSPW; ++sl) one[sl][0] = acc[sl][1];
          left_accumulate<T, 1, true>(w, panel, rows | 1, 0, 0, 0, 16, kb, rows, 0, 1, lane, wave, one);
#pragma unroll
          for (int sl = 0; sl < kLeftSPW; ++sl) acc[sl][1] = one[sl][0];
        }
        __syncthreads();                                                            // every wave is done with the first half's LDS copy
        MO_FBSTAMP(6);
        left_stage<T, 2>(w, panel, kb, rows, wd, 1, 0, lane, wave, acc);
        __syncthreads();
        MO_FBSTAMP(4);
        status = uni(left_factor_half<T>(w, panel, kb + 16, wd - 16, rows - 16, tid, found_zero));
        MO_FBSTAMP(5);
      }
      kb += wd;
    } else {
      // ---- more than 320 rows: a 16-column block (8 where the LDS holds no more), the update accumulated and staged in chunks of 320 rows
      const int wd = NB < 16 ? NB : 16;
      for (int r0 = 0; r0 < rows; r0 += 64 * kLeftSPW) {
        Acc acc[kLeftSPW][1];
        left_init<T, 1>(w, kb, rows, wd, r0, lane, wave, acc);
        left_accumulate<T, 1, false>(w, w.H, w.ldh, kb, kb, 0, kb, 0, rows, r0, 0, lane, wave, acc);
        left_stage<T, 1>(w, panel, kb, rows, wd, 0, r0, lane, wave, acc);
      }
      __syncthreads();
      status = uni(left_factor_half<T>(w, panel, kb, wd, rows, tid, found_zero));
      kb += wd;
    }
    __threadfence_block();
    __syncthreads();   // the next block reads columns other threads wrote
  }
  __syncthreads();
  return status;
}
// LARGE: (L D L^T) sol = rhs in place with H in global memory, in blocks of SB <= 32 columns (one round of global-memory latency per block
// instead of one per column).  Forward: the block's SB x SB triangle is staged in LDS and solved by wave 0 with the unknowns in registers
// (lane = row, one readlane per column); then every row below gets its SB-term update from SB independent coalesced column reads.  Backward:
// the block's dot products with the part of the solution below it first (waves over columns, lanes over rows, SB independent reads per
// lane), then the transposed triangle on wave 0 (lane = column).  `region` = elements of the panel region (free between factorisations).
template <typename T>
__device__ MO_INLINE void block_solve(const Ws<T>& w, int P, int region, int tid) {
  int SB = 32;
  while (SB > 1 && SB * (SB + 1) + 2 * SB > region) SB >>= 1;
  const int ldb = SB + 1;
  T* const dblk = w.Jc;                 // the triangle, column-major, ld = SB + 1
  T* const tvec = dblk + SB * ldb;      // SB scaled unknowns / SB partial dot products
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = kThreads >> 6;
  for (int c0 = 0; c0 < P; c0 += SB) {
    const int wd = P - c0 < SB ? P - c0 : SB;
    tid = opaque(tid);
    const int lane = tid & 63;
    if (wd == 32) {
      // A whole 32-column block (round 4): nothing staged, nothing waited for twice.  Every thread first ISSUES the 32 loads of its row
      // below the block; wave 0 takes its rows of the triangle straight from global memory (lane = row: coalesced), solves them with the
      // unknowns in registers and compile-time lane indices, and publishes the scaled unknowns; after ONE barrier the rows below -- their
      // operands long since arrived -- take their 32-term update.  Same products, same order as the staged version below.
      const int i0 = c0 + 32 + tid;
      T hv[32];
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) hv[jj] = (T)0;
      if (i0 < P) {
        const T* row = w.H + (size_t)i0 + (size_t)c0 * w.ldh;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) hv[jj] = row[(size_t)jj * w.ldh];
      }
      if (wave == 0) {
        const int lr = lane < 32 ? lane : 31;
        const T* trow = w.H + (size_t)(c0 + lr) + (size_t)c0 * w.ldh;
        T tri[32];
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) tri[jj] = trow[(size_t)jj * w.ldh];     // (on and above the diagonal: read, never used)
        const T invl = w.invd[c0 + lr];
        T zi = w.rhs[c0 + lr];
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
          const T tj = rl(zi, jj) * rl(invl, jj);
          if (lane > jj && lane < 32) zi -= tri[jj] * tj;
        }
        if (lane < 32) { w.rhs[c0 + lane] = zi; tvec[lane] = zi * invl; }
      }
      __syncthreads();
      if (i0 < P) {
        T acc = (T)0;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) acc += hv[jj] * tvec[jj];
        w.rhs[i0] -= acc;
      }
      for (int i = i0 + kThreads; i < P; i += kThreads) {       // (more than 256 rows below the block)
        const T* row = w.H + (size_t)i + (size_t)c0 * w.ldh;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) hv[jj] = row[(size_t)jj * w.ldh];
        T acc = (T)0;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) acc += hv[jj] * tvec[jj];
        w.rhs[i] -= acc;
      }
      __syncthreads();
      continue;
    }
#pragma unroll 4
    for (int idx = tid; idx < wd * wd; idx += kThreads) {
      const int jj = idx / wd, i = idx - jj * wd;
      dblk[i + jj * ldb] = i > jj ? w.H[(size_t)(c0 + i) + (size_t)(c0 + jj) * w.ldh] : (T)0;
    }
    __syncthreads();
    if (tid < 64) {
      T zi = lane < wd ? w.rhs[c0 + lane] : (T)0;
      for (int jj = 0; jj < wd; ++jj) {
        const T tj = rl(zi, jj) * w.invd[c0 + jj];
        if (lane > jj && lane < wd) zi -= dblk[lane + jj * ldb] * tj;
      }
      if (lane < wd) { w.rhs[c0 + lane] = zi; tvec[lane] = zi * w.invd[c0 + lane]; }
    }
    __syncthreads();
    for (int i = c0 + wd + tid; i < P; i += kThreads) {
      const T* row = w.H + (size_t)i + (size_t)c0 * w.ldh;
      T acc = (T)0;
      if (wd == 32) {                 // a whole block: its 32 loads are in flight together (one round of latency, not four)
        T hv[32];
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) hv[jj] = row[(size_t)jj * w.ldh];
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) acc += hv[jj] * tvec[jj];
      } else {
#pragma unroll 8
        for (int jj = 0; jj < wd; ++jj) acc += row[(size_t)jj * w.ldh] * tvec[jj];
      }
      w.rhs[i] -= acc;
    }
    __syncthreads();
  }
  const int last = ((P - 1) / SB) * SB;
  for (int c0 = last; c0 >= 0; c0 -= SB) {
    const int wd = P - c0 < SB ? P - c0 : SB, below = c0 + wd;
    const bool whole = wd == 32;
    tid = opaque(tid);
    const int lane = tid & 63;
    T colv[32];                                  // whole block: lane = column of the triangle, its entries below the diagonal in registers
    if (whole) {
      if (wave == 0) {
        const int lc = lane < 32 ? lane : 31;
        const T* cp = w.H + (size_t)c0 + (size_t)(c0 + lc) * w.ldh;
#pragma unroll
        for (int i = 0; i < 32; ++i) colv[i] = cp[i];                            // (on and above the diagonal: read, never used)
      }
    } else {
#pragma unroll 4
      for (int idx = tid; idx < wd * wd; idx += kThreads) {
        const int jj = idx / wd, i = idx - jj * wd;
        dblk[i + jj * ldb] = i > jj ? w.H[(size_t)(c0 + i) + (size_t)(c0 + jj) * w.ldh] : (T)0;
      }
    }
    // the dot products with the solution below the block: a wave takes every nwaves-th column, four of its columns at a time with the
    // loads of four 64-row chunks each in flight together (16 a lane; one column at a time was a round of latency per column).  Per
    // column the sum runs over the chunks in order, as before.
    for (int jb = wave; jb < wd; jb += 4 * nwaves) {
      T sacc[4] = {(T)0, (T)0, (T)0, (T)0};
      for (int ib = below; ib < P; ib += 4 * 64) {
        T hv[4][4], xv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int i = ib + lane + 64 * t;
          xv[t] = i < P ? w.rhs[i] : (T)0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int jj = jb + c * nwaves;
            hv[c][t] = (T)0;
            if (jj < wd && ib + 64 * t < P)                       // (scalar)
              hv[c][t] = w.H[(size_t)(i < P ? i : P - 1) + (size_t)(c0 + jj) * w.ldh];
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (ib + 64 * t < P) sacc[c] += (ib + lane + 64 * t < P ? hv[c][t] : (T)0) * xv[t];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int jj = jb + c * nwaves;
        if (jj >= wd) continue;                                   // (scalar)
        const T tot = wave_sum(sacc[c]);
        if (lane == 0) tvec[jj] = tot;
      }
    }
    __syncthreads();
    if (whole) {
      if (wave == 0) {                           // the transposed triangle on wave 0, compile-time lane indices
        const int lc = lane < 32 ? lane : 31;
        const T invl = w.invd[c0 + lc];
        T v = w.rhs[c0 + lc] - tvec[lc], mine = (T)0;
#pragma unroll
        for (int i = 31; i >= 0; --i) {
          const T xi = rl(v, i) * rl(invl, i);
          if (lane == i) mine = xi;
          if (lane < i) v -= colv[i] * xi;
        }
        if (lane < 32) w.rhs[c0 + lane] = mine;
      }
    } else if (tid < 64) {
      T v = lane < wd ? w.rhs[c0 + lane] - tvec[lane] : (T)0, mine = (T)0;
      for (int i = wd - 1; i >= 0; --i) {
        const T xi = rl(v, i) * w.invd[c0 + i];
        if (lane == i) mine = xi;
        if (lane < i) v -= dblk[i + lane * ldb] * xi;
      }
      if (lane < wd) w.rhs[c0 + lane] = mine;
    }
    __syncthreads();
  }
}

// Reduced-KKT assembly (Sigma on the diagonal, qp.cc:293-298) + LDL^T.  Returns MO_STATUS_*.
template <typename T, int TG, int R, bool LARGE = false>
__device__ MO_INLINE_FACTOR int assemble_and_factor(MO_FACTOR_WS(T) w, int n, int k, int m, bool include_ineq, int tid, int m_r = 0) {
  const int P = n + k;
  if (include_ineq) {
    const T* s = w.vars + n; const T* z = w.vars + n + m + k;
    for (int c = tid; c < m; c += kThreads)
      if (!(s[c] > (T)0)) w.iflag[0] = 1;                                        // F_ASSERT qp.cc:285
    __syncthreads();
    if (w.iflag[0]) return MO_STATUS_NONPOSITIVE_SLACK;
    for (int v = tid; v < n; v += kThreads) {
      T hv = w.H[v + (size_t)v * w.ldh];
      for (int c = 0; c < m; ++c)
        if (w.cv[c] == v) hv += w.ca[c] * (z[c] / s[c]) * w.ca[c];               // :296, accumulates in constraint order
      w.H[v + (size_t)v * w.ldh] = hv;
    }
    __syncthreads();
  }
  // ---- LDL^T with the matrix distributed over the workgroup's REGISTERS (the kernel is instantiated per thread grid TG and block count R)
  if constexpr (LARGE) return factor_blocked<T>(w, P, panel_cols_for(n, k, m, m_r, (int)sizeof(T)), tid);
  else return factor_in_registers<T, TG, R>(w, P, tid);
}

// Solve (L D L^T) sol = rhs in place, wave 0 only; H holds W = L D below the diagonal, invd = 1/D.
template <typename T>
__device__ MO_INLINE void wave_solve(const Ws<T>& w, int P, int lane) {
  const int r0 = lane, r1 = lane + 64, r2 = lane + 128;
  T b0 = r0 < P ? w.rhs[r0] : (T)0, b1 = r1 < P ? w.rhs[r1] : (T)0, b2 = r2 < P ? w.rhs[r2] : (T)0;
  for (int kk = 0; kk < P; ++kk) {  // forward: t = L^-1 b
    const T src = kk < 64 ? b0 : (kk < 128 ? b1 : b2);
    const T tk = rl(src, kk & 63) * w.invd[kk];
    const T* col = w.H + (size_t)kk * w.ldh;
    if (r0 > kk && r0 < P) b0 -= col[r0] * tk;
    if (r1 > kk && r1 < P) b1 -= col[r1] * tk;
    if (r2 > kk && r2 < P) b2 -= col[r2] * tk;
  }
  for (int kk = P - 1; kk >= 0; --kk) {  // backward: x_k = invd_k (t_k - sum_{i>k} w_ik x_i), column oriented
    const T src = kk < 64 ? b0 : (kk < 128 ? b1 : b2);
    const T xk = rl(src, kk & 63) * w.invd[kk];
    const T* row = w.H + kk;
    if (r0 < kk) b0 -= row[(size_t)r0 * w.ldh] * xk;
    if (r1 < kk) b1 -= row[(size_t)r1 * w.ldh] * xk;
    if (r2 < kk) b2 -= row[(size_t)r2 * w.ldh] * xk;
  }
  if (r0 < P) w.rhs[r0] = b0 * w.invd[r0];
  if (r1 < P) w.rhs[r1] = b1 * w.invd[r1];
  if (r2 < P) w.rhs[r2] = b2 * w.invd[r2];
}

// SolveForUpdate (qp.cc:318-364) / SolveForUpdateNoInequalities (qp.cc:366-386) with the factorisation in H.
template <typename T, bool LARGE = false>
__device__ MO_INLINE void solve_for_update(const Ws<T>& w, int n, int k, int m, T mu, bool include_ineq, int tid) {
  const int P = n + k;
  const T* s = w.vars + n; const T* z = w.vars + n + m + k;
  const T* r_d = w.res; const T* r_comp = w.res + n; const T* r_pe = w.res + n + m; const T* r_pi = w.res + n + m + k;
  const T* ds_aff = w.daff + n; const T* dz_aff = w.daff + n + m + k;
  for (int v = tid; v < n; v += kThreads) {
    T ra = r_d[v];                                                                // :337
    if (include_ineq) {
      for (int c = 0; c < m; ++c) {
        if (w.cv[c] == v) {                                                       // :340-341
          ra += w.ca[c] * (z[c] / s[c]) * r_pi[c];
          ra += w.ca[c] * (r_comp[c] + (ds_aff[c] * dz_aff[c]) - mu) / s[c];
        }
      }
    }
    w.rhs[v] = -ra;
  }
  for (int q = tid; q < k; q += kThreads) w.rhs[n + q] = -r_pe[q];
  __syncthreads();
  if constexpr (LARGE) {
    block_solve(w, P, w.region, tid);
  } else {
    if (tid < 64) wave_solve(w, P, tid);
    __syncthreads();
  }
  T* dx = w.delta; T* ds = w.delta + n; T* dy = w.delta + n + m; T* dz = w.delta + n + m + k;
  for (int i = tid; i < n; i += kThreads) dx[i] = w.rhs[i];
  for (int q = tid; q < k; q += kThreads) dy[q] = -w.rhs[n + q];                   // py is negated, :353
  __syncthreads();
  for (int c = tid; c < m; c += kThreads) {
    if (include_ineq) {                                                           // :359-363
      const T dsv = w.ca[c] * dx[w.cv[c]] + r_pi[c];
      ds[c] = dsv;
      dz[c] = -(z[c] / s[c]) * dsv - ((T)1 / s[c]) * (r_comp[c] + (ds_aff[c] * dz_aff[c]) - mu);
    } else {
      ds[c] = (T)0; dz[c] = (T)0;
    }
  }
  __syncthreads();
}

// ComputeAlpha, qp.cc:485-507 -> w.red[5] (primal), w.red[6] (dual)
template <typename T>
__device__ MO_INLINE void compute_alpha(const Ws<T>& w, int n, int k, int m, T tau, int tid) {
  if (tid < 64) {
    const T* s = w.vars + n; const T* z = w.vars + n + m + k;
    const T* ds = w.delta + n; const T* dz = w.delta + n + m + k;
    T ap = 1, ad = 1;
    for (int i = tid; i < m; i += 64) {
      if (s[i] + ds[i] <= (T)0 && absT(ds[i]) > (T)0) { const T cnd = -tau * s[i] / ds[i]; ap = cnd < ap ? cnd : ap; }
      if (z[i] + dz[i] <= (T)0 && absT(dz[i]) > (T)0) { const T cnd = -tau * z[i] / dz[i]; ad = cnd < ad ? cnd : ad; }
    }
    ap = wave_min(ap); ad = wave_min(ad);
    if (tid == 0) { w.red[5] = ap; w.red[6] = ad; }
  }
  __syncthreads();
}

// ComputePredictorCorrectorMuAffine, qp.cc:519-537 -> w.red[7]
template <typename T>
__device__ MO_INLINE void compute_mu_affine(const Ws<T>& w, int n, int k, int m, T mu, T ap, T ad, int tid) {
  if (tid < 64) {
    const T* s = w.vars + n; const T* z = w.vars + n + m + k;
    const T* ds = w.daff + n; const T* dz = w.daff + n + m + k;
    T a = 0, b = 0, c = 0;
    for (int i = tid; i < m; i += 64) { a += s[i] * dz[i]; b += z[i] * ds[i]; c += ds[i] * dz[i]; }
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if (tid == 0) {
      T ma = mu;
      ma += ad * a / (T)m;
      ma += ap * b / (T)m;
      ma += (ad * ap) * c / (T)m;
      w.red[7] = ma > (T)0 ? ma : (T)0;
    }
  }
  __syncthreads();
}

// Iterate's solve part (qp.cc:163-193), state update excluded.  Residual must be current, H = G + A.
// ip[6] receives IPIterationOutputs.  Returns MO_STATUS_*.
template <typename T, int TG, int R, bool LARGE = false>
__device__ MO_INLINE int newton_direction(const Ws<T>& w, int n, int k, int m, T mu_input, int strategy, T tau, T* ip, int tid, int m_r = 0) {
  const int V = n + 2 * m + k;
  ip[0] = mu_input; ip[1] = 1; ip[2] = 1; ip[3] = nanT<T>(); ip[4] = nanT<T>(); ip[5] = nanT<T>();
  for (int i = tid; i < V; i += kThreads) w.daff[i] = (T)0;                        // delta_affine_.setZero(), :315
#ifdef MO_GENERIC_STAMPS
  unsigned long long nd_prev = __builtin_amdgcn_s_memtime();
#endif
  const int st = assemble_and_factor<T, TG, R, LARGE>(w, n, k, m, true, tid, m_r);
#ifdef MO_GENERIC_PROBE_FENCE
  __threadfence_block();
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
#endif
  MO_NDSTAMP(0);
  if (st != MO_STATUS_OK) return st;
  if (m == 0) {
    solve_for_update<T, LARGE>(w, n, k, m, (T)0, true, tid);                                // :165-167
  } else if (strategy != MO_PREDICTOR_CORRECTOR) {
    solve_for_update<T, LARGE>(w, n, k, m, mu_input, true, tid);                            // :169
    MO_NDSTAMP(1);
  } else {
    solve_for_update<T, LARGE>(w, n, k, m, (T)0, true, tid);                                // :173
    compute_alpha(w, n, k, m, (T)1, tid);                                         // :174
    ip[3] = w.red[5]; ip[4] = w.red[6];
    for (int i = tid; i < V; i += kThreads) w.daff[i] = w.delta[i];               // :177
    __syncthreads();
    compute_mu_affine(w, n, k, m, mu_input, ip[3], ip[4], tid);                   // :180
    ip[5] = w.red[7];
    const T ratio = ip[5] / mu_input;
    const T sigma = ratio * ratio * ratio;                                        // :182
    ip[0] = sigma * mu_input;                                                     // :183
    solve_for_update<T, LARGE>(w, n, k, m, ip[0], true, tid);                               // :187
  }
  if (m > 0) {                                                                    // :191-193
    compute_alpha(w, n, k, m, tau, tid);
    ip[1] = w.red[5]; ip[2] = w.red[6];
#ifdef MO_GENERIC_PROBE
    if (strategy != MO_PREDICTOR_CORRECTOR) { ip[3] = w.vars[n + m + k]; ip[4] = w.delta[n + m + k]; ip[5] = w.red[6]; }
#endif
  }
  return MO_STATUS_OK;
}

// x,s += alpha_p (dx,ds); y,z += alpha_d (dy,dz), qp.cc:196-199
template <typename T>
__device__ MO_INLINE void update_state(const Ws<T>& w, int n, int k, int m, T ap, T ad, int tid) {
  const int V = n + 2 * m + k;
  for (int i = tid; i < V; i += kThreads) {
    const bool primal = i < n + m;
    w.vars[i] += w.delta[i] * (primal ? ap : ad);
  }
  __syncthreads();
}

// ---- the kernel ------------------------------------------------------------------------------------------------
// TG x TG = the thread grid of the register-distributed factorisation (8 x 8: single-wave workgroups, 16 x 16: 256 threads),
// R = ceil((n + k) / TG) blocks per thread and dimension (also the block count of the J^T J register tiling: n <= TG R).
// LARGE: H in the workgroup's global workspace (a.H_work), left-looking blocked factorisation, 128-wide J^T J super-blocks: any n + k the LDS
// vectors allow, and every system of at least MO_LARGE_MIN_P.
template <typename T, int MODE, int TG, int R, bool LARGE = false>
__global__ __launch_bounds__(TG * TG, LARGE ? 2 : 1) void kkt_generic_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ws<T> w;
  int n = a.n, k = a.k, m = a.m, m_r = a.m_r;
  if constexpr (LARGE) carve_large(w, smem, (T*)a.H_work + (size_t)blockIdx.x * (size_t)a.H_work_stride, n, k, m, m_r);
  else carve(w, smem, n, k, m, m_r);
  const int tid0 = threadIdx.x;
  const bool j_level = a.J != nullptr;

#ifdef MO_GENERIC_STAMPS
  unsigned long long gstamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gstamp_prev = __builtin_amdgcn_s_memtime();
#endif
  for (long long p = blockIdx.x; p < a.batch; p += gridDim.x) {
    if (MODE == MODE_SOLVE && a.skip && a.skip[p * a.skip_stride] >= 0) continue;  // uniform: finished in the caller's outer loop
    __syncthreads();  // previous problem's readers are done with LDS
    // The sizes are the same for every problem, and LLVM knows it: it hoisted every scalar that depends only on them -- the tile masks of
    // each J^T J block, panel widths, loop bounds, hundreds of them -- and every predicate of the thread index out of this loop, where they
    // stayed live across all phases: 1 100 SGPR spills to VGPR lanes in the LARGE kernels, and the VGPRs holding those lanes spilled to
    // scratch in turn (each reload a vmcnt(0) wait behind the other workgroup's memory traffic).  Opaque copies per problem: the scalars are
    // recomputed where they are used (the LARGE phases take a fresh copy of the thread index themselves).
    asm volatile("" : "+s"(n), "+s"(k), "+s"(m), "+s"(m_r));
    const int tid = opaque(tid0);
    const int V = n + 2 * m + k;
    const T* Jp = j_level ? (const T*)a.J + p * a.J_stride : nullptr;
    const T* rp = j_level ? (const T*)a.r + p * a.r_stride : nullptr;
    const T* Gp = a.G ? (const T*)a.G + p * a.G_stride : nullptr;
    const T* cp = a.c ? (const T*)a.c + p * a.c_stride : nullptr;
    const T* Ap = k > 0 ? (const T*)a.A + p * a.A_stride : nullptr;
    const T* bp = k > 0 ? (const T*)a.b + p * a.b_stride : nullptr;
    int G_ld = a.G_ld;

    // constraints + state into LDS
    if (tid < 8) w.iflag[tid] = 0;
    __syncthreads();
    if (m > 0 && MODE != MODE_LINEARIZE) {
      const int* cvp = a.cons_var + p * a.cons_stride;
      const T* cap = (const T*)a.cons_a + p * a.cons_stride;
      const T* cbp = (const T*)a.cons_b + p * a.cons_stride;
      for (int c = tid; c < m; c += kThreads) {
        const int v = cvp[c];
        if (v < 0 || v >= n) { w.iflag[2] = 1; w.cv[c] = 0; } else { w.cv[c] = v; }
        w.ca[c] = cap[c]; w.cb[c] = cbp[c];
      }
    }
    if (a.vars && MODE != MODE_LINEARIZE) {
      const T* vp = (const T*)a.vars + p * a.vars_stride;
      for (int i = tid; i < V; i += kThreads) w.vars[i] = vp[i];
    }
    T mu_p = (T)0;
    if (a.mu) mu_p = ((const T*)a.mu)[p * a.mu_stride];

    // cost: QP-level (G, c) or J-level (J, r, lambda)
    MO_GSTAMP(0);
    load_qp(w, n, k, j_level ? (const T*)nullptr : Gp, G_ld, cp, Ap, a.A_ld, bp, tid);
    MO_GSTAMP(1);
    if (j_level) {
      const T lam = a.lambda_vec ? ((const T*)a.lambda_vec)[p * a.lambda_vec_stride] : (T)a.lambda;  // per-problem LM state
      accumulate_jtj<T, TG, R, LARGE>(w, n, m_r, Jp, a.J_ld, a.J_row_major, rp, lam, tid);
      MO_GSTAMP(2);
      if (MODE == MODE_LINEARIZE || MODE == MODE_SOLVE) {
        // LINEARIZE output, or the per-problem G scratch the Solve loop reloads after each factorisation
        T* Go = (T*)a.G_out + p * a.G_out_stride;
        for (int j0 = 0; j0 < n; j0 += 8)                      // (eight loads, then eight stores: see load_qp)
          for (int i = tid; i < n; i += kThreads) {
            T v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (j0 + e < n && i >= j0 + e) ? w.H[i + (size_t)(j0 + e) * w.ldh] : (T)0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (j0 + e < n) Go[i + (size_t)(j0 + e) * a.G_out_ld] = v[e];
          }
        T* co = (T*)a.c_out + p * a.c_out_stride;
        for (int i = tid; i < n; i += kThreads) co[i] = w.cvec[i];
        if (a.half_sq_out && tid == 0) ((T*)a.half_sq_out)[p * (a.half_sq_stride ? a.half_sq_stride : 1)] = w.red[8];
        Gp = Go; cp = co; G_ld = a.G_out_ld;
        __threadfence();  // the Solve loop re-reads this scratch from other threads of the workgroup
        __syncthreads();
      }
    }
    if (MODE == MODE_LINEARIZE) continue;

    int st = w.iflag[2] ? MO_STATUS_BAD_INDEX : MO_STATUS_OK;  // uniform: iflag written before the barriers above

    if (MODE == MODE_RESIDUAL) {
      const bool inc = !(a.flags & MO_STEP_NO_INEQUALITIES);
      eval_kkt<LARGE>(w, n, k, m, inc, tid);
      compute_errors(w, n, k, m, mu_p, tid);
      T* ro = (T*)a.r_out + p * a.r_out_stride;
      for (int i = tid; i < V; i += kThreads) ro[i] = w.res[i];
      if (a.kkt_out && tid < 4) ((T*)a.kkt_out)[p * 4 + tid] = w.red[tid];  // LDS, runtime index is fine
      if (a.status && tid == 0) a.status[p] = st;
      continue;
    }

    if (MODE == MODE_STEP || MODE == MODE_ITERATE) {
      T ip[6];
      const bool no_ineq = (a.flags & MO_STEP_NO_INEQUALITIES) != 0;
      for (int i = tid; i < V; i += kThreads) { w.delta[i] = (T)0; w.daff[i] = (T)0; }
      if (st == MO_STATUS_OK) {
        eval_kkt<LARGE>(w, n, k, m, !no_ineq, tid);
        MO_GSTAMP(3);
        if (no_ineq) {
          st = assemble_and_factor<T, TG, R, LARGE>(w, n, k, m, false, tid, m_r);
          if (st == MO_STATUS_OK) solve_for_update<T, LARGE>(w, n, k, m, (T)0, false, tid);
          ip[0] = mu_p; ip[1] = 1; ip[2] = 1; ip[3] = ip[4] = ip[5] = nanT<T>();
        } else {
          const int strat = (MODE == MODE_ITERATE) ? a.barrier_strategy : MO_COMPLEMENTARITY;
          st = newton_direction<T, TG, R, LARGE>(w, n, k, m, mu_p, strat, (T)a.tau, ip, tid, m_r);
        }
      }
      MO_GSTAMP(4);
      if (st == MO_STATUS_OK) {  // non-finite direction?
        bool bad = false;
        for (int i = tid; i < V; i += kThreads) bad |= !finiteT(w.delta[i]);
        if (bad) w.iflag[3] = 1;
        __syncthreads();
        if (w.iflag[3]) st = MO_STATUS_NONFINITE;
      }
      if (MODE == MODE_ITERATE && st == MO_STATUS_OK) {
        update_state(w, n, k, m, ip[1], ip[2], tid);
        T* vp = (T*)a.vars + p * a.vars_stride;
        for (int i = tid; i < V; i += kThreads) vp[i] = w.vars[i];
      }
      if (a.delta) {
        T* dp = (T*)a.delta + p * a.delta_stride;
        for (int i = tid; i < V; i += kThreads) dp[i] = st == MO_STATUS_OK ? w.delta[i] : nanT<T>();
      }
      if (tid == 0) {  // constant indices only: a runtime-indexed register array would live in scratch
        if (a.alpha) {
          ((T*)a.alpha)[p * 2] = st == MO_STATUS_OK ? ip[1] : nanT<T>();
          ((T*)a.alpha)[p * 2 + 1] = st == MO_STATUS_OK ? ip[2] : nanT<T>();
        }
        if (a.ip_out) {
#pragma unroll
          for (int i = 0; i < MO_IP_RECORD; ++i) ((T*)a.ip_out)[p * MO_IP_RECORD + i] = ip[i];
        }
      }
      if (a.status && tid == 0) a.status[p] = st;
      MO_GSTAMP(6);
      continue;
    }

    // ---- MODE_SOLVE: QPInteriorPointSolver::Solve, qp.cc:100-151 ------------------------------------------------
    {
      const mo_solve_params& sp = a.sp;
      T* x = w.vars; T* s = w.vars + n; T* y = w.vars + n + m; T* z = w.vars + n + m + k;
      int term = MO_MAX_ITERATIONS, iters = 0;
      for (int i = tid; i < V; i += kThreads) { w.delta[i] = (T)0; w.daff[i] = (T)0; }
      __syncthreads();
      // ComputeInitialGuess, qp.cc:439-482
      if (st == MO_STATUS_OK && sp.initial_guess_method != MO_GUESS_USER_PROVIDED) {
        for (int i = tid; i < n; i += kThreads) x[i] = (T)0;
        for (int q = tid; q < k; q += kThreads) y[q] = (T)0;
        __syncthreads();
        if (sp.initial_guess_method == MO_GUESS_SOLVE_EQUALITY_CONSTRAINED) {     // :455-460
          eval_kkt<LARGE>(w, n, k, m, false, tid);
          st = assemble_and_factor<T, TG, R, LARGE>(w, n, k, m, false, tid, m_r);
          if (st == MO_STATUS_OK) {
            solve_for_update<T, LARGE>(w, n, k, m, (T)0, false, tid);
            for (int i = tid; i < n; i += kThreads) x[i] = w.delta[i];
            for (int q = tid; q < k; q += kThreads) y[q] = w.delta[n + m + q];
          }
          __syncthreads();
          load_qp(w, n, k, Gp, G_ld, cp, Ap, a.A_ld, bp, tid);  // the factorisation overwrote G
        }
        for (int v = tid; v < n; v += kThreads) {                                 // :464-467 ClampX in constraint order
          T xv = x[v];
          for (int c = 0; c < m; ++c) {
            if (w.cv[c] == v) {
              const T ca = w.ca[c], cb = w.cb[c];
              if (ca < (T)0) { const T lim = cb / -ca; xv = xv < lim ? xv : lim; }
              else { const T lim = -cb / ca; xv = xv > lim ? xv : lim; }
            }
          }
          x[v] = xv;
        }
        __syncthreads();
        for (int c = tid; c < m; c += kThreads) {                                 // :470-481
          const T sv = w.ca[c] * x[w.cv[c]] + w.cb[c];
          s[c] = sv > (T)1.0e-9 ? sv : (T)1.0e-9;
          z[c] = (T)1 / s[c];
        }
        __syncthreads();
      }
      T mu = (T)sp.initial_mu;
      if (st == MO_STATUS_OK) {
        eval_kkt<LARGE>(w, n, k, m, true, tid);                                          // :110
        if (sp.initialize_mu_with_complementarity) { compute_mu(w, n, k, m, tid); mu = w.red[4]; }  // :115
      }
      T* iter_out = a.iterations ? (T*)a.iterations + (size_t)p * sp.max_iterations * MO_ITER_RECORD : nullptr;
      for (int it = 0; st == MO_STATUS_OK && it < sp.max_iterations; ++it) {
        T rec[MO_ITER_RECORD];
        compute_errors(w, n, k, m, mu, tid);                                      // :118
        rec[0] = w.red[0]; rec[1] = w.red[1]; rec[2] = w.red[2]; rec[3] = w.red[3];
        __syncthreads();
        // Iterate, qp.cc:153-201 (its leading EvaluateKKTConditions would recompute the residual we already hold)
        st = newton_direction<T, TG, R, LARGE>(w, n, k, m, mu, sp.barrier_strategy, (T)0.995, rec + 8, tid, m_r);
        if (st != MO_STATUS_OK) break;
        update_state(w, n, k, m, rec[9], rec[10], tid);
        load_qp(w, n, k, Gp, G_ld, cp, Ap, a.A_ld, bp, tid);
        eval_kkt<LARGE>(w, n, k, m, true, tid);                                          // :125
        compute_errors(w, n, k, m, mu, tid);                                      // :127
        rec[4] = w.red[0]; rec[5] = w.red[1]; rec[6] = w.red[2]; rec[7] = w.red[3];
        __syncthreads();
        compute_mu(w, n, k, m, tid);
        const T cur_mu = w.red[4];
        if (iter_out && tid == 0) {
#pragma unroll
          for (int i = 0; i < MO_ITER_RECORD; ++i) iter_out[(size_t)it * MO_ITER_RECORD + i] = rec[i];
        }
        iters = it + 1;
        T kmax = rec[4];
        kmax = rec[5] > kmax ? rec[5] : kmax; kmax = rec[6] > kmax ? rec[6] : kmax; kmax = rec[7] > kmax ? rec[7] : kmax;
        if (kmax < (T)sp.termination_kkt_tol && cur_mu < (T)sp.termination_complementarity_tol) {  // :132-137
          term = MO_SATISFIED_KKT_TOL;
          break;
        }
        if (kmax <= mu || !sp.decrease_mu_only_on_small_error) {                  // :140-146
          if (sp.barrier_strategy == MO_FIXED_DECREASE) mu *= (T)sp.sigma;
          else mu = (T)sp.sigma * cur_mu;
        }
        __syncthreads();
      }
      __syncthreads();
      T* vp = (T*)a.vars + p * a.vars_stride;
      for (int i = tid; i < V; i += kThreads) vp[i] = w.vars[i];
      if (tid == 0) {
        if (a.termination) a.termination[p] = term;
        if (a.num_iterations) a.num_iterations[p] = iters;
        if (a.status) a.status[p] = st;
        if (a.lagrange) {                                                         // qp.cc:539-546
          T mn = nanT<T>(), linf = nanT<T>();
          if (k > 0) {
            mn = y[0]; linf = absT(y[0]);
            for (int q = 1; q < k; ++q) { mn = y[q] < mn ? y[q] : mn; linf = absT(y[q]) > linf ? absT(y[q]) : linf; }
          }
          ((T*)a.lagrange)[p * 2] = mn; ((T*)a.lagrange)[p * 2 + 1] = linf;
        }
      }
    }
  }
#ifdef MO_GENERIC_STAMPS
  if (threadIdx.x == 0 && a.debug) {
    for (int i = 0; i < 8; ++i) atomicAdd(a.debug + i, gstamp_acc[i]);
  }
#endif
}


// ---- QPNullSpaceSolver::Solve (qp.cc:679-729) ---------------------------------------------------------------------------------
// One workgroup per problem, everything LDS-resident in the generic kernel's workspace (m = 0): G (full symmetric) in H[0:n, 0:n],
// M = A_eq^T (n x k) in the rows n.. of H (M(i, q) = H[n + q + i ldh]).  The reference's steps, restated without forming Q:
//   colPivHouseholderQr of A_eq^T (:687)       k Householder steps with column pivoting on the largest remaining column norm; reflectors
//                                              stay in M below R (v_0 = 1 implicit); rank = #{|R_jj| > |R|_max eps min(n, k)} (Eigen's default)
//   u = Q1 R1^-T P^T (-b_eq)       (:703-704)  forward substitution + the reflectors applied to [t; 0]
//   G_reduced = Q2^T G Q2          (:708)      two-sided application of the reflectors to G (symmetric rank-2 updates), trailing block
//   LLT, fails iff a pivot <= 0    (:711-714)  right-looking Cholesky in place -> MO_STATUS_NOT_POSITIVE_DEFINITE
//   y = -(Q2^T (c + G u)), LL^T    (:718-721)  c + G u is formed BEFORE G is transformed; reflectors applied to it; two triangular solves
//   x = u + Q2 y                   (:725)      the reflectors applied to [0; y]
// A rank-deficient A_eq (rank r < k): Q1 = first r columns, R1 = leading r x r block (the reference's own k x k solve against Q1's r
// columns is a size mismatch there -- an Eigen assertion --, so this follows the consistent reading: solve with R1 only).

// LDS layout of the null-space kernel: G (n x n, odd leading dimension) | M = A_eq^T (n x k, same leading dimension) | u, g, pv, wv,
// cvec (n each) | beq, tau, pidx, nrm (k each) | J row chunk | 16 scalars | 8 flags.  Tighter than the generic (n + k)^2 workspace:
// n = 128, k = 16 fits the 160 KiB of a CU.
template <typename T>
__host__ __device__ inline size_t nullspace_elems(int n, int k, int m_r) {
  const int cr = chunk_rows_for(n, m_r, (int)sizeof(T));
  return (size_t)odd_ld(n) * (n + k) + 5 * (size_t)n + 4 * (size_t)k + (size_t)cr * n + cr + 16;
}
template <typename T> struct NullOps {
  const Ws<T>& w; T* Mb; int n, k;
  __device__ inline T& G(int i, int l) const { return w.H[i + (size_t)l * w.ldh]; }
  __device__ inline T& M(int i, int q) const { return Mb[i + (size_t)q * w.ldh]; }
  // vec <- H_j vec for the reflector stored in column j of M (wave 0 only; vec has n entries in LDS)
  __device__ inline void reflect(T* vec, int j, int lane) const {
    const T tau = w.invd[j];
    if (tau == (T)0) return;  // wave-uniform
    T s = (T)0;
    for (int i = j + lane; i < n; i += 64) s += (i == j ? (T)1 : M(i, j)) * vec[i];
    s = wave_sum(s) * tau;
    for (int i = j + lane; i < n; i += 64) vec[i] -= s * (i == j ? (T)1 : M(i, j));
    wave_lds_fence();
  }
};

template <typename T>
__global__ __launch_bounds__(kMaxThreads) void nullspace_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ws<T> w;
  const int n = a.n, k = a.k, m_r = a.m_r;
  T* lp = reinterpret_cast<T*>(smem);
  w.ldh = odd_ld(n);
  w.H = lp; lp += (size_t)w.ldh * n;
  T* const Mb = lp; lp += (size_t)w.ldh * k;
  T* const u = lp; lp += n;     // particular solution
  T* const g = lp; lp += n;     // c + G u, later Q^T (c + G u), later Q [0; y]
  T* const pv = lp; lp += n;    // scratch of the two-sided update
  T* const wv = lp; lp += n;
  w.cvec = lp; lp += n;
  w.beq = lp; lp += k;
  T* const tau = lp; lp += k;   // Householder coefficients
  T* const pidx = lp; lp += k;  // column permutation (stored as T)
  T* const nrm = lp; lp += k;   // column norms
  w.chunk_rows = chunk_rows_for(n, m_r, (int)sizeof(T));
  w.region = 0;
  w.Jc = lp; lp += (size_t)w.chunk_rows * n; w.rc = lp; lp += w.chunk_rows;
  w.red = lp; lp += 16;
  w.iflag = reinterpret_cast<int*>(lp);
  w.invd = tau;                 // NullOps::reflect reads the coefficients through w.invd
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const NullOps<T> op{w, Mb, n, k};
  const T eps = sizeof(T) == 8 ? (T)2.220446049250313e-16 : (T)1.1920929e-7f;
  for (long long p = blockIdx.x; p < a.batch; p += gridDim.x) {
    __syncthreads();
    const bool j_level = a.J != nullptr;
    const T* Ap = (const T*)a.A + p * a.A_stride;
    const T* bp = (const T*)a.b + p * a.b_stride;
    if (tid < 8) w.iflag[tid] = 0;
    load_qp(w, n, 0, j_level ? (const T*)nullptr : (const T*)a.G + p * a.G_stride, a.G_ld, j_level ? (const T*)nullptr : (const T*)a.c + p * a.c_stride,
            (const T*)nullptr, 0, (const T*)nullptr, tid);
    for (int idx = tid; idx < n * k; idx += kThreads) {                       // M = A_eq^T
      const int i = idx / k, q = idx - i * k;
      op.M(i, q) = Ap[q + (size_t)i * a.A_ld];
    }
    for (int q = tid; q < k; q += kThreads) w.beq[q] = bp[q];
    if (j_level) {
      const T lam = a.lambda_vec ? ((const T*)a.lambda_vec)[p * a.lambda_vec_stride] : (T)a.lambda;
      // the register tiling of J^T J covers n <= TG R columns: 8 x 6 for single-wave workgroups (n + k <= 48), 16 x 9 up to n = 144 and
      // 16 x 12 beyond (only fp32 plans get there: n = 145 ... 192 still fits the LDS in fp32, in fp64 n <= 128 does)
      if (kThreads == 64) accumulate_jtj<T, 8, 6>(w, n, m_r, (const T*)a.J + p * a.J_stride, a.J_ld, a.J_row_major, (const T*)a.r + p * a.r_stride, lam, tid);
      else if (n <= 144) accumulate_jtj<T, 16, 9>(w, n, m_r, (const T*)a.J + p * a.J_stride, a.J_ld, a.J_row_major, (const T*)a.r + p * a.r_stride, lam, tid);
      else if constexpr (sizeof(T) == 4) accumulate_jtj<T, 16, 12>(w, n, m_r, (const T*)a.J + p * a.J_stride, a.J_ld, a.J_row_major, (const T*)a.r + p * a.r_stride, lam, tid);
    }
    for (int l = wave; l < n; l += kWaves)   // selfadjointView<Lower>: mirror the lower triangle
      for (int i = lane; i < l; i += 64) op.G(i, l) = op.G(l, i);
    for (int q = tid; q < k; q += kThreads) { tau[q] = (T)0; pidx[q] = (T)q; }
    __syncthreads();

    // ---- Householder QR of M = A_eq^T with column pivoting (qp.cc:687)
    T maxpivot = (T)0;
    for (int j = 0; j < k; ++j) {
      for (int q = j + wave; q < k; q += kWaves) {
        T s = (T)0;
        for (int i = j + lane; i < n; i += 64) { const T v = op.M(i, q); s += v * v; }
        s = wave_sum(s);
        if (lane == 0) nrm[q] = s;
      }
      __syncthreads();
      int piv = j;
      T big = nrm[j];
      for (int q = j + 1; q < k; ++q) if (nrm[q] > big) { big = nrm[q]; piv = q; }   // uniform: every thread reads the same LDS values
      if (big == (T)0) break;                     // the remaining columns are exactly zero: no further reflectors
      __syncthreads();
      if (piv != j) {
        for (int i = tid; i < n; i += kThreads) { const T t = op.M(i, j); op.M(i, j) = op.M(i, piv); op.M(i, piv) = t; }
        if (tid == 0) { const T t = pidx[j]; pidx[j] = pidx[piv]; pidx[piv] = t; }
      }
      __syncthreads();
      // makeHouseholder on x = M(j.., j): beta = -sign(x0) |x|, tau = (beta - x0) / beta, essential = tail / (x0 - beta)
      const T c0 = op.M(j, j);
      const T tail2 = big - c0 * c0 > (T)0 ? big - c0 * c0 : (T)0;
      T beta = c0, tj = (T)0;
      if (tail2 > (T)0) {
        T tsum = (T)0;                            // recomputed (not big - c0^2): no cancellation
        for (int i = j + 1 + lane; i < n; i += 64) { const T v = op.M(i, j); tsum += v * v; }
        tsum = wave_sum(tsum);
        if (tsum > (T)0) {
          beta = sqrtT(c0 * c0 + tsum);
          if (c0 >= (T)0) beta = -beta;
          tj = (beta - c0) / beta;
        }
      }
      __syncthreads();
      if (tj != (T)0) {
        const T inv = (T)1 / (c0 - beta);
        for (int i = j + 1 + tid; i < n; i += kThreads) op.M(i, j) *= inv;
      } else {
        for (int i = j + 1 + tid; i < n; i += kThreads) op.M(i, j) = (T)0;
      }
      if (tid == 0) { op.M(j, j) = beta; tau[j] = tj; }
      __syncthreads();
      if (tj != (T)0) {
        for (int q = j + 1 + wave; q < k; q += kWaves) {   // remaining columns <- H_j columns
          T s = (T)0;
          for (int i = j + lane; i < n; i += 64) s += (i == j ? (T)1 : op.M(i, j)) * op.M(i, q);
          s = wave_sum(s) * tj;
          for (int i = j + lane; i < n; i += 64) op.M(i, q) -= s * (i == j ? (T)1 : op.M(i, j));
        }
      }
      maxpivot = absT(beta) > maxpivot ? absT(beta) : maxpivot;
      __syncthreads();
    }
    int rank = 0;
    {
      const T thr = maxpivot * eps * (T)(n < k ? n : k);
      for (int j = 0; j < k && j < n; ++j) rank += (absT(op.M(j, j)) > thr) ? 1 : 0;
    }

    // ---- u = Q1 R1^-T P^T (-b_eq) (qp.cc:703-704); g = c + G u; g <- Q^T g
    if (wave == 0) {
      for (int i = lane; i < n; i += 64) u[i] = (T)0;
      if (lane == 0) {
        for (int i = 0; i < rank; ++i) {          // R1^T t = rhs, rhs_i = -b[pidx_i]
          T acc = -w.beq[(int)pidx[i]];
          for (int jj = 0; jj < i; ++jj) acc -= op.M(jj, i) * u[jj];
          u[i] = acc / op.M(i, i);
        }
      }
      wave_lds_fence();
      for (int j = (k < n ? k : n) - 1; j >= 0; --j) op.reflect(u, j, lane);   // Q [t; 0] = H_0 ... H_{k-1} [t; 0]
    }
    __syncthreads();
    for (int i = tid; i < n; i += kThreads) {
      T acc = w.cvec[i];
      for (int l = 0; l < n; ++l) acc += op.G(i, l) * u[l];
      g[i] = acc;
    }
    __syncthreads();
    if (wave == 0)
      for (int j = 0; j < k && j < n; ++j) op.reflect(g, j, lane);             // Q^T g = H_{k-1} ... H_0 g
    __syncthreads();

    // ---- G <- Q^T G Q (qp.cc:708), reflector by reflector on the block that still matters (indices >= j)
    for (int j = 0; j < k && j < n; ++j) {
      const T tj = tau[j];
      if (tj == (T)0) continue;                                                // uniform
      for (int i = j + tid; i < n; i += kThreads) {                            // p = tau G v
        T acc = (T)0;
        for (int l = j; l < n; ++l) acc += op.G(i, l) * (l == j ? (T)1 : op.M(l, j));
        pv[i] = acc * tj;
      }
      __syncthreads();
      if (wave == 0) {
        T s = (T)0;
        for (int i = j + lane; i < n; i += 64) s += (i == j ? (T)1 : op.M(i, j)) * pv[i];
        s = wave_sum(s) * (T)0.5 * tj;                                         // alpha = tau (v^T p) / 2
        for (int i = j + lane; i < n; i += 64) wv[i] = pv[i] - s * (i == j ? (T)1 : op.M(i, j));
      }
      __syncthreads();
      for (int l = j + wave; l < n; l += kWaves) {                             // G -= v w^T + w v^T
        const T vl = l == j ? (T)1 : op.M(l, j), wl = wv[l];
        for (int i = j + lane; i < n; i += 64) op.G(i, l) -= (i == j ? (T)1 : op.M(i, j)) * wl + wv[i] * vl;
      }
      __syncthreads();
    }

    // ---- LLT of G_reduced = G[rank.., rank..] (qp.cc:711-714), right-looking, lower triangle
    const int q0 = rank, qn = n - rank;
    int st = MO_STATUS_OK;
    for (int kk = 0; kk < qn; ++kk) {
      const T d = op.G(q0 + kk, q0 + kk);
      if (!(d > (T)0)) { st = MO_STATUS_NOT_POSITIVE_DEFINITE; break; }        // uniform (every thread reads the same LDS word)
      const T l = sqrtT(d), inv = (T)1 / l;
      __syncthreads();
      for (int i = kk + tid; i < qn; i += kThreads) op.G(q0 + i, q0 + kk) = i == kk ? l : op.G(q0 + i, q0 + kk) * inv;
      __syncthreads();
      for (int jj = kk + 1 + wave; jj < qn; jj += kWaves) {
        const T wj = op.G(q0 + jj, q0 + kk);
        for (int i = jj + lane; i < qn; i += 64) op.G(q0 + i, q0 + jj) -= op.G(q0 + i, q0 + kk) * wj;
      }
      __syncthreads();
    }
    __syncthreads();

    // ---- y = -(Q2^T (c + G u)) through L L^T (qp.cc:718-721); x = u + Q2 y (:725)
    if (st == MO_STATUS_OK && wave == 0) {
      T* y = g + q0;                                                           // rhs_y = -g[rank..] in place
      for (int i = lane; i < qn; i += 64) y[i] = -y[i];
      wave_lds_fence();
      for (int kk = 0; kk < qn; ++kk) {                                        // forward, column oriented
        const T yk = y[kk] / op.G(q0 + kk, q0 + kk);
        wave_lds_fence();                                                      // every lane has read y[kk] before lane 0 replaces it
        if (lane == 0) y[kk] = yk;
        for (int i = kk + 1 + lane; i < qn; i += 64) y[i] -= op.G(q0 + i, q0 + kk) * yk;
        wave_lds_fence();
      }
      for (int kk = qn - 1; kk >= 0; --kk) {                                   // backward with L^T
        T s = (T)0;
        for (int i = kk + 1 + lane; i < qn; i += 64) s += op.G(q0 + i, q0 + kk) * y[i];
        s = wave_sum(s);
        if (lane == 0) y[kk] = (y[kk] - s) / op.G(q0 + kk, q0 + kk);
        wave_lds_fence();
      }
      for (int i = lane; i < q0; i += 64) g[i] = (T)0;                         // [0; y]
      wave_lds_fence();
      for (int j = (k < n ? k : n) - 1; j >= 0; --j) op.reflect(g, j, lane);
    }
    __syncthreads();
    bool bad = false;
    if (st == MO_STATUS_OK) {
      for (int i = tid; i < n; i += kThreads) bad |= !finiteT(u[i] + g[i]);
      if (bad) w.iflag[3] = 1;
    }
    __syncthreads();
    if (st == MO_STATUS_OK && w.iflag[3]) st = MO_STATUS_NONFINITE;
    T* xo = (T*)a.delta + p * a.delta_stride;
    for (int i = tid; i < n; i += kThreads) xo[i] = st == MO_STATUS_OK ? u[i] + g[i] : nanT<T>();
    if (a.status && tid == 0) a.status[p] = st;
  }
}

}  // namespace

size_t generic_lds_bytes(const KernelArgs& a, int elem_size) {
  const size_t e = elem_size == 8 ? ws_elems<double>(a.n, a.k, a.m, a.m_r) : ws_elems<float>(a.n, a.k, a.m, a.m_r);
  size_t bytes = e * elem_size + (size_t)(a.m + 8) * sizeof(int);
  return (bytes + 15) & ~(size_t)15;
}

// The LDS-resident kernel takes n + k <= 192 (the register-distributed factorisation: 16 x 16 threads x 12 x 12 blocks) whose workspace
// fits the 160 KiB of a CU; everything beyond runs with H in a global workspace (LARGE).
// Systems of at least this size take the LARGE path (H in the plan's global workspace) although the LDS-resident kernel covers n + k <= 192
// where H fits the LDS: since round 4 LARGE is the faster kernel from n + k = 72 on (one box, batch 65 536 / 16 384; LDS-resident vs LARGE:
// n + k = 56: 7.84 vs 5.38 M steps/s, 64: 5.72 vs 5.21 M, 72 (BASELINE configs[2] shape): 2.86 vs 3.87 M, 130 (n = 90, k = 40): 0.61 vs
// 1.86 M, Solve 92.9 k vs 229 k/s) -- two workgroups per CU whatever the size, matrix-core J^T J, no barrier per pivot.
#ifndef MO_LARGE_MIN_P
#define MO_LARGE_MIN_P 72
#endif
bool generic_needs_large(const KernelArgs& a, int elem_size) { return a.n + a.k >= MO_LARGE_MIN_P || generic_lds_bytes(a, elem_size) > 160 * 1024; }
size_t generic_large_lds_bytes(const KernelArgs& a, int elem_size) {
  const int nb_cols = panel_cols_for(a.n, a.k, a.m, a.m_r, elem_size);
  if (nb_cols == 0) return (size_t)1 << 30;   // not even the vectors and an 8-column panel fit
  // the J^T J staging needs four rows of its widest super-block (two 128-column blocks, stride 272, + r) in the panel region
  if (a.m_r > 0 && (size_t)((a.n + a.k) | 1) * nb_cols < 4 * (size_t)(large_jtj_stride(a.n < 256 ? a.n : 256) + 1)) return (size_t)1 << 30;
  const size_t e = elem_size == 8 ? ws_elems_large<double>(a.n, a.k, a.m, a.m_r) : ws_elems_large<float>(a.n, a.k, a.m, a.m_r);
  return (e * elem_size + (size_t)(a.m + 8) * sizeof(int) + 15) & ~(size_t)15;
}
size_t generic_large_workspace_elems(const KernelArgs& a) { const size_t P = (size_t)a.n + a.k; return P * (size_t)large_ld((int)P); }
int generic_large_grid(const KernelArgs& a, int elem_size, int num_cus) {
  const size_t lds = generic_large_lds_bytes(a, elem_size);
  int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 2) per_cu = 2;   // the kernel is register-budgeted for two workgroups per CU (__launch_bounds__(256, 2)): more would only queue,
                                // and every workgroup of the grid owns a slot of the plan's H workspace (n + k = 72: 46 KB each)
  return num_cus * per_cu;
}

static hipError_t launch_generic_large(const KernelArgs& a, int dtype, int num_cus, hipStream_t stream) {
  const int elem = dtype == MO_F64 ? 8 : 4;
  const size_t lds = generic_large_lds_bytes(a, elem);
  if (lds > 160 * 1024 || !a.H_work) return hipErrorInvalidValue;
  long long grid = generic_large_grid(a, elem, num_cus);
  if (grid > a.batch) grid = a.batch;
  if (a.H_work_slots > 0 && grid > a.H_work_slots) grid = a.H_work_slots;   // one workspace per workgroup; the kernel strides over the batch
  if (grid < 1) grid = 1;
  hipError_t e = hipSuccess;
#define MO_LAUNCH_LARGE(TYPE, MODE_)                                                                                                     \
  do {                                                                                                                                  \
    e = hipFuncSetAttribute((const void*)kkt_generic_kernel<TYPE, MODE_, 16, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                                                                      \
    hipLaunchKernelGGL((kkt_generic_kernel<TYPE, MODE_, 16, 6, true>), dim3((unsigned)grid), dim3(256), lds, stream, a);                 \
  } while (0)
#define MO_DISPATCH_LARGE(TYPE)                                       \
  switch (a.mode) {                                                   \
    case MODE_LINEARIZE: MO_LAUNCH_LARGE(TYPE, MODE_LINEARIZE); break; \
    case MODE_RESIDUAL: MO_LAUNCH_LARGE(TYPE, MODE_RESIDUAL); break;   \
    case MODE_STEP: MO_LAUNCH_LARGE(TYPE, MODE_STEP); break;           \
    case MODE_ITERATE: MO_LAUNCH_LARGE(TYPE, MODE_ITERATE); break;     \
    case MODE_SOLVE: MO_LAUNCH_LARGE(TYPE, MODE_SOLVE); break;         \
    default: return hipErrorInvalidValue;                             \
  }
#ifdef MO_GENERIC_LARGE_STEP_ONLY   // development builds: the fp64 step kernel alone (register-pressure probes)
  if (dtype != MO_F64 || a.mode != MODE_STEP) return hipErrorInvalidValue;
  MO_LAUNCH_LARGE(double, MODE_STEP);
#else
  if (dtype == MO_F64) { MO_DISPATCH_LARGE(double) } else { MO_DISPATCH_LARGE(float) }
#endif
#undef MO_DISPATCH_LARGE
#undef MO_LAUNCH_LARGE
  return hipGetLastError();
}

hipError_t launch_generic(const KernelArgs& a, int dtype, int num_cus, hipStream_t stream) {
  const int elem = dtype == MO_F64 ? 8 : 4;
  if (generic_needs_large(a, elem)) return launch_generic_large(a, dtype, num_cus, stream);
#ifdef MO_GENERIC_LARGE_ONLY   // development builds of the LARGE path alone (a fifth of the translation unit's compile time)
  return hipErrorInvalidValue;
#else
  const size_t lds = generic_lds_bytes(a, elem);
#ifdef MO_TUNING
  static const int env_threads = [] { const char* e = getenv("MO_GENERIC_THREADS"); return e ? atoi(e) : 0; }();  // tuning knob
#else
  constexpr int env_threads = 0;
#endif
  int threads = (a.n + a.k <= 48) ? 64 : kMaxThreads;  // measured: cfg 2 (P = 36) 8.5 M vs 6.8 M steps/s, cfg 3 (P = 72) 0.75 M vs 1.9 M
  if (env_threads == 256) threads = 256;  // (the register-distributed factorisation wants an 8 x 8 or a 16 x 16 thread grid)
  const int max_per_cu = 32 / (threads / 64);
  int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > max_per_cu) per_cu = max_per_cu;
  long long grid = (long long)num_cus * per_cu;
  if (grid > a.batch) grid = a.batch;
  if (grid < 1) grid = 1;
  hipError_t e = hipSuccess;
  const int P = a.n + a.k;
#define MO_LAUNCH_GENERIC(TYPE, MODE_, TG_, R_)                                                                               \
  do {                                                                                                                       \
    e = hipFuncSetAttribute((const void*)kkt_generic_kernel<TYPE, MODE_, TG_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                                       \
    if (e != hipSuccess) return e;                                                                                           \
    hipLaunchKernelGGL((kkt_generic_kernel<TYPE, MODE_, TG_, R_>), dim3((unsigned)grid), dim3(threads), lds, stream, a);      \
  } while (0)
// (n + k >= MO_LARGE_MIN_P = 72 went to launch_generic_large above: the 16 x 16 grids with 9 and 12 blocks per thread of rounds 2 - 3 are
// only instantiated where the knob keeps larger systems on the LDS-resident kernel)
#if MO_LARGE_MIN_P <= 81
#define MO_LAUNCH_FACTORISING(TYPE, MODE_)                     \
  do {                                                         \
    if (threads == 64) MO_LAUNCH_GENERIC(TYPE, MODE_, 8, 6);   \
    else MO_LAUNCH_GENERIC(TYPE, MODE_, 16, 5);                \
  } while (0)
#else
#define MO_LAUNCH_FACTORISING(TYPE, MODE_)                     \
  do {                                                         \
    if (threads == 64) MO_LAUNCH_GENERIC(TYPE, MODE_, 8, 6);   \
    else if (P <= 80) MO_LAUNCH_GENERIC(TYPE, MODE_, 16, 5);   \
    else if (P <= 144) MO_LAUNCH_GENERIC(TYPE, MODE_, 16, 9);  \
    else MO_LAUNCH_GENERIC(TYPE, MODE_, 16, 12);               \
  } while (0)
#endif
#define MO_DISPATCH_MODE(TYPE)                                          \
  switch (a.mode) {                                                     \
    case MODE_LINEARIZE: MO_LAUNCH_FACTORISING(TYPE, MODE_LINEARIZE); break;   \
    case MODE_RESIDUAL: MO_LAUNCH_FACTORISING(TYPE, MODE_RESIDUAL); break;     \
    case MODE_STEP: MO_LAUNCH_FACTORISING(TYPE, MODE_STEP); break;       \
    case MODE_ITERATE: MO_LAUNCH_FACTORISING(TYPE, MODE_ITERATE); break; \
    case MODE_SOLVE: MO_LAUNCH_FACTORISING(TYPE, MODE_SOLVE); break;     \
    default: return hipErrorInvalidValue;                               \
  }
  if (dtype == MO_F64) { MO_DISPATCH_MODE(double) } else { MO_DISPATCH_MODE(float) }
#undef MO_DISPATCH_MODE
#undef MO_LAUNCH_FACTORISING
#undef MO_LAUNCH_GENERIC
  return hipGetLastError();
#endif
}

size_t nullspace_lds_bytes(int n, int k, int m_r, int elem_size) {
  const size_t e = elem_size == 8 ? nullspace_elems<double>(n, k, m_r) : nullspace_elems<float>(n, k, m_r);
  return ((e * elem_size + 8 * sizeof(int)) + 15) & ~(size_t)15;
}

hipError_t launch_nullspace(const KernelArgs& a, int dtype, int num_cus, hipStream_t stream) {
  const int elem = dtype == MO_F64 ? 8 : 4;
  KernelArgs b = a;
  b.m = 0;
  const size_t lds = nullspace_lds_bytes(a.n, a.k, a.m_r, elem);
  if (a.J && a.n > (dtype == MO_F64 ? 144 : 192)) return hipErrorInvalidValue;  // beyond the J^T J register tiling (mo_plan_create's n + k <= 192 keeps fp32 inside)
  const int threads = (a.n + a.k <= 48) ? 64 : kMaxThreads;
  const int max_per_cu = 32 / (threads / 64);
  int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > max_per_cu) per_cu = max_per_cu;
  long long grid = (long long)num_cus * per_cu;
  if (grid > a.batch) grid = a.batch;
  if (grid < 1) grid = 1;
  hipError_t e;
  if (dtype == MO_F64) {
    e = hipFuncSetAttribute((const void*)nullspace_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((nullspace_kernel<double>), dim3((unsigned)grid), dim3(threads), lds, stream, b);
  } else {
    e = hipFuncSetAttribute((const void*)nullspace_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((nullspace_kernel<float>), dim3((unsigned)grid), dim3(threads), lds, stream, b);
  }
  return hipGetLastError();
}

}  // namespace mo
