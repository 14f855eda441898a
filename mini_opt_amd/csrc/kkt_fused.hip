// kkt_fused.hip -- fused single-wavefront KKT kernels for gfx950 (MI355X), fp64: Newton step, Iterate / Solve, linearisation.
//
// One 64-lane wavefront owns one QP at a time and keeps the WHOLE reduced KKT matrix in registers as 16x16 tiles in the
// v_mfma_f64_16x16x4_f64 C/D fragment layout (lane (g = l>>4, j = l&15), register t holds element (row g + 4t, column j)).
// Shapes: any n <= 128 (padded inside the kernel to a 32 / 64 / 96 / 128 tile grid), k <= 31 (16 .. 31: a second y tile, kkt_fused_ny2.hip),
// m <= 256 (one, two or four constraint slots per lane; four: kkt_fused_mc4.hip); see fused_supported().
//
//   P0  The small per-problem vectors (x, s, z, y, b_eq, constraints) go global -> LDS by dword DMA, behind the ring fill.
//   P1  J (m_r x n, row-major) is streamed from HBM exactly once through a per-wave LDS-DMA ring straight into MFMA operand
//       registers: G = J^T J is accumulated on the matrix cores as the upper block triangle of tiles, c = J^T r on the VALU.
//       A lane's 16-byte piece holds two adjacent columns, which induces a fixed permutation of the variables
//       (position 16c + i  <->  column 32(c>>1) + 2i + (c&1)); the KKT system is solved in that order.  QP-level input loads the
//       tiles from G instead (identity order).                                     [residual.hpp:206-224, nonlinear.cc:182-189]
//   P2-4 lambda and the barrier diagonal Sigma (qp.cc:293-298) go onto the diagonal tiles; A_eq^T and the right-hand
//       side form one more tile column [A_eq^T | rhs] (the rhs in column 15).
//   P5  Block LDL^T with 16x16 pivot blocks: each diagonal tile is inverted in registers by a symmetric sweep (wave
//       broadcasts only), the panel Z = T^-1 U and the trailing update U_bc -= U_ab^T Z_c run on the matrix cores.
//       Because the right-hand side rides along as a tile column, the forward substitution is free.  [qp.cc:302]
//   P6  Backward substitution on the VALU, arranged so that no fragment-layout conversion is needed.
//   P7  ds, dz (qp.cc:359-363), alpha (qp.cc:485-507), status, stores of delta.
//
// The step kernel solves for the NEW iterate (x+, -y+): [G+Sigma, A^T; A, 0] [x+; -y+] = [rhs_x; -b_eq] with
// rhs_x[v] = -c[v] + sum_{i on v} a_i (z_i (s_i - b_i) + mu) / s_i, which is the reference's reduced system
// (qp.cc:255-268) with K [x; -y] added to both sides: identical direction delta = (x+ - x, ds, y+ - y, dz) without the
// G x, A x and A^T y products of EvaluateKKTConditions (qp.cc:404-408).  The Solve kernel keeps the reference's residual form.
//
// Roofline: per problem 73.6 KB of algorithmic HBM traffic and 440 f64 MFMAs (320 for J^T J at n = 64); see DESIGN.md.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "mo_kernels.h"

namespace mo {
namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

// Phase stamps exist only in the diagnostic build of tools/phase_timer.hip; the product kernel executes none.
#ifdef MO_FUSED_STAMPS
#define MO_STAMP(i)                                                                                   \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    unsigned long long t__;                                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                       \
    stamp_acc[i] += t__ - stamp_prev;                                                                 \
    stamp_prev = t__;                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#else
#define MO_STAMP(i) do { } while (0)
#endif

constexpr int kRC = 15;  // tile column that carries the right-hand side in the [A_eq^T | rhs] tile column

// ---- cross-lane helpers ------------------------------------------------------------------------------------------
__device__ inline double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
// value of `v` in lane (byte_addr / 4)
__device__ inline double bpermute_f64(int byte_addr, double v) {
  int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
  int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int LANE_IN_ROW> __device__ inline double row_bcast(double v) {  // every lane of a 16-lane row <- lane LANE_IN_ROW
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + LANE_IN_ROW, 0xf, 0xf, false);  // all lanes written: no "old" value to set up
  hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + LANE_IN_ROW, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ inline double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Sum / min over the 16 lanes of each row, result in every lane of the row.  DPP only (no LDS crossbar):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror.
// MO_ROWSUM_SWIZZLE (A/B builds): the same butterfly (bit-identical sums) through ds_swizzle_b32 -- the exchange goes over the LDS crossbar
// (two LDS-pipe instructions per step) and only the four adds stay on the VALU: 4 instead of 12 VALU instructions per row sum.
template <int PATTERN> __device__ inline double swizzle_f64(double v) {
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN);
  return __hiloint2double(hi, lo);
}
__device__ inline double row_sum(double v) {
#ifdef MO_ROWSUM_SWIZZLE
  v += swizzle_f64<0x041F>(v);   // bit mode: lane ^ 1
  v += swizzle_f64<0x081F>(v);   // lane ^ 2
  v += swizzle_f64<0x101F>(v);   // lane ^ 4
  v += swizzle_f64<0x201F>(v);   // lane ^ 8
#else
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
#endif
  return v;
}
// FOUR row sums at once (round 4).  p0 .. p3 are four values per lane whose sums over the 16 lanes of each row are wanted; the plain way is four
// butterflies of four steps (48 VALU instructions: two 32-bit DPP moves and an add per step and value).  The transposed butterfly halves the number
// of live values at each of the first two steps instead -- lanes exchange with j ^ 1 and keep p0 / p2 (even) or p1 / p3 (odd), then with j ^ 2 --
// and finishes the one remaining value with two rotations inside the row: 27 VALU instructions, and lane j ends up with the sum of p[j & 3]
// (all four quads of the row hold the same four sums).  A 64-bit DPP move only knows row_newbcast on this part, so the exchanges stay 32-bit pairs.
// The first two steps add exactly what row_sum adds (a + b is commutative bit for bit); the last two pair the quad sums as (Q0 + Q3) + (Q2 + Q1)
// where row_sum's mirrors give (Q0 + Q1) + (Q3 + Q2): same sum, last-bit differences are possible.
__device__ inline int select_lanes32(unsigned long long mask, int if_set, int if_clear) {   // v_cndmask_b32 on a constant lane mask held in SGPRs
  int r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
  return r;
}
__device__ inline double select_lanes(unsigned long long mask, double if_set, double if_clear) {
  return __hiloint2double(select_lanes32(mask, __double2hiint(if_set), __double2hiint(if_clear)),
                          select_lanes32(mask, __double2loint(if_set), __double2loint(if_clear)));
}
// lanes exchange with j ^ 1 (XOR1) or j ^ 2: lanes of MASK keep `b` and send `a`, the others keep `a` and send `b`; every lane adds what it receives
template <int CTRL, unsigned long long MASK> __device__ inline double exchange_add(double a, double b) {
  const double keep = select_lanes(MASK, b, a), send = select_lanes(MASK, a, b);
  return keep + dpp_f64<CTRL>(send);
}
__device__ inline double row_sum4_scatter(double p0, double p1, double p2, double p3) {   // lane j: sum over its row of p[j & 3]
  const double q01 = exchange_add<0xB1, 0xAAAAAAAAAAAAAAAAull>(p0, p1);   // quad_perm [1,0,3,2]; odd lanes keep p1
  const double q23 = exchange_add<0xB1, 0xAAAAAAAAAAAAAAAAull>(p2, p3);
  double w = exchange_add<0x4E, 0xCCCCCCCCCCCCCCCCull>(q01, q23);         // quad_perm [2,3,0,1]; lanes with (j & 2) keep q23
  w += dpp_f64<0x124>(w);                                                  // row_ror:4  (lane j <- j - 4: the same j & 3)
  w += dpp_f64<0x128>(w);                                                  // row_ror:8
  return w;
}
__device__ inline double row_min(double v) {
  v = fmin(v, dpp_f64<0xB1>(v));
  v = fmin(v, dpp_f64<0x4E>(v));
  v = fmin(v, dpp_f64<0x141>(v));
  v = fmin(v, dpp_f64<0x140>(v));
  return v;
}
// v_permlane16_swap(a, b) -> {[a.R0, b.R0, a.R2, b.R2], [a.R1, b.R1, a.R3, b.R3]};  v_permlane32_swap(a, b) ->
// {[a.R0, a.R1, b.R0, b.R1], [a.R2, a.R3, b.R2, b.R3]}  (R = 16-lane row; verified by tools/microbench.hip).
struct Rows2 { double a, b; };
__device__ inline Rows2 swap16_f64(double v) {
  const u2v lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  const u2v hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  return Rows2{__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1])};
}
__device__ inline Rows2 swap32_f64(double v) {
  const u2v lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  const u2v hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  return Rows2{__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1])};
}
// Sum / min over the four rows (same lane-in-row), result in all four rows.
__device__ inline double cross_row_sum(double v) {
  const Rows2 p = swap16_f64(v);
  const double s = p.a + p.b;
  const Rows2 q = swap32_f64(s);
  return q.a + q.b;
}
__device__ inline double cross_row_min(double v) {
  const Rows2 p = swap16_f64(v);
  const double s = fmin(p.a, p.b);
  const Rows2 q = swap32_f64(s);
  return fmin(q.a, q.b);
}
// Row SG of v replicated into all four rows.
template <int SG> __device__ inline double bcast_from_row(double v) {
  const Rows2 p = swap16_f64(v);                  // p.a = [v0,v0,v2,v2], p.b = [v1,v1,v3,v3]
  const Rows2 q = swap32_f64((SG & 1) ? p.b : p.a);  // q.a = [y0,y1,y0,y1], q.b = [y2,y3,y2,y3]
  return (SG & 2) ? q.b : q.a;
}
__device__ inline double rcp_f64(double d) {
  double q = __builtin_amdgcn_rcp(d);
  q = fma(q, fma(-d, q, 1.0), q);
  q = fma(q, fma(-d, q, 1.0), q);
  return q;
}
__device__ inline d4 mfma4(const d4& a, const d4& b, d4 c) {  // c += A^T-fragment(a) * fragment(b) over the 16 tile rows
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], c, 0, 0, 0);
  return c;
}

// V16 vector (value at tile position 16c + j) <-> natural-order array.  J-level kernels solve in the pair-permuted variable
// order the 16-byte J loads induce (position 16c + i <-> variable 32(c>>1) + 2i + (c&1)); QP-level kernels (QPL: G, c given)
// use the identity, and then nothing needs 16-byte alignment.
template <int NT, bool QPL> __device__ inline void ldv(const double* arr, int j, double (&v)[NT]) {
  if (QPL) {
#pragma unroll
    for (int c = 0; c < NT; ++c) v[c] = arr[16 * c + j];
  } else {
#pragma unroll
    for (int h = 0; h < NT / 2; ++h) { const d2 t = *(const d2*)(arr + 32 * h + 2 * j); v[2 * h] = t[0]; v[2 * h + 1] = t[1]; }
  }
}
template <int NT, bool QPL> __device__ inline void stv(double* arr, int j, const double (&v)[NT]) {
  if (QPL) {
#pragma unroll
    for (int c = 0; c < NT; ++c) arr[16 * c + j] = v[c];
  } else {
#pragma unroll
    for (int h = 0; h < NT / 2; ++h) { d2 t; t[0] = v[2 * h]; t[1] = v[2 * h + 1]; *(d2*)(arr + 32 * h + 2 * j) = t; }
  }
}
// The same on the caller's arrays of length nn <= 16 NT (the kernels pad the system to whole tiles: padded variables have zero
// cost / constraint columns, a unit diagonal and therefore a zero solution).  J-level input has nn even, so a pair is in or out.
template <int NT, bool QPL> __device__ inline void ldv_n(const double* arr, int j, int nn, double (&v)[NT]) {
  if (QPL) {
#pragma unroll
    for (int c = 0; c < NT; ++c) v[c] = (16 * c + j < nn) ? arr[16 * c + j] : 0.0;
  } else {
#pragma unroll
    for (int h = 0; h < NT / 2; ++h) {  // two 8-byte accesses: the caller's state / direction vectors need no 16-byte alignment
      v[2 * h] = (32 * h + 2 * j < nn) ? arr[32 * h + 2 * j] : 0.0;
      v[2 * h + 1] = (32 * h + 2 * j + 1 < nn) ? arr[32 * h + 2 * j + 1] : 0.0;  // an odd nn ends inside a pair
    }
  }
}
template <int NT, bool QPL> __device__ inline void stv_n(double* arr, int j, int nn, const double (&v)[NT]) {
  if (QPL) {
#pragma unroll
    for (int c = 0; c < NT; ++c)
      if (16 * c + j < nn) arr[16 * c + j] = v[c];
  } else {
#pragma unroll
    for (int h = 0; h < NT / 2; ++h) {
      if (32 * h + 2 * j < nn) arr[32 * h + 2 * j] = v[2 * h];
      if (32 * h + 2 * j + 1 < nn) arr[32 * h + 2 * j + 1] = v[2 * h + 1];
    }
  }
}
// 1.0 where the variable at tile position 16c + j is padding (index >= nn), else 0.0
template <int NT, bool QPL> __device__ inline void pad_diag(int j, int nn, double (&v)[NT]) {
#pragma unroll
  for (int c = 0; c < NT; ++c) {
    const int nat = QPL ? 16 * c + j : 32 * (c >> 1) + 2 * j + (c & 1);
    v[c] = nat >= nn ? 1.0 : 0.0;
  }
}
// [A_eq^T | .] tile columns: element (row r = g + 4t of block c, column j) of y tile q = A_eq(16q + j, variable of position 16c + r).
// NY y tiles carry up to 16 NY - 1 equality rows (column 15 of the LAST tile column is the right-hand side).
template <int NT, bool QPL, int NY = 1>
__device__ inline void load_a_tiles(const double* Ap, int A_ld, int k, int nn, int g, int j, d4 (&U)[(NT + NY) * (NT + NY)]) {
  constexpr int NB = NT + NY;
  const int col_l = (QPL ? 1 : 2) * g;                             // per-lane part of the column index
  const double* Al = Ap + j + (size_t)col_l * A_ld;
#pragma unroll
  for (int c = 0; c < NT; ++c) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col_u = QPL ? (16 * c + 4 * t) : (32 * (c >> 1) + 8 * t + (c & 1));  // wave-uniform part
#pragma unroll
      for (int q = 0; q < NY; ++q)
        U[c * NB + NT + q][t] = (16 * q + j < k && col_u + col_l < nn) ? Al[16 * q + (size_t)col_u * A_ld] : 0.0;
    }
  }
}
// QP-level cost: the G tiles come straight from the caller's column-major G (only its lower triangle is read, qp.cc:289):
// tile (a, b), element (r, j) = G(16b + j, 16a + r) for b > a (128-byte rows across the lanes), mirrored inside diagonal tiles.
template <int NT, int NY = 1>
__device__ inline void load_g_tiles(const double* G, int ld, const double* cg, int nn, int g, int j, d4 (&U)[(NT + NY) * (NT + NY)],
                                    double (&cvec)[NT]) {
  constexpr int NB = NT + NY;
#pragma unroll
  for (int ta = 0; ta < NT; ++ta) {
#pragma unroll
    for (int tb = ta; tb < NT; ++tb) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = g + 4 * t;
        int row = 16 * tb + j, col = 16 * ta + r;
        if (ta == tb && r > j) { row = 16 * ta + r; col = 16 * ta + j; }
        U[ta * NB + tb][t] = (row < nn && col < nn) ? G[row + (size_t)col * ld] : 0.0;
      }
    }
    cvec[ta] = (16 * ta + j < nn) ? cg[16 * ta + j] : 0.0;
  }
}

// Lane index recomputed on the spot (two VALU ops).  The asm is volatile on purpose: nothing derived from it can be hoisted
// out of a loop and kept live in VGPRs the tiles need.
__device__ inline int lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// The kernel's argument block as the hardware sees it (kernarg segment, constant address space), behind a pointer the
// compiler cannot see through: loads through it stay where they are written instead of being hoisted to the kernel entry.
typedef const KernelArgs __attribute__((address_space(4)))* KArgs;
__device__ inline KArgs fresh_args() {
  KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

// LDS-DMA: every lane's 16 bytes at `gsrc` land at LDS byte address lds_dst + 16 * lane (no VGPR destination).
// hipcc does not count this load: its completion is waited for by hand with wait_vmcnt<N>() (loads retire in order).
__device__ inline void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// Scalar-base forms: global address = sbase (wave-uniform, SGPR pair) + voff (per-lane 32-bit byte offset) + IMM; the LDS
// address is M0 + IMM + 16 * lane (the instruction offset applies to BOTH sides, found the hard way).  The address
// arithmetic of a stream then lives on the scalar unit; the VALU (which the f64 MFMA shares) sees none of it.
// The base and the LDS address are wave-uniform by construction; readfirstlane says so to hipcc where its divergence analysis
// gives up (values carried around a loop with data-dependent exits) -- it folds away when the value already sits in SGPRs.
__device__ inline const void* uniform_ptr(const void* p) {
  const unsigned long long b = (unsigned long long)(uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return (const void*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
// The J stream's loads carry the non-temporal hint: J is read once, by one CU, and without the hint it washes the lines other waves
// re-read every pass of a Solve (spills, the parked tiles that did not fit the LDS) out of the XCD's L2 -- measured on the cfg 3 Solve:
// FETCH_SIZE 4.30 -> 3.23 GB per launch, +0.8 % solves/s; the step kernel is indifferent (A/B builds: -DMO_J_NO_NT).
#ifdef MO_J_NO_NT
#define MO_J_POLICY ""
#else
#define MO_J_POLICY " nt"
#endif
template <int IMM> __device__ inline void dma16_s(const void* sbase_in, unsigned voff, unsigned lds_dst_in) {
  const void* sbase = uniform_ptr(sbase_in);
  const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_in);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4" MO_J_POLICY "\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst), "i"(IMM)
      : "memory");
}
__device__ inline void dma4_s(const void* sbase_in, unsigned voff, unsigned lds_dst_in) {
  const void* sbase = uniform_ptr(sbase_in);
  const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_in);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
// Same with 4 bytes per lane: LDS byte address lds_dst + 4 * lane.  64 lanes move 32 consecutive doubles.
__device__ inline void dma4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// `count` consecutive doubles (count <= 256, wave-uniform) from global memory to LDS without touching a VGPR destination.
// `src` must be wave-uniform.
__device__ inline void dma_doubles(const double* src, unsigned lds_dst, int count, int lane) {
  const unsigned voff = 4u * (unsigned)lane;
  if (lane < 2 * count) dma4_s(src, voff, lds_dst);
  if (lane + 64 < 2 * count) dma4_s(reinterpret_cast<const char*>(src) + 256, voff, lds_dst + 256);
  if (count > 64) {  // wave-uniform; up to 128 doubles (n = 96 / 128)
    if (lane + 128 < 2 * count) dma4_s(reinterpret_cast<const char*>(src) + 512, voff, lds_dst + 512);
    if (lane + 192 < 2 * count) dma4_s(reinterpret_cast<const char*>(src) + 768, voff, lds_dst + 768);
  }
  if (count > 128) {  // wave-uniform; up to 256 doubles (four constraint slots per lane)
#pragma unroll
    for (int c = 4; c < 8; ++c)
      if (lane + 64 * c < 2 * count) dma4_s(reinterpret_cast<const char*>(src) + 256 * c, voff, lds_dst + 256 * c);
  }
}
template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ inline void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }  // single-wave workgroup

// Symmetric sweep of pivots 0..NPIV-1 of a symmetric 16x16 tile held in the C/D layout.  Afterwards the swept block holds
// -T11^-1, the swept x unswept block T11^-1 T12 (the solution for an augmented right-hand-side column) and the unswept
// block the Schur complement.  Returns false if a pivot is zero or not finite.
// Broadcast flavours (SW): the f64 MFMA and the VALU share one datapath on gfx950 (tools/coissue.hip: beside back-to-back
// f64 MFMAs a second wave issues only ~1.5 v_fma_f64 per MFMA), so VALU instructions are as precious as MFMAs, while the
// LDS crossbar (ds_bpermute) is idle but has ~100+ cycles of latency.  0: both broadcasts through ds_bpermute (fewest
// VALU instructions; wants >= 3-4 waves per SIMD), 1: row k through ds_bpermute, column k through DPP, 2: VALU only.
template <int K, int SW>
__device__ inline void sweep_step(d4& T, bool& ok, int g, int j) {
  constexpr int src_g = K & 3, src_t = K >> 2;
  const double rowreg = T[src_t];  // every broadcast below is taken before any register of T is modified
  const double d = readlane_f64(rowreg, 16 * src_g + K);
  ok = ok && (fabs(d) > 0.0) && (fabs(d) < INFINITY);
  const double inv = rcp_f64(d);
  // T(k, j) for this lane's column j, in every row
  const double rowk = SW == 2 ? bcast_from_row<src_g>(rowreg) : bpermute_f64((16 * src_g + j) * 4, rowreg);
  const double rk = (j == K) ? -inv : rowk * inv;
  double f[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)  // T(g + 4t, k): column k of this lane's own rows ( = T(k, g + 4t) by symmetry)
    f[t] = SW == 0 ? bpermute_f64((16 * src_g + g + 4 * t) * 4, rowreg) : row_bcast<K>(T[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const double a0 = (j == K) ? 0.0 : T[t];
    double nv = fma(-f[t], rk, a0);
    if (t == src_t) nv = (g == src_g) ? rk : nv;
    T[t] = nv;
  }
}
// Pad behind an EXEC-masked VALU write inside an asm block.  What the next instruction may need: a DPP read of the VGPR just written
// (2 wait states: the s_mov_b64 that restores EXEC is one, `s_nop 0` the other).  The 5 wait states of "EXEC written -> DPP" apply to
// VALU writes of EXEC (v_cmpx) only; an s_mov to EXEC is interlocked by the hardware (hipcc itself puts DPP ops right behind
// s_or_b64 exec).  Rounds 1-2 padded with `s_nop 4`; A/B builds restore that with -DMO_MASKED_PAD_4.
#ifdef MO_MASKED_PAD_4
#define MO_MASKED_PAD "4"
#else
#define MO_MASKED_PAD "0"
#endif
// 64-bit helpers for the lean sweep (SW == 3): one v_mov_b64_dpp instead of two 32-bit DPP movs, and EXEC-masked v_mov_b64
// instead of v_cndmask_b32 pairs.  The asm blocks restore EXEC and end with the wait states a following DPP op needs.
template <int LANE_IN_ROW> __device__ inline double row_bcast64(double v) {
  return __longlong_as_double(__builtin_amdgcn_mov_dpp(__double_as_longlong(v), 0x150 + LANE_IN_ROW, 0xf, 0xf, false));
}
// The lane masks are compile-time immediates (two s_mov_b32 literals): as SGPR operands hipcc keeps all twenty of them live
// across the elimination, overflows the SGPR file and pays for it in v_readlane / v_writelane VALU work.
template <unsigned long long MASK> __device__ inline void masked_set(double& dst, double src) {  // dst = src in the lanes of MASK
  unsigned long long save;
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_mov_b64 %[d], %[s]\n\t"
               "s_mov_b64 exec, %[sv]\n\ts_nop " MO_MASKED_PAD
               : [d] "+v"(dst), [sv] "=&s"(save)
               : [s] "v"(src), [lo] "i"((unsigned)(MASK & 0xffffffffull)), [hi] "i"((unsigned)(MASK >> 32)));
}
template <unsigned long long MASK> __device__ inline void masked_set_neg(double& dst, double src) {  // dst = -src in the lanes of MASK
  unsigned long long save;  // v_max_f64 with both operands negated: the negation rides on the source modifiers (no xor + mov)
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_max_f64 %[d], -%[s], -%[s]\n\t"
               "s_mov_b64 exec, %[sv]\n\ts_nop " MO_MASKED_PAD
               : [d] "+v"(dst), [sv] "=&s"(save)
               : [s] "v"(src), [lo] "i"((unsigned)(MASK & 0xffffffffull)), [hi] "i"((unsigned)(MASK >> 32)));
}
template <unsigned long long MASK> __device__ inline void masked_zero4(d4& T) {  // T[0..3] = 0 in the lanes of MASK
  unsigned long long save;
  double t0 = T[0], t1 = T[1], t2 = T[2], t3 = T[3];
  asm volatile(
      "s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_mov_b64 %[a], 0\n\tv_mov_b64 %[b], 0\n\t"
      "v_mov_b64 %[c], 0\n\tv_mov_b64 %[d], 0\n\ts_mov_b64 exec, %[sv]\n\ts_nop " MO_MASKED_PAD
      : [a] "+v"(t0), [b] "+v"(t1), [c] "+v"(t2), [d] "+v"(t3), [sv] "=&s"(save)
      : [lo] "i"((unsigned)(MASK & 0xffffffffull)), [hi] "i"((unsigned)(MASK >> 32)));
  T[0] = t0; T[1] = t1; T[2] = t2; T[3] = t3;
}
// `bad` accumulates 0 * (1/d): it turns NaN at a zero (or NaN) pivot and stays 0.0 otherwise -- one FMA per pivot instead
// of a compare whose sixteen scalar results hipcc parks in spill lanes.  (An infinite pivot is not flagged here; it makes the
// tile NaN and surfaces as MO_STATUS_NONFINITE.)
template <int K, bool BPF>
__device__ inline void sweep_step_lean(d4& T, double& bad, int g, int j) {
  constexpr int src_g = K & 3, src_t = K >> 2;
  constexpr unsigned long long mcol = 0x0001000100010001ull << K;           // the four lanes of tile column k (j == K)
  constexpr unsigned long long mrow = 0xFFFFull << (16 * src_g);            // the 16 lanes of the row group that holds row k
  const double rowreg = T[src_t];  // every broadcast below is taken before any register of T is modified
  const double d = readlane_f64(rowreg, 16 * src_g + K);
  double inv = __builtin_amdgcn_rcp(d);
  inv = fma(inv, fma(-d, inv, 1.0), inv);  // one Newton step: |inv d - 1| < 2e-15 (tools/microbench.hip)
  asm volatile("v_fma_f64 %0, %1, 0, %0" : "+v"(bad) : "v"(inv));  // volatile: hipcc would sink sixteen of these to the end
#ifdef MO_SWEEP_VALU_ROW   // A/B builds: the row broadcast on the VALU (four v_permlane swaps) instead of over the LDS crossbar (two ds_bpermute)
  double rk = bcast_from_row<src_g>(rowreg) * inv;
#else
  double rk = bpermute_f64((16 * src_g + j) * 4, rowreg) * inv;  // T(k, j) / d for this lane's column j, in every row
#endif
  masked_set_neg<mcol>(rk, inv);
  double f[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)  // T(g + 4t, k): column k of this lane's own rows ( = T(k, g + 4t) by symmetry when BPF)
    f[t] = BPF ? bpermute_f64(4 * g + (64 * src_g + 16 * t), rowreg) : row_bcast64<K>(T[t]);
  masked_zero4<mcol>(T);
#pragma unroll
  for (int t = 0; t < 4; ++t) T[t] = fma(-f[t], rk, T[t]);
  double rowk_new = T[src_t];
  masked_set<mrow>(rowk_new, rk);
  T[src_t] = rowk_new;
}
// ---- the fused-broadcast sweep (SW == 5) --------------------------------------------------------------------------------------
// v_fmac_f64 has a DPP form on gfx950 (VOP2), and row_newbcast may read the destination register itself (tools/dpp_fmac.hip checks it
// bit for bit):  T[t] <- T[t] - T[t](row lane K) * rk  is ONE instruction where the lean sweep spends a v_mov_b64_dpp and a v_fma_f64.
// The pivot column cannot ride along (its lanes ARE the broadcast source): rk is zeroed there for the fused update and the column is
// scaled by 1/d afterwards under an EXEC mask.  The pivot check moves to the scalar unit: d passes through SGPRs anyway (v_readlane), so
// "zero" and "not finite" are integer tests on its bits, accumulated as a running min / max -- no VALU instruction at all.
template <unsigned long long MASK> __device__ inline void masked_zero1(double& v) {  // v = 0 in the lanes of MASK
  unsigned long long save;
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_mov_b64 %[d], 0\n\t"
               "s_mov_b64 exec, %[sv]\n\ts_nop " MO_MASKED_PAD
               : [d] "+v"(v), [sv] "=&s"(save)
               : [lo] "i"((unsigned)(MASK & 0xffffffffull)), [hi] "i"((unsigned)(MASK >> 32)));
}
template <unsigned long long MASK> __device__ inline void masked_mul4(d4& T, double f) {  // T[0..3] *= f in the lanes of MASK
  unsigned long long save;
  double t0 = T[0], t1 = T[1], t2 = T[2], t3 = T[3];
  asm volatile(
      "s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_mul_f64 %[a], %[a], %[f]\n\tv_mul_f64 %[b], %[b], %[f]\n\t"
      "v_mul_f64 %[c], %[c], %[f]\n\tv_mul_f64 %[d], %[d], %[f]\n\ts_mov_b64 exec, %[sv]\n\ts_nop " MO_MASKED_PAD
      : [a] "+v"(t0), [b] "+v"(t1), [c] "+v"(t2), [d] "+v"(t3), [sv] "=&s"(save)
      : [f] "v"(f), [lo] "i"((unsigned)(MASK & 0xffffffffull)), [hi] "i"((unsigned)(MASK >> 32)));
  T[0] = t0; T[1] = t1; T[2] = t2; T[3] = t3;
}
template <int K> __device__ inline void fmac_bcast4(d4& T, double rk0) {  // T[t] -= T[t](lane K of the row) * rk0, t = 0..3
  double t0 = T[0], t1 = T[1], t2 = T[2], t3 = T[3];
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %[a], -%[a], %[r] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %[b], -%[b], %[r] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %[c], -%[c], %[r] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %[d], -%[d], %[r] row_newbcast:%[k] row_mask:0xf bank_mask:0xf"
      : [a] "+v"(t0), [b] "+v"(t1), [c] "+v"(t2), [d] "+v"(t3)
      : [r] "v"(rk0), [k] "i"(K));
  T[0] = t0; T[1] = t1; T[2] = t2; T[3] = t3;
}
struct PivotBits { unsigned min_mag, max_hi; };  // running min of the pivots' magnitude bits / max of their high words (scalar registers)
template <int K>
__device__ inline void sweep_step_fmac(d4& T, PivotBits& pb, int g, int j) {
  constexpr int src_g = K & 3, src_t = K >> 2;
  constexpr unsigned long long mcol = 0x0001000100010001ull << K;           // the four lanes of tile column k (j == K)
  constexpr unsigned long long mrow = 0xFFFFull << (16 * src_g);            // the 16 lanes of the row group that holds row k
  const double rowreg = T[src_t];  // every broadcast below is taken before any register of T is modified
  const unsigned dlo = (unsigned)__builtin_amdgcn_readlane(__double2loint(rowreg), 16 * src_g + K);
  const unsigned dhi = (unsigned)__builtin_amdgcn_readlane(__double2hiint(rowreg), 16 * src_g + K);
  const double d = __hiloint2double((int)dhi, (int)dlo);
  {  // scalar unit: |d| == 0  <=>  mag == 0;  d not finite  <=>  (hi & 0x7fffffff) >= 0x7ff00000
    const unsigned ahi = dhi & 0x7fffffffu;
    const unsigned mag = ahi | (dlo >> 1) | (dlo & 1u);
    pb.min_mag = mag < pb.min_mag ? mag : pb.min_mag;
    pb.max_hi = ahi > pb.max_hi ? ahi : pb.max_hi;
  }
  double inv = __builtin_amdgcn_rcp(d);
  inv = fma(inv, fma(-d, inv, 1.0), inv);  // one Newton step: |inv d - 1| < 2e-15 (tools/microbench.hip)
  double rk = bpermute_f64((16 * src_g + j) * 4, rowreg) * inv;  // T(k, j) / d for this lane's column j, in every row
  double rk0 = rk;
  masked_zero1<mcol>(rk0);           // the pivot column sits the fused update out
  fmac_bcast4<K>(T, rk0);            // T(i, j) -= T(i, k) T(k, j) / d   (j != k)
  masked_mul4<mcol>(T, inv);         // T(i, k) <- T(i, k) / d
  masked_set_neg<mcol>(rk, inv);     // T(k, k) <- -1 / d
  double rowk_new = T[src_t];
  masked_set<mrow>(rowk_new, rk);    // row k <- T(k, j) / d
  T[src_t] = rowk_new;
}
template <int K, int KEND> struct SweepLoopFmac {
  static __device__ inline void run(d4& T, PivotBits& pb, int npiv, int g, int j) {
    if (K < npiv) sweep_step_fmac<K>(T, pb, g, j);  // wave-uniform
    SweepLoopFmac<K + 1, KEND>::run(T, pb, npiv, g, j);
  }
};
template <int KEND> struct SweepLoopFmac<KEND, KEND> {
  static __device__ inline void run(d4&, PivotBits&, int, int, int) {}
};
template <int K, int KEND, int SW> struct SweepLoop {
  static __device__ inline void run(d4& T, bool& ok, double& bad, int npiv, int g, int j) {
    if (K < npiv) {  // wave-uniform
      if (SW >= 3) sweep_step_lean<K, SW == 4>(T, bad, g, j);
      else sweep_step<K, SW>(T, ok, g, j);
    }
    SweepLoop<K + 1, KEND, SW>::run(T, ok, bad, npiv, g, j);
  }
};
template <int KEND, int SW> struct SweepLoop<KEND, KEND, SW> {
  static __device__ inline void run(d4&, bool&, double&, int, int, int) {}
};
template <int SW> __device__ inline bool sweep_tile(d4& T, int npiv, int g, int j) {
  if (SW == 5) {
    PivotBits pb{0xffffffffu, 0u};
    SweepLoopFmac<0, 16>::run(T, pb, npiv, g, j);
    return pb.min_mag != 0u && pb.max_hi < 0x7ff00000u;
  }
  bool ok = true;
  double bad = 0.0;
  SweepLoop<0, 16, SW>::run(T, ok, bad, npiv, g, j);
  return ok && (bad == 0.0);
}

// ---- reusable components ----------------------------------------------------------------------------------------
// J stream: J (m_r x N, row-major) goes HBM -> LDS ring -> MFMA operand registers exactly once per pass.  Lane (g, j) of
// 4-row group s fetches J(4s+g, 32h+2j .. +1) and later reads the same 16 bytes back, so the ring needs no layout: it is a
// per-lane FIFO that costs no VGPRs.  The source addresses are scalar bases (bumped on the SALU) plus one constant per-lane
// offset, and the ring slots are compile-time, so the per-group VALU work is the four J^T r FMAs and nothing else.
// FLAT (odd n: rows of J are not 16-byte aligned, a 16-byte piece would straddle two rows): the 4-row group is copied as ONE flat
// run of 32 nn bytes by dword DMAs (N / 8 instructions, the ones beyond the run parked on a trash word so that the hand-counted
// vmcnt stays exact) and each lane picks its operands out of the slot with 8-byte LDS reads, masking the columns >= nn.
// GATHER (JMODE 2: any other layout of J -- column-major, a leading dimension beyond n, rows that are only 8-byte aligned): every lane
// fetches its own operands J(4s+g, col) as two dwords each through per-lane LDS-DMA addresses (wave-uniform part of the address on the
// scalar unit: base + column part; one VGPR holds the lane part g * row_stride + 2j * column_stride).  2 NT + 1 DMAs per group, operand c
// at dwords [2c][lane] and [2c+1][lane] of the slot.  The memory system sees 4-byte pieces: this is the flexible path, not the fast one.
constexpr int JMODE_VECTOR = 0, JMODE_FLAT = 1, JMODE_GATHER = 2;
template <int NT, int D, int JMODE = JMODE_VECTOR, int NY = 1, bool ROW0 = true>
struct JStream {
  static constexpr bool FLAT = JMODE == JMODE_FLAT, GATHER = JMODE == JMODE_GATHER;
  static constexpr int N = 16 * NT, NB = NT + NY, NH = NT / 2, NI = GATHER ? 2 * NT : (FLAT ? N / 8 : NH), DPS = NI + 1, SLOT = NH * 1024 + 64;
  static_assert((D - 1) * DPS <= 63, "vmcnt is a 6-bit counter");
  static_assert(D >= 2 && D <= 8, "ring depth");
  const char* jbase;   // wave-uniform: row 4s of J
  const char* rbase;   // wave-uniform: r + 4s
  unsigned joff, roff; // per-lane byte offsets inside a 4-row group / inside r[4s .. 4s+3]
  unsigned jstep;      // bytes per 4-row group (4 nn doubles)
  const double* tail_j; const double* tail_r; int rem, row_len, g_, j_;  // the m_r % 4 rows left over after the last full group
  bool act0, act1, act2, act3;  // does this lane's piece h lie inside the row (nn < N pads the system; the ring is zeroed once)
  const char* lane_piece;
  const char* r_elem;
  unsigned ring_base;
  int lane, nsteps;
  long long rs_, cs_;  // GATHER: row / column stride of J in bytes (wave-uniform)

  // rs / cs: element strides between rows / columns of J (row-major packed: nn, 1).  Only the GATHER stream looks at them.
  __device__ inline void init(const double* Jp, const double* rg, const char* smem, unsigned ring_base_, int lane_, int g, int j, int m_r,
                              int nn = N, long long rs = -1, long long cs = 1) {
    if (rs < 0) rs = nn;
    jbase = reinterpret_cast<const char*>(Jp);
    rbase = reinterpret_cast<const char*>(rg);
    rs_ = 8 * rs; cs_ = 8 * cs;
    joff = GATHER ? (unsigned)(g * rs_ + 2 * j * cs_) : (unsigned)(g * nn + 2 * j) * 8u;
    jstep = GATHER ? (unsigned)(4 * rs_) : 32u * (unsigned)nn;
    act0 = 2 * j < nn; act1 = 32 + 2 * j < nn; act2 = 64 + 2 * j < nn; act3 = 96 + 2 * j < nn;
    roff = 4u * (unsigned)lane_;                             // lanes 0..7 fetch the eight dwords of r[4s .. 4s+3] (no 16-byte alignment)
    lane_piece = smem + lane_ * 16;                          // this lane's 16 bytes inside a 1 KiB DMA piece
    r_elem = smem + NH * 1024 + 8 * g;                       // r[4s + g] inside a slot
    ring_base = ring_base_; lane = lane_; nsteps = m_r >> 2;
    rem = m_r & 3; row_len = nn; g_ = g; j_ = j;
    tail_j = Jp + (size_t)(4 * nsteps) * (GATHER ? rs : (long long)nn); tail_r = rg + 4 * nsteps;
  }
  template <int IMM> __device__ static inline void dma4_si(const void* sbase_in, unsigned voff, unsigned lds_dst_in) {  // dma4_s + instruction offset
    const void* sbase = uniform_ptr(sbase_in);
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_in);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst), "i"(IMM)
        : "memory");
  }
  template <int SL> __device__ inline void issue() {  // DMAs of the next not-yet-issued 4-row group into ring slot SL
    const unsigned dst = ring_base + SL * SLOT;
    if (GATHER) {
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const int col0 = 32 * (c >> 1) + (c & 1);                 // column of lane j = 0 (wave-uniform)
        if (col0 < row_len) {                                     // lane 0 is inside: lanes beyond the row sit the DMA out, the ring holds zeros there
          const char* sb = jbase + (long long)col0 * cs_;
          if (col0 + 2 * j_ < row_len) { dma4_si<0>(sb, joff, dst + 512 * c); dma4_si<4>(sb, joff, dst + 512 * c + 256 - 4); }
        } else {                                                  // operand wholly outside the row: keep the DMA count constant
          if (lane == 0) { dma4_si<0>(jbase, 0u, dst + NH * 1024 + 48); dma4_si<0>(jbase, 0u, dst + NH * 1024 + 48); }
        }
      }
      if (lane < 8) dma4_s(rbase, roff, dst + NH * 1024);
      jbase += jstep;
      rbase += 32;
      return;
    }
    if (FLAT) {
      const int run = 32 * row_len;  // bytes of the group
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (256 * i < run) {         // wave-uniform; lane 0 is always inside
          if (256 * i + 4 * lane < run) dma4_s(jbase + 256 * i, 4u * (unsigned)lane, dst + 256 * i);
        } else {
          if (lane == 0) dma4_s(jbase, 0u, dst + NH * 1024 + 48);  // keeps the DMA count per group constant; lands on a trash word
        }
      }
      if (lane < 8) dma4_s(rbase, roff, dst + NH * 1024);
      jbase += jstep;
      rbase += 32;
      return;
    }
    if (act0) dma16_s<0>(jbase, joff, dst);                            // at least lane 0 of every piece is inside the row
    if (NH > 1 && act1) dma16_s<256>(jbase, joff, dst + 1024 - 256);  // the instruction offset also advances the LDS address
    if (NH > 2 && act2) dma16_s<512>(jbase, joff, dst + 2048 - 512);
    if (NH > 3 && act3) dma16_s<768>(jbase, joff, dst + 3072 - 768);
    if (lane < 8) dma4_s(rbase, roff, dst + NH * 1024);
    jbase += jstep;
    rbase += 32;
  }
  __device__ inline void wait_for_oldest(int younger) const {  // `younger` groups (DPS DMAs each) may stay in flight
    if (younger >= D - 1) { wait_vmcnt<(D - 1) * DPS>(); return; }
    switch (younger) {  // tail of the stream (wave-uniform)
      case 0: wait_vmcnt<0>(); break;
      case 1: wait_vmcnt<1 * DPS>(); break;
      case 2: wait_vmcnt<(D > 2 ? 2 : 0) * DPS>(); break;
      case 3: wait_vmcnt<(D > 3 ? 3 : 0) * DPS>(); break;
      case 4: wait_vmcnt<(D > 4 ? 4 : 0) * DPS>(); break;
      case 5: wait_vmcnt<(D > 5 ? 5 : 0) * DPS>(); break;
      default: wait_vmcnt<(D > 6 ? 6 : 0) * DPS>(); break;
    }
  }
  double rsq;  // sum of r[4s + g]^2 seen by this lane (accumulated only by run<true>: the standalone linearisation wants 0.5 |r|^2)
  template <int SL, bool RSQ = false> __device__ inline void consume(int q, d4 (&U)[NB * NB], double (&cpart)[NT]) {  // group q sits in slot SL
    wait_for_oldest(nsteps - 1 - q);
    double ops[NT];
    if (GATHER) {
      const int* src = reinterpret_cast<const int*>(lane_piece - lane * 16 + SL * SLOT) + lane;
#pragma unroll
      for (int c = 0; c < NT; ++c) ops[c] = __hiloint2double(src[128 * c + 64], src[128 * c]);
    } else if (FLAT) {
      const double* src = reinterpret_cast<const double*>(lane_piece - lane * 16 + SL * SLOT) + g_ * row_len + 2 * j_;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const double v0 = src[32 * h], v1 = src[32 * h + 1];
        ops[2 * h] = (32 * h + 2 * j_ < row_len) ? v0 : 0.0;
        ops[2 * h + 1] = (32 * h + 2 * j_ + 1 < row_len) ? v1 : 0.0;
      }
    } else {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const d2 v = *(const d2*)(lane_piece + SL * SLOT + h * 1024);
        ops[2 * h] = v[0]; ops[2 * h + 1] = v[1];
      }
    }
    const double rq = *(const double*)(r_elem + SL * SLOT);
    lds_fence();  // the slot's bytes are in registers before the slot is handed back to the DMA engine
    if (q + D < nsteps) issue<SL>();
    if (RSQ) rsq = fma(rq, rq, rsq);
#pragma unroll
    for (int ta = 0; ta < NT; ++ta) {
      cpart[ta] = fma(ops[ta], rq, cpart[ta]);
#pragma unroll
      for (int tb = ta; tb < NT; ++tb) jtj_mfma(U[ta * NB + tb], ops[ta], ops[tb], ta);
    }
  }
  // One J^T J MFMA.  On the 128 grid the 36 accumulator tiles (288 registers) exceed the 256 AGPRs, and with one wave per SIMD hipcc selects
  // the AGPR form for every MFMA result: it shuttled five tiles through v_accvgpr_write / read around their MFMAs in every 4-row group (320
  // moves per 144 MFMAs).  Tile row 0 is never an MFMA result again after this stream (block step 0 only reads it), so its eight MFMAs are
  // written as inline assembly with a VGPR accumulator: 28 tiles stay in AGPRs, 8 in VGPRs, no moves.  The compiler cannot see the MFMA
  // inside the asm, so the wait states it inserts around MFMAs are ours to supply: `s_nop 3` in front (VALU write of an operand or of EXEC
  // -> MFMA read) and finish() behind the stream (16-pass DGEMM result -> VALU / LDS / memory access: 18).  What cannot be supplied from
  // here are wait states in front of an access the COMPILER places between two of these MFMAs -- a spill of a row-0 tile above all.  The
  // kernels with a second y tile on this grid (55 tiles, heavy spilling) did exactly that and their Solve differed from run to run, so
  // they keep the builtin; for the others tools/isa_lint.py checks every listing: no instruction but an MFMA may touch the destination
  // registers of a VGPR-form MFMA inside the blocks that hold one.
#ifdef MO_ROW0_ALL_NY   // diagnostic: also with a second y tile (what tools/isa_lint.py flags there is the reason it is off)
  static constexpr bool ROW0_VGPR = NT == 8 && ROW0;
#elif !defined(MO_NO_ROW0_VGPR)
  static constexpr bool ROW0_VGPR = NT == 8 && NY == 1 && ROW0;
#else
  static constexpr bool ROW0_VGPR = false;
#endif
#ifndef MO_ROW0_ASM_QUAL
#define MO_ROW0_ASM_QUAL
#endif
#ifndef MO_ROW0_PREFIX
#define MO_ROW0_PREFIX "s_nop 3\n\t"
#endif
  __device__ static inline void jtj_mfma(d4& acc, double a, double b, int ta) {
    if (ROW0_VGPR && ta == 0) asm MO_ROW0_ASM_QUAL (MO_ROW0_PREFIX "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  __device__ static inline void finish() {
    if (ROW0_VGPR) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // 16-pass DGEMM result -> VALU read / write: 18 wait states at most
  }
#define MO_FOR_SLOTS(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
  __device__ inline void prologue() {  // fill the ring
#define MO_ISSUE(u) if (u < D && u < nsteps) issue<(u < D ? u : 0)>();
    MO_FOR_SLOTS(MO_ISSUE)
#undef MO_ISSUE
  }
  template <bool RSQ = false>
  __device__ inline void run(d4 (&U)[NB * NB], double (&cpart)[NT]) {  // G tiles += J^T J, cpart += J^T r partials
    if (RSQ) rsq = 0.0;
    for (int q0 = 0; q0 < nsteps; q0 += D) {
#define MO_CONSUME(u) if (u < D && q0 + u < nsteps) consume<(u < D ? u : 0), RSQ>(q0 + u, U, cpart);
      MO_FOR_SLOTS(MO_CONSUME)
#undef MO_CONSUME
    }
    if (rem) {  // wave-uniform: up to three rows that do not fill a 4-row group -- plain loads, lanes of the missing rows feed zeros
      double ops[NT];
#pragma unroll
      for (int c = 0; c < NT; ++c) ops[c] = 0.0;
      double rq = 0.0;
      if (g_ < rem) {
        if (GATHER) {
          const char* row = reinterpret_cast<const char*>(tail_j) + g_ * rs_ + 2 * j_ * cs_;
#pragma unroll
          for (int c = 0; c < NT; ++c) {
            const int col0 = 32 * (c >> 1) + (c & 1);
            if (col0 + 2 * j_ < row_len) ops[c] = *reinterpret_cast<const double*>(row + (long long)col0 * cs_);
          }
        } else {
          const double* row = tail_j + (size_t)g_ * row_len + 2 * j_;
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            if (32 * h + 2 * j_ < row_len) ops[2 * h] = row[32 * h];
            if (32 * h + 2 * j_ + 1 < row_len) ops[2 * h + 1] = row[32 * h + 1];
          }
        }
        rq = tail_r[g_];
      }
      if (RSQ) rsq = fma(rq, rq, rsq);
#pragma unroll
      for (int ta = 0; ta < NT; ++ta) {
        cpart[ta] = fma(ops[ta], rq, cpart[ta]);
#pragma unroll
        for (int tb = ta; tb < NT; ++tb) jtj_mfma(U[ta * NB + tb], ops[ta], ops[tb], ta);
      }
    }
    finish();
  }
#undef MO_FOR_SLOTS
};

// Block LDL^T with 16x16 pivot blocks over the (NT+1) x (NT+1) upper block triangle of tiles (the last block column is
// [A_eq^T | rhs]).  Afterwards the diagonal tiles hold -T^-1, the off-diagonal tiles their forward-eliminated values.
// Pivots of diagonal tile pa: 16 for the x tiles and for every y tile but the last, which holds the remaining k - 16 (NY - 1) equality rows.
template <int NT, int NY> __device__ inline int tile_pivots(int pa, int k) { return pa < NT + NY - 1 ? 16 : k - 16 * (NY - 1); }
template <int NT, int SW, int NY = 1>
__device__ inline bool block_eliminate(d4 (&U)[(NT + NY) * (NT + NY)], int k, int g, int j) {
  constexpr int NB = NT + NY;
  bool ok = true;
#pragma unroll
  for (int pa = 0; pa < NB; ++pa) {
    ok = sweep_tile<SW>(U[pa * NB + pa], tile_pivots<NT, NY>(pa, k), g, j) && ok;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pc = pa + 1; pc < NB; ++pc) {
      d4 negZ = mfma4(U[pa * NB + pa], U[pa * NB + pc], d4{0.0, 0.0, 0.0, 0.0});  // (-T^-1) U_ac  (T^-1 is symmetric)
#pragma unroll
      for (int pb = pa + 1; pb <= pc; ++pb) U[pb * NB + pc] = mfma4(U[pa * NB + pb], negZ, U[pb * NB + pc]);
      __builtin_amdgcn_sched_barrier(0);  // one panel tile at a time: keeps a single -Z tile live (register pressure)
    }
  }
  return ok;
}

#ifndef MO_LA_MIN_NT
#define MO_LA_MIN_NT 6   // smallest tile grid (NT = n / 16) whose step kernel eliminates with look-ahead (A/B knob; 99 = only where SW == 6 asks)
#endif
// ---- look-ahead elimination (SW == 6) ------------------------------------------------------------------------------------------
// The diagonal sweeps are dependent VALU chains, the trailing updates independent MFMA chains; in program order "sweep, then all updates"
// a wave alternates between a phase that can only wait on itself and a phase that only feeds the matrix pipe.  Here the update of block
// step pa is split: the column of tile (pa+1, pa+1) first, then the sweep of that tile INTERLEAVED, pivot by pivot, with the remaining
// updates of step pa (they touch neither that tile nor its operands) -- the classic look-ahead of dense factorisations, inside one
// wavefront.  Same arithmetic, same order of operations per tile: results are bit-identical to block_eliminate.
template <int NB> constexpr int la_count(int pa) { int c = 0; for (int pc = pa + 2; pc < NB; ++pc) c += 1 + (pc - pa); return c; }
template <int NB> constexpr int la_pc(int pa, int idx) {
  for (int pc = pa + 2; pc < NB; ++pc) { const int n = 1 + (pc - pa); if (idx < n) return pc; idx -= n; }
  return -1;
}
template <int NB> constexpr int la_sub(int pa, int idx) {  // 0: the panel product -Z = (-T^-1) U_ac;  s >= 1: update of tile (pa + s, pc)
  for (int pc = pa + 2; pc < NB; ++pc) { const int n = 1 + (pc - pa); if (idx < n) return idx; idx -= n; }
  return -1;
}
// The Q-th of the four MFMAs of work item IDX of block step PA (an item = one 16x16x16 tile product = mfma4).
template <int NT, int PA, int IDX, int Q> __device__ inline void la_mfma(d4 (&U)[(NT + 1) * (NT + 1)], d4& negZ) {
  constexpr int NB = NT + 1;
  if constexpr (IDX < la_count<NB>(PA)) {
    constexpr int pc = la_pc<NB>(PA, IDX), sub = la_sub<NB>(PA, IDX);
    if constexpr (sub == 0) {
      if constexpr (Q == 0) negZ = d4{0.0, 0.0, 0.0, 0.0};
      negZ = __builtin_amdgcn_mfma_f64_16x16x4f64(U[PA * NB + PA][Q], U[PA * NB + pc][Q], negZ, 0, 0, 0);
    } else {
      U[(PA + sub) * NB + pc] = __builtin_amdgcn_mfma_f64_16x16x4f64(U[PA * NB + (PA + sub)][Q], negZ[Q], U[(PA + sub) * NB + pc], 0, 0, 0);
    }
  }
}
// Work items per pivot of block step PA: one on the 32 / 64 grids (at most 12 items for 16 pivots), up to three on the 96 / 128 grids.
template <int NB> constexpr int la_per_pivot(int pa) { return (la_count<NB>(pa) + 15) / 16; }
// What goes to wait point P (0 .. 3) of pivot K: with one item per pivot its P-th MFMA; with M > 1 the whole item K M + (its slot at P) --
// items stay in order (all four MFMAs of an item before the next one: the updates of a panel need its finished -Z, the next panel product
// overwrites it).
template <int NT, int PA, int K, int P> __device__ inline void la_point(d4 (&U)[(NT + 1) * (NT + 1)], d4& negZ) {
  constexpr int M = la_per_pivot<NT + 1>(PA);
  if constexpr (M <= 1) {
    la_mfma<NT, PA, K, P>(U, negZ);
  } else {
    constexpr int slot = M == 2 ? (P == 0 ? 0 : (P == 2 ? 1 : -1)) : (P < M ? P : -1);
    if constexpr (slot >= 0) {
      la_mfma<NT, PA, K * M + slot, 0>(U, negZ); la_mfma<NT, PA, K * M + slot, 1>(U, negZ);
      la_mfma<NT, PA, K * M + slot, 2>(U, negZ); la_mfma<NT, PA, K * M + slot, 3>(U, negZ);
    }
  }
}
// sweep_step_lean with the four MFMAs of one work item placed at the points where its dependent chain waits (in-order issue: an MFMA
// behind the whole step would wait with it): behind the reciprocal's issue, behind the row broadcast's LDS request, behind the column
// broadcasts, behind the rank-1 update.
template <int NT, int PA, int K>
__device__ inline void sweep_step_work(d4 (&U)[(NT + 1) * (NT + 1)], d4& negZ, double& bad, bool active, int g, int j) {
  constexpr int NB = NT + 1;
  constexpr int src_g = K & 3, src_t = K >> 2;
  constexpr unsigned long long mcol = 0x0001000100010001ull << K;
  constexpr unsigned long long mrow = 0xFFFFull << (16 * src_g);
  d4& T = U[(PA + 1) * NB + (PA + 1)];
  if (!active) {  // wave-uniform: pivots beyond the y tile's k rows -- only the work
    la_point<NT, PA, K, 0>(U, negZ); la_point<NT, PA, K, 1>(U, negZ); la_point<NT, PA, K, 2>(U, negZ); la_point<NT, PA, K, 3>(U, negZ);
    return;
  }
  const double rowreg = T[src_t];
  const double d = readlane_f64(rowreg, 16 * src_g + K);
  double inv = __builtin_amdgcn_rcp(d);
  const double rowk = bpermute_f64((16 * src_g + j) * 4, rowreg);
  __builtin_amdgcn_sched_barrier(0);
  la_point<NT, PA, K, 0>(U, negZ);                       // covers v_rcp_f64 and the LDS round trip of the row broadcast
  __builtin_amdgcn_sched_barrier(0);
  inv = fma(inv, fma(-d, inv, 1.0), inv);
  asm volatile("v_fma_f64 %0, %1, 0, %0" : "+v"(bad) : "v"(inv));
  double f[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) f[t] = row_bcast64<K>(T[t]);
  __builtin_amdgcn_sched_barrier(0);
  la_point<NT, PA, K, 1>(U, negZ);
  __builtin_amdgcn_sched_barrier(0);
  double rk = rowk * inv;
  masked_set_neg<mcol>(rk, inv);
  masked_zero4<mcol>(T);
  __builtin_amdgcn_sched_barrier(0);
  la_point<NT, PA, K, 2>(U, negZ);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) T[t] = fma(-f[t], rk, T[t]);
  double rowk_new = T[src_t];
  masked_set<mrow>(rowk_new, rk);
  T[src_t] = rowk_new;
  __builtin_amdgcn_sched_barrier(0);
  la_point<NT, PA, K, 3>(U, negZ);
  __builtin_amdgcn_sched_barrier(0);
}
template <int NT, int PA, int K> struct SweepWithWork {
  static __device__ inline void run(d4 (&U)[(NT + 1) * (NT + 1)], d4& negZ, double& bad, int npiv, int g, int j) {
    sweep_step_work<NT, PA, K>(U, negZ, bad, K < npiv, g, j);
    SweepWithWork<NT, PA, K + 1>::run(U, negZ, bad, npiv, g, j);
  }
};
template <int NT, int PA> struct SweepWithWork<NT, PA, 16> {
  static __device__ inline void run(d4 (&)[(NT + 1) * (NT + 1)], d4&, double&, int, int, int) {
    static_assert(la_count<NT + 1>(PA) <= 16 * la_per_pivot<NT + 1>(PA) && la_per_pivot<NT + 1>(PA) <= 4, "work items per pivot");
  }
};
template <int NT, int PA> struct LookAheadSteps {
  static __device__ inline void run(d4 (&U)[(NT + 1) * (NT + 1)], double& bad, int k, int g, int j) {
    constexpr int NB = NT + 1;
    // the column of the next diagonal tile first
    d4 negZ = mfma4(U[PA * NB + PA], U[PA * NB + PA + 1], d4{0.0, 0.0, 0.0, 0.0});
    U[(PA + 1) * NB + PA + 1] = mfma4(U[PA * NB + PA + 1], negZ, U[(PA + 1) * NB + PA + 1]);
    __builtin_amdgcn_sched_barrier(0);
    SweepWithWork<NT, PA, 0>::run(U, negZ, bad, PA + 1 < NT ? 16 : k, g, j);
    if constexpr (PA + 2 < NB) LookAheadSteps<NT, PA + 1>::run(U, bad, k, g, j);
  }
};
template <int NT>
__device__ inline bool block_eliminate_lookahead(d4 (&U)[(NT + 1) * (NT + 1)], int k, int g, int j) {
  const bool ok0 = sweep_tile<3>(U[0], 16, g, j);
  __builtin_amdgcn_sched_barrier(0);
  double bad = 0.0;
  LookAheadSteps<NT, 0>::run(U, bad, k, g, j);
  return ok0 && (bad == 0.0);
}

// Backward substitution after block_eliminate; xb[c] = solution at permuted position 16c + j (replicated over g),
// xb[NT + q] = the solution of y tile q in lanes 16 q + j < k.
template <int NT, int NY = 1>
__device__ inline void back_substitute(const d4 (&U)[(NT + NY) * (NT + NY)], int k, int j, double (&xb)[NT + NY]) {
  constexpr int NB = NT + NY, L = NB - 1;
  {
    double v = 0.0;  // sits in column kRC of the swept LAST y tile: element (q, kRC) at lane (q & 3, kRC), register q >> 2
    const int src = (16 * (j & 3) + kRC) * 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double w = bpermute_f64(src, U[L * NB + L][t]);
      if ((j >> 2) == t) v = w;
    }
    xb[L] = (16 * (NY - 1) + j < k) ? v : 0.0;
  }
#pragma unroll
  for (int pa = L - 1; pa >= 0; --pa) {
    double vt[4], pt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pt[t] = 0.0;
#pragma unroll
      for (int pb = pa + 1; pb < NB; ++pb) pt[t] = fma(U[pa * NB + pb][t], xb[pb], pt[t]);
    }
    // the four row sums (over the row's 16 lanes = the columns of the tile row) as ONE transposed butterfly: lane j gets the sum of pt[j & 3],
    // four row broadcasts hand every lane all four -- 31 VALU instructions where four plain butterflies take 48
    const double ws = row_sum4_scatter(pt[0], pt[1], pt[2], pt[3]);
    vt[0] = row_bcast64<kRC>(U[pa * NB + L][0]) - row_bcast64<0>(ws);   // forward-eliminated rhs minus the already solved blocks
    vt[1] = row_bcast64<kRC>(U[pa * NB + L][1]) - row_bcast64<1>(ws);
    vt[2] = row_bcast64<kRC>(U[pa * NB + L][2]) - row_bcast64<2>(ws);
    vt[3] = row_bcast64<kRC>(U[pa * NB + L][3]) - row_bcast64<3>(ws);
    double q = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) q = fma(U[pa * NB + pa][t], vt[t], q);  // (-T^-1) v, summed over this lane's 4 rows
    xb[pa] = -cross_row_sum(q);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// A SECOND right-hand side through the factors block_eliminate left in the tiles (diagonal tiles: -T^-1, row panels: their
// forward-eliminated values) -- what Mehrotra's corrector needs: same matrix, new rhs.  Vectors are V16 (value at lane j, replicated
// over g); the row-layout copies a tile product needs go through a 16-double LDS hop.  rb[] is consumed; the forward-eliminated
// blocks are parked in rbuf_x / rbuf_y for the substitution.  xb[c] = solution at permuted position 16c + j, xb[NT] = the y block.
template <int NT, int NY = 1>
__device__ inline void solve_second_rhs(const d4 (&U)[(NT + NY) * (NT + NY)], int k, int g, int j, double (&rb)[NT + NY], double* hop,
                                        double* rbuf_x, double* rbuf_y, double (&xb)[NT + NY]) {
  constexpr int NB = NT + NY, L = NB - 1;
  // lanes beyond the k equalities carry the first solve's leftovers in the tiles: keep them out
#pragma unroll
  for (int q = 0; q < NY; ++q) rb[NT + q] = (16 * q + j < k) ? rb[NT + q] : 0.0;
#pragma unroll
  for (int pa = 0; pa < NB; ++pa) {  // forward: r_b += U_ab^T (-T_a^-1 r_a), b > a
    if (pa >= NT) rb[pa] = (16 * (pa - NT) + j < k) ? rb[pa] : 0.0;
    if (g == 0) { hop[j] = rb[pa]; (pa < NT ? rbuf_x + 16 * pa : rbuf_y + 16 * (pa - NT))[j] = rb[pa]; }
    lds_fence();
    if (pa < L) {
      double q = 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) q = fma(U[pa * NB + pa][t], hop[g + 4 * t], q);
      const double w = cross_row_sum(q);  // (-T_a^-1 r_a)(j)
      lds_fence();                        // hop has been read
      if (g == 0) hop[j] = w;
      lds_fence();
      double wr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) wr[t] = hop[g + 4 * t];
#pragma unroll
      for (int pb = pa + 1; pb < NB; ++pb) {
        double q2 = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) q2 = fma(U[pa * NB + pb][t], wr[t], q2);
        rb[pb] += cross_row_sum(q2);
        if constexpr (NY >= 2) asm volatile("" : "+v"(rb[pb]));   // one cross-row sum at a time (register pressure: 168 tile registers)
      }
      lds_fence();                        // hop has been read before the next block overwrites it
    }
  }
#pragma unroll
  for (int pa = L; pa >= 0; --pa) {  // backward, as back_substitute() but with the rhs read from LDS
    const double* rsrc = pa < NT ? rbuf_x + 16 * pa : rbuf_y + 16 * (pa - NT);
    double vt[4], pt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pt[t] = 0.0;
#pragma unroll
      for (int pb = pa + 1; pb < NB; ++pb) pt[t] = fma(U[pa * NB + pb][t], xb[pb], pt[t]);
    }
    if (pa < L) {   // (as in back_substitute: one transposed butterfly for the four row sums)
      const double ws = row_sum4_scatter(pt[0], pt[1], pt[2], pt[3]);
      pt[0] = row_bcast64<0>(ws); pt[1] = row_bcast64<1>(ws); pt[2] = row_bcast64<2>(ws); pt[3] = row_bcast64<3>(ws);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) vt[t] = rsrc[g + 4 * t] - pt[t];
    double q = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) q = fma(U[pa * NB + pa][t], vt[t], q);
    xb[pa] = -cross_row_sum(q);
    if (pa >= NT) xb[pa] = (16 * (pa - NT) + j < k) ? xb[pa] : 0.0;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the kernel ------------------------------------------------------------------------------------------------
// NT = n / 16 (2 or 4), WPS = waves per SIMD the register budget is sized for.  k <= 15 (index 15 of the y tile carries the right-hand side), m <= 64 MC are
// checked by fused_supported().
// MC = constraint slots per lane (m <= 64 MC).
template <int NT, int WPS, int MC = 1, int NY = 1> struct FusedCfg {
  static constexpr int N = 16 * NT;
  static constexpr int NH = NT / 2;                 // 16-byte J loads per lane per 4-row group
  static constexpr int DPS = NH + 1;                // LDS-DMA instructions per 4-row group (J pieces + 32 B of r)
  static constexpr int SLOT = NH * 1024 + 64;       // ring slot: 4 rows of J (lane-linear) + r[4s..4s+3]
  static constexpr int MCAP = 64 * MC;
  static constexpr int VEC = (3 * N + 4 * MCAP + MCAP / 2 + 32 * NY) * 8;  // xs, diagS|rp, rhsS|dxs, cons a/b/s/z, cons var (int), y[16 NY], b_eq[16 NY]
  static constexpr int D_FIT = ((160 * 1024) / (4 * WPS) - VEC) / SLOT;  // what the 160 KiB of a CU leave per wave
  static constexpr int D_TUNED = NT > 4 ? 4 : (NT == 4 ? (WPS >= 3 ? 4 : 7) : (WPS >= 4 ? 4 : 8));
  static constexpr int D = MC == 1 ? D_TUNED : (D_FIT > 8 ? 8 : D_FIT);  // ring depth (4-row groups in flight per wave), LDS-limited
  static_assert(D >= 2, "LDS budget");
  static constexpr int LDS = D * SLOT + VEC;
};

// One workgroup of 4*WPS independent waves per CU (so that exactly WPS waves sit on every SIMD).  The waves never
// synchronise with each other; each owns its slice of the workgroup's LDS.
template <int NT, int WPS, int SW, bool QPL, int MC = 1, int JMODE = JMODE_VECTOR, int NY = 1>
__global__ __launch_bounds__(256 * WPS, WPS) void kkt_fused_f64_kernel(const KernelArgs a) {
  using C = FusedCfg<NT, WPS, MC, NY>;
  constexpr int MCAP = C::MCAP;
  constexpr int N = C::N, NB = NT + NY, LT = NB - 1, SLOT = C::SLOT, D = C::D;  // LT: the tile column that carries the right-hand side
  constexpr int WAVES = 4 * WPS;

  // ONE shared array (a second __shared__ object beside an LDS-DMA target makes hipcc drain vmcnt before LDS reads)
  __shared__ __attribute__((aligned(16))) char smem_all[WAVES * C::LDS];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* const smem = smem_all + wave * C::LDS;
  double* const xs = reinterpret_cast<double*>(smem + D * SLOT);  // x, natural order
  double* const diagS = xs + N;                                   // barrier diagonal per variable, natural order
  double* const rhsS = diagS + N;                                 // inequality part of the rhs per variable, natural order
  double* const rp = diagS;                                       // right-hand side, permuted order (diagS is dead by then)
  double* const dxs = rhsS;                                       // dx, natural order (rhsS is dead by then)
  double* const cA = rhsS + N;                                    // per-constraint a, b, s, z and variable index; y; b_eq:
  double* const cB = cA + MCAP;                                   //   LDS-DMA targets of P0 (the small per-problem vectors
  double* const cS = cB + MCAP;                                   //   cost no VGPRs while J streams)
  double* const cZ = cS + MCAP;
  int* const cV = reinterpret_cast<int*>(cZ + MCAP);
  double* const yb = cZ + MCAP + MCAP / 2;
  double* const bb = yb + 16 * NY;
  const unsigned ring_base = (unsigned)(uintptr_t)smem;           // LDS byte address of the ring (low 32 bits of the flat address)
  const unsigned vec_base = ring_base + D * SLOT;                 // LDS byte address of xs

  // MO_STEP_NO_INEQUALITIES (SolveForUpdateNoInequalities, qp.cc:366-386): the constraints take no part (m = 0 below); the state and
  // direction vectors keep their [x | s(m_lay) | y | z(m_lay)] layout, ds = dz = 0 and both step lengths are 1.
  const int m_lay = a.m;
  const bool no_ineq = (a.flags & MO_STEP_NO_INEQUALITIES) != 0;
  const int k = a.k, m = no_ineq ? 0 : a.m, m_r = a.m_r;
  const int nn = a.n;  // actual number of variables <= N; the system is padded to whole tiles (unit diagonal, zero solution)
  // Once per wave: the ring (lanes whose J piece lies beyond the row never receive DMA data and must read zeros) and x's padding.
  for (int i = (int)(threadIdx.x & 63); i < D * SLOT / 8; i += 64) reinterpret_cast<double*>(smem)[i] = 0.0;
  for (int i = (int)(threadIdx.x & 63); i < N; i += 64) xs[i] = 0.0;
  lds_fence();
#ifdef MO_FUSED_STAMPS
  unsigned long long stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
  const unsigned long long stamp_t0 = stamp_prev, stamp_rt0 = __builtin_amdgcn_s_memrealtime();
#endif

  // Problems are handed out from a device-wide ticket counter (zeroed on the stream before the launch): the SIMD
  // arbitrates oldest-wave-first and CUs do not run at identical speed, so a static split leaves 12-30 % of the waves
  // idle at the end (measured with tools/phase_timer).  Tickets are taken in guided chunks (up to 8 problems while the
  // queue is long, single problems at the end) because one counter word sustains only ~88 M atomics/s, and the next
  // chunk is requested at the START of the current chunk's last problem, so the atomic's latency hides under the J stream.
  const int chunk_shift = 63 - __builtin_clzll((unsigned long long)gridDim.x * WAVES * 4);  // ~ remaining / (4 waves' worth)
  // Small launches -- at most a.static_rounds problems per wave -- are split STATICALLY, round by round, in slot-major wave order (first one wave on
  // every SIMD of every CU, then the second wave of every SIMD, ...): no ticket at all.  A wave must otherwise wait for a ticket just to
  // learn that nothing is left, and 3 072 waves asking one counter word at ~88 M atomics/s is 35 us -- as long as the whole first round of
  // BASELINE configs[1] (4 096 problems).  A partial round then also lands one wave per SIMD instead of three per SIMD on a third of the CUs.
  const long long waves_all = (long long)gridDim.x * WAVES;
  const bool st_rounds = a.static_rounds > 0 && a.batch <= (long long)a.static_rounds * waves_all;   // wave-uniform
  auto chunk_for = [&](long long observed) -> int {
    if (st_rounds) return 1;
    const long long c = (a.batch - observed) >> chunk_shift;
    return c < 1 ? 1 : (c > 8 ? 8 : (int)c);
  };
  auto take_ticket = [&](int chunk, long long p_now) -> unsigned long long {
    if (st_rounds) return (unsigned long long)p_now;   // static rounds: the next problem of this wave is p_now + waves_all (= "ticket" p_now + ticket_base)
    unsigned long long t = 0;
    if (lane_id() == 0) t = atomicAdd(a.ticket, (unsigned long long)chunk);
    return t;
  };
  auto uniform64 = [](unsigned long long v) -> long long {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  // Start stagger.  Every problem costs the same, so the waves that share a SIMD (waves w, w + 4, w + 8 of the workgroup) would march
  // through the phases in lockstep for the whole launch -- three J streams together, then three dependent pivot chains together.  Delaying
  // the second and third wave of each SIMD once, by about a third of a problem each (a.stagger units of 127 x 64 cycles), keeps one wave
  // in the matrix-bound phase while another is in the sweeps: +1.4 % at cfg 3 (A/B on one box), nothing at cfg 2.
  if (a.stagger > 0 && !st_rounds) {
    const int slot = wave >> 2;
    for (int i = 0; i < slot * a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // The FIRST chunk of every wave is static (wave w of the persistent grid takes problems [w c0, (w + 1) c0)); tickets from the counter start
  // behind that part.  All waves asking one counter word for their first ticket at kernel start costs 3 072 / 88 M atomics/s = 35 us: most
  // of a small launch (BASELINE configs[1]: 4 096 problems) and 2 % of the headline one.
  int chunk = chunk_for(0);
  const long long ticket_base = (long long)gridDim.x * WAVES * chunk;
  long long p = st_rounds ? (long long)(wave >> 2) * ((long long)gridDim.x * 4) + (long long)blockIdx.x * 4 + (wave & 3) : ((long long)blockIdx.x * WAVES + wave) * chunk;
  long long chunk_end = p + chunk;

  while (p < a.batch) {
    const bool last_of_chunk = p + 1 >= chunk_end;  // wave-uniform
    int next_chunk = 0;
    unsigned long long next_ticket = 0;
    if (last_of_chunk) {
      next_chunk = chunk_for(p);
      next_ticket = take_ticket(next_chunk, p);
    }
    // Lane coordinates are made opaque once per problem so that nothing derived from them (gather addresses, masks,
    // bpermute addresses) is hoisted out of the problem loop and kept live for the whole kernel: the tile registers
    // need the room.
    const int lane = lane_id();
    const int g = lane >> 4, j = lane & 15;

    const double* Jp = QPL ? nullptr : (const double*)a.J + p * a.J_stride;
    const double* rg = QPL ? nullptr : (const double*)a.r + p * a.r_stride;
    // Everything but the J stream's own operands is re-read from the kernarg segment where it is used (scalar loads, K$
    // hits): held in SGPRs for the whole kernel they overflow the SGPR file and come back as v_readlane VALU work.
    KArgs ka = fresh_args();
    const double* vp = (const double*)ka->vars + p * ka->vars_stride;

    // The ring is filled FIRST: the J stream's memory latency then overlaps the address arithmetic and the small loads of P0.
    JStream<NT, D, JMODE, NY> stream;
    if (!QPL) {
      stream.init(Jp, rg, smem, ring_base, lane, g, j, m_r, nn, a.J_row_major ? (long long)a.J_ld : 1ll, a.J_row_major ? 1ll : (long long)a.J_ld);
      stream.prologue();
    }
    MO_STAMP(13);   // (diagnostic build) loop top, ticket, addresses, ring fill issued

    // ---- P0: every small vector of this problem goes global -> LDS by DMA, issued behind the ring fill; nothing of it sits in
    //          a VGPR while J streams (the tile registers need the room), and the last wait of the stream covers it.
    d4 U[NB * NB];
#pragma unroll
    for (int q = 0; q < NB * NB; ++q) U[q] = d4{0.0, 0.0, 0.0, 0.0};
    dma_doubles(vp, vec_base, nn, lane);                                                    // x -> xs
    if (m > 0) {
      const long long coff = p * ka->cons_stride;
      if (lane < m) dma4_s(ka->cons_var + coff, 4u * (unsigned)lane, vec_base + (3 * N + 4 * MCAP) * 8);
#pragma unroll
      for (int ci = 1; ci < MC; ++ci)
        if (lane + 64 * ci < m) dma4_s(ka->cons_var + coff + 64 * ci, 4u * (unsigned)lane, vec_base + (3 * N + 4 * MCAP) * 8 + 256 * ci);
      dma_doubles((const double*)ka->cons_a + coff, vec_base + (3 * N) * 8, m, lane);
      dma_doubles((const double*)ka->cons_b + coff, vec_base + (3 * N + MCAP) * 8, m, lane);
      dma_doubles(vp + nn, vec_base + (3 * N + 2 * MCAP) * 8, m, lane);                      // s
      dma_doubles(vp + nn + m + k, vec_base + (3 * N + 3 * MCAP) * 8, m, lane);              // z
    }
    if (k > 0) {
      dma_doubles(vp + nn + m_lay, vec_base + (3 * N + 4 * MCAP + MCAP / 2) * 8, k, lane);   // y
      dma_doubles((const double*)ka->b + p * ka->b_stride, vec_base + (3 * N + 4 * MCAP + MCAP / 2 + 16 * NY) * 8, k, lane);  // b_eq
    }
    MO_STAMP(14);   // (diagnostic build) tile registers zeroed, small-vector DMAs issued
    // tile column NT = [A_eq^T | rhs] (rhs is merged in after P3); y diagonal tile = [0, -b_eq; -b_eq^T, 0]
    load_a_tiles<NT, QPL, NY>(k > 0 ? (const double*)ka->A + p * ka->A_stride : nullptr, ka->A_ld, k, nn, g, j, U);

    MO_STAMP(0);
    // ---- P1: stream J once through the LDS-DMA ring; G = J^T J on the matrix cores (upper block triangle of tiles),
    //          c = J^T r on the VALU.  Lane (g, j) of 4-row group s fetches J(4s+g, 32h+2j .. +1) and later reads the same
    //          16 bytes back, so the ring needs no layout: it is a per-lane FIFO that costs no VGPRs.
    double cpart[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) cpart[c] = 0.0;
    double cvec[NT];  // c (= J^T r) at position 16c + j (replicated over g)
    if (QPL) {
      load_g_tiles<NT, NY>((const double*)ka->G + p * ka->G_stride, ka->G_ld, (const double*)ka->c + p * ka->c_stride, nn, g, j, U, cvec);
    } else {
#ifdef MO_FUSED_STAMPS   // diagnostic build: how long does a wave wait for its FIRST group of J (the latency a cross-problem prefetch could hide)?
      stream.wait_for_oldest(stream.nsteps - 1);
      MO_STAMP(7);
#endif
      stream.run(U, cpart);
#pragma unroll
      for (int c = 0; c < NT; ++c) cvec[c] = cross_row_sum(cpart[c]);
    }
    MO_STAMP(1);

    // ---- P3: per-constraint barrier terms, scattered per variable through LDS (duplicates on one variable accumulate)
    wait_vmcnt<0>();  // the P0 DMAs have landed (they are older than the last ring DMA the stream waited for)
    ka = fresh_args();
    const double mu = ka->mu ? ((const double*)ka->mu)[p * ka->mu_stride] : 0.0;
    if (lane < N / 2) {
      diagS[2 * lane] = 0.0; diagS[2 * lane + 1] = 0.0;
      rhsS[2 * lane] = 0.0; rhsS[2 * lane + 1] = 0.0;
    }
    int cvar[MC]; double ca[MC], cb[MC], cs[MC], cz[MC], cs_inv[MC];  // constraint lane + 64 ci
    bool bad_index[MC];
#pragma unroll
    for (int ci = 0; ci < MC; ++ci) {
      const int ix = lane + 64 * ci;
      cvar[ci] = 0; ca[ci] = 1.0; cb[ci] = 0.0; cs[ci] = 1.0; cz[ci] = 0.0;
      if (ix < m) { cvar[ci] = cV[ix]; ca[ci] = cA[ix]; cb[ci] = cB[ix]; cs[ci] = cS[ix]; cz[ci] = cZ[ix]; }
    }
    lds_fence();  // diagS / rhsS initialised
    bool lane_bad_slack = false, lane_bad_index = false;
#pragma unroll
    for (int ci = 0; ci < MC; ++ci) {
      const int ix = lane + 64 * ci;
      bad_index[ci] = (ix < m) && ((cvar[ci] < 0) || (cvar[ci] >= nn));
      if (bad_index[ci]) cvar[ci] = 0;
      lane_bad_index = lane_bad_index || bad_index[ci];
      lane_bad_slack = lane_bad_slack || ((ix < m) && !(cs[ci] > 0.0));
      cs_inv[ci] = rcp_f64(cs[ci]);                                  // 1/s to ~1 ulp (two Newton steps on v_rcp_f64)
      if (ix < m) {
        const double zs = cz[ci] * cs_inv[ci];
        atomicAdd(&diagS[cvar[ci]], ca[ci] * zs * ca[ci]);                                  // qp.cc:296
        atomicAdd(&rhsS[cvar[ci]], ca[ci] * (cz[ci] * (cs[ci] - cb[ci]) + mu) * cs_inv[ci]);  // x+ form of qp.cc:340-341
      }
    }
    const bool slack_bad = __any(lane_bad_slack);
    bool any_bad_index = __any(lane_bad_index);
    if (no_ineq && m_lay > 0) {  // the index check of Setup (qp.cc:70-72) does not depend on the flag
      bool bad = false;
      const int* cvp = ka->cons_var + p * ka->cons_stride;
      for (int ix = lane; ix < m_lay; ix += 64) { const int v = cvp[ix]; bad = bad || v < 0 || v >= nn; }
      any_bad_index = __any(bad);
    }
    lds_fence();
    double dS[NT], rS[NT];
    ldv<NT, QPL>(diagS, j, dS);
    ldv<NT, QPL>(rhsS, j, rS);
    if (g == 0) {
#pragma unroll
      for (int c = 0; c < NT; ++c) rp[16 * c + j] = rS[c] - cvec[c];
    }
    lds_fence();

    // ---- P2/P4: lambda + Sigma on the diagonal tiles (position (r, r): lanes with j == g + 4t); rhs into column kRC
#pragma unroll
    for (int t = 0; t < 4; ++t) {  // last y diagonal tile = [0, -b_eq; -b_eq^T, 0]: (r, kRC) = -b[r], (kRC, q) = -b[q]
      constexpr int e0 = 16 * (NY - 1);   // first equality row of the last y tile
      double v = 0.0;
      if (j == kRC) v = (e0 + g + 4 * t < k) ? -bb[e0 + g + 4 * t] : 0.0;
      if (g + 4 * t == kRC) v = (e0 + j < k) ? -bb[e0 + j] : 0.0;
      U[LT * NB + LT][t] = v;
#pragma unroll
      for (int q = 0; q < NY - 1; ++q) {   // the full y tiles in front of it: right-hand side -b_eq[16 q .. 16 q + 15] in column kRC of tile (NT + q, LT)
        if (j == kRC) U[(NT + q) * NB + LT][t] = -bb[16 * q + g + 4 * t];
      }
    }
    const double lam_in = ka->lambda_vec ? ((const double*)ka->lambda_vec)[p * ka->lambda_vec_stride] : ka->lambda;
    const double lam = (!QPL && lam_in > 0.0) ? lam_in : 0.0;  // nonlinear.cc:187-189 (a given G already carries it)
    double padv[NT];
    pad_diag<NT, QPL>(j, nn, padv);
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        U[c * NB + c][t] += (j == g + 4 * t) ? (lam + dS[c] + padv[c]) : 0.0;
        const double rv = rp[16 * c + g + 4 * t];
        if (j == kRC) U[c * NB + LT][t] = rv;
      }
    }

    MO_STAMP(2);
    // ---- P5: block elimination with 16x16 pivot blocks
    __builtin_amdgcn_sched_barrier(0);
    // The dependent pivot chains get issue priority over the J streams of the other waves (those always have an MFMA ready and fill
    // whatever is left): +0.4 % on two boxes; priority on the J stream instead: -0.8 %.
    if (a.chain_prio) __builtin_amdgcn_s_setprio(1);
    bool ok = true;
    constexpr bool kLookAhead = SW == 6 || (NT >= MO_LA_MIN_NT && NY == 1);   // the large grids run one or two waves per SIMD: little else hides the sweeps there
    if constexpr (kLookAhead) {
      ok = block_eliminate_lookahead<NT>(U, k, g, j);
    } else {
#pragma unroll
    for (int pa = 0; pa < NB; ++pa) {
      ok = sweep_tile<SW>(U[pa * NB + pa], tile_pivots<NT, NY>(pa, k), g, j) && ok;
      __builtin_amdgcn_sched_barrier(0);
      MO_STAMP(3);
#pragma unroll
      for (int pc = pa + 1; pc < NB; ++pc) {
        d4 negZ = mfma4(U[pa * NB + pa], U[pa * NB + pc], d4{0.0, 0.0, 0.0, 0.0});  // (-T^-1) U_ac  (T^-1 is symmetric)
#pragma unroll
        for (int pb = pa + 1; pb <= pc; ++pb) U[pb * NB + pc] = mfma4(U[pa * NB + pb], negZ, U[pb * NB + pc]);
        __builtin_amdgcn_sched_barrier(0);  // one panel tile at a time: keeps a single -Z tile live (register pressure)
      }
      MO_STAMP(4);
    }
    }

    __builtin_amdgcn_sched_barrier(0);
    // ---- P6: backward substitution; xb[c] = solution at permuted position 16c + j (replicated over g)
    double xb[NB];
    back_substitute<NT, NY>(U, k, j, xb);

    MO_STAMP(5);
    // ---- P7: direction, step lengths, status
    ka = fresh_args();
    double dxv[NT];
    {
      double xn[NT];
      ldv<NT, QPL>(xs, j, xn);
#pragma unroll
      for (int c = 0; c < NT; ++c) dxv[c] = xb[c] - xn[c];
    }
    bool finite = true;
#pragma unroll
    for (int c = 0; c < NT; ++c) finite = finite && (fabs(dxv[c]) < INFINITY);
    if (g == 0) stv<NT, QPL>(dxs, j, dxv);
    lds_fence();
    double dsv[MC], dzv[MC], ap = 1.0, ad = 1.0;
#pragma unroll
    for (int ci = 0; ci < MC; ++ci) {
      const int ix = lane + 64 * ci;
      dsv[ci] = 0.0; dzv[ci] = 0.0;
      if (ix < m) {
        const double ca2 = cA[ix], cb2 = cB[ix], cs2 = cS[ix], cz2 = cZ[ix];       // re-read: not kept live across P5
        const int cvar2 = bad_index[ci] ? 0 : cV[ix];
        const double csi = MC == 1 ? cs_inv[ci] : rcp_f64(cs2);                     // one slot: kept in a register; more: recomputed
        const double r_pi = ca2 * xs[cvar2] + cb2 - cs2;                            // qp.cc:416
        dsv[ci] = ca2 * dxs[cvar2] + r_pi;                                          // qp.cc:361
        dzv[ci] = -(cz2 * csi) * dsv[ci] - csi * (cs2 * cz2 - mu);                  // qp.cc:362
        const double tau = ka->tau;
        if (cs2 + dsv[ci] <= 0.0 && fabs(dsv[ci]) > 0.0) ap = fmin(ap, -tau * cs2 * rcp_f64(dsv[ci]));  // qp.cc:498-503
        if (cz2 + dzv[ci] <= 0.0 && fabs(dzv[ci]) > 0.0) ad = fmin(ad, -tau * cz2 * rcp_f64(dzv[ci]));
        finite = finite && (fabs(dsv[ci]) < INFINITY) && (fabs(dzv[ci]) < INFINITY);
      }
    }
    ap = cross_row_min(row_min(ap));
    ad = cross_row_min(row_min(ad));
    double dyv[NY];                                                                 // y+ - y, equality row 16 q + j
#pragma unroll
    for (int q = 0; q < NY; ++q) {
      dyv[q] = (16 * q + j < k) ? (-xb[NT + q] - yb[16 * q + j]) : 0.0;
      finite = finite && (fabs(dyv[q]) < INFINITY);
    }
    int st = MO_STATUS_OK;
    if (!__all(finite)) st = MO_STATUS_NONFINITE;
    if (!ok) st = MO_STATUS_FACTORIZATION_FAILED;
    if (slack_bad) st = MO_STATUS_NONPOSITIVE_SLACK;
    if (any_bad_index) st = MO_STATUS_BAD_INDEX;
    const double nanv = __builtin_nan("");
    double* dp = (double*)ka->delta + p * ka->delta_stride;
    if (g == 0) {
      double outv[NT];
#pragma unroll
      for (int c = 0; c < NT; ++c) outv[c] = st == MO_STATUS_OK ? dxv[c] : nanv;
      stv_n<NT, QPL>(dp, j, nn, outv);
#pragma unroll
      for (int q = 0; q < NY; ++q)
        if (16 * q + j < k) dp[nn + m_lay + 16 * q + j] = st == MO_STATUS_OK ? dyv[q] : nanv;
    }
    if (no_ineq) {  // ds = dz = 0 (qp.cc:366-386 writes only dx, dy)
      for (int ix = lane; ix < m_lay; ix += 64) { dp[nn + ix] = st == MO_STATUS_OK ? 0.0 : nanv; dp[nn + m_lay + k + ix] = st == MO_STATUS_OK ? 0.0 : nanv; }
    }
#pragma unroll
    for (int ci = 0; ci < MC; ++ci) {
      const int ix = lane + 64 * ci;
      if (ix < m) {
        dp[nn + ix] = st == MO_STATUS_OK ? dsv[ci] : nanv;
        dp[nn + m + k + ix] = st == MO_STATUS_OK ? dzv[ci] : nanv;
      }
    }
    if (lane == 0) {
      if (ka->alpha) {
        ((double*)ka->alpha)[2 * p] = st == MO_STATUS_OK ? ap : nanv;
        ((double*)ka->alpha)[2 * p + 1] = st == MO_STATUS_OK ? ad : nanv;
      }
      if (ka->status) ka->status[p] = st;
    }
    lds_fence();  // LDS vectors are re-initialised by the next problem
    if (a.chain_prio) __builtin_amdgcn_s_setprio(0);
    MO_STAMP(6);
    if (last_of_chunk) {
      p = uniform64(next_ticket) + ticket_base;
      chunk_end = p + next_chunk;
    } else {
      ++p;
    }
  }
#ifdef MO_FUSED_STAMPS
  if ((threadIdx.x & 63) == 0 && a.debug) {
    for (int i = 0; i < 8; ++i) atomicAdd(a.debug + i, stamp_acc[i]);
    atomicAdd(a.debug + 13, stamp_acc[13]); atomicAdd(a.debug + 14, stamp_acc[14]);
    atomicAdd(a.debug + 8, 1ull);
    const unsigned long long life = stamp_prev - stamp_t0, rt = __builtin_amdgcn_s_memrealtime() - stamp_rt0;
    atomicMax(a.debug + 9, life);
    atomicMin(a.debug + 10, life);
    atomicAdd(a.debug + 11, rt);
    atomicMax(a.debug + 12, rt);
  }
#endif
}

// =====================================================================================================================
// Standalone linearisation (mo_linearize, the cost part of mo_fill_qp): LinearizeAndFillQP's G = J^T J + lambda I (lower triangle,
// strict upper written as 0), c = J^T r, 0.5 |r|^2 (nonlinear.cc:182-189; residual.hpp:206-224) with the same J stream as the step
// kernel.  The tiles leave the registers through scattered 8-byte stores (position -> natural index), 64 KB per problem at n = 64.
template <int NT, int WPS, int JMODE = JMODE_VECTOR>
__global__ __launch_bounds__(256 * WPS, WPS) void kkt_fused_linearize_kernel(const KernelArgs a) {
  using C = FusedCfg<NT, WPS>;
  constexpr int NB = NT + 1, SLOT = C::SLOT, D = C::D;
  constexpr int WAVES = 4 * WPS;
  __shared__ __attribute__((aligned(16))) char smem_all[WAVES * C::LDS];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* const smem = smem_all + wave * C::LDS;
  const unsigned ring_base = (unsigned)(uintptr_t)smem;
  const int nn = a.n, m_r = a.m_r;
  for (int i = (int)(threadIdx.x & 63); i < D * SLOT / 8; i += 64) reinterpret_cast<double*>(smem)[i] = 0.0;
  lds_fence();
  const int chunk_shift = 63 - __builtin_clzll((unsigned long long)gridDim.x * WAVES * 4);
  // Small launches -- at most a.static_rounds problems per wave -- are split STATICALLY, round by round, in slot-major wave order (first one wave on
  // every SIMD of every CU, then the second wave of every SIMD, ...): no ticket at all.  A wave must otherwise wait for a ticket just to
  // learn that nothing is left, and 3 072 waves asking one counter word at ~88 M atomics/s is 35 us -- as long as the whole first round of
  // BASELINE configs[1] (4 096 problems).  A partial round then also lands one wave per SIMD instead of three per SIMD on a third of the CUs.
  const long long waves_all = (long long)gridDim.x * WAVES;
  const bool st_rounds = a.static_rounds > 0 && a.batch <= (long long)a.static_rounds * waves_all;   // wave-uniform
  auto chunk_for = [&](long long observed) -> int {
    if (st_rounds) return 1;
    const long long c = (a.batch - observed) >> chunk_shift;
    return c < 1 ? 1 : (c > 8 ? 8 : (int)c);
  };
  auto take_ticket = [&](int chunk, long long p_now) -> unsigned long long {
    if (st_rounds) return (unsigned long long)p_now;   // static rounds: the next problem of this wave is p_now + waves_all (= "ticket" p_now + ticket_base)
    unsigned long long t = 0;
    if (lane_id() == 0) t = atomicAdd(a.ticket, (unsigned long long)chunk);
    return t;
  };
  auto uniform64 = [](unsigned long long v) -> long long {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  // The FIRST chunk of every wave is static (wave w of the persistent grid takes problems [w c0, (w + 1) c0)); tickets from the counter start
  // behind that part.  All waves asking one counter word for their first ticket at kernel start costs 3 072 / 88 M atomics/s = 35 us: most
  // of a small launch (BASELINE configs[1]: 4 096 problems) and 2 % of the headline one.
  int chunk = chunk_for(0);
  const long long ticket_base = (long long)gridDim.x * WAVES * chunk;
  long long p = st_rounds ? (long long)(wave >> 2) * ((long long)gridDim.x * 4) + (long long)blockIdx.x * 4 + (wave & 3) : ((long long)blockIdx.x * WAVES + wave) * chunk;
  long long chunk_end = p + chunk;
  while (p < a.batch) {
    const bool last_of_chunk = p + 1 >= chunk_end;
    int next_chunk = 0;
    unsigned long long next_ticket = 0;
    if (last_of_chunk) { next_chunk = chunk_for(p); next_ticket = take_ticket(next_chunk, p); }
    const int lane = lane_id();
    const int g = lane >> 4, j = lane & 15;
    JStream<NT, D, JMODE> stream;
    stream.init((const double*)a.J + p * a.J_stride, (const double*)a.r + p * a.r_stride, smem, ring_base, lane, g, j, m_r, nn,
                a.J_row_major ? (long long)a.J_ld : 1ll, a.J_row_major ? 1ll : (long long)a.J_ld);
    stream.prologue();
    d4 U[NB * NB];
#pragma unroll
    for (int q = 0; q < NB * NB; ++q) U[q] = d4{0.0, 0.0, 0.0, 0.0};
    double cpart[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) cpart[c] = 0.0;
    stream.template run<true>(U, cpart);
    double cvec[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) cvec[c] = cross_row_sum(cpart[c]);
    const double half_sq = 0.5 * cross_row_sum(stream.rsq);  // every lane of row g holds the sum over its rows 4s + g
    const double lam_in = a.lambda_vec ? ((const double*)a.lambda_vec)[p * a.lambda_vec_stride] : a.lambda;
    const double lam = lam_in > 0.0 ? lam_in : 0.0;  // nonlinear.cc:187-189
    double* Go = (double*)a.G_out + p * a.G_out_stride;
    const int ld = a.G_out_ld;
    // Whole-line stores.  The tile layout scatters G: a lane's four registers are four rows two apart, the lanes of a row are columns two
    // apart, and two tiles interleave in every 32 natural indices.  The 2 x 2 tiles (2 sa + p, 2 sb + q) together ARE the natural block
    // rows [32 sa, +32) x columns [32 sb, +32): each is staged through the (idle) ring as the 32 x 32 block of the LOWER triangle it belongs
    // to, column-major, and leaves as eight 16-byte-per-lane stores -- every 128-byte line of G written whole, once; the strict upper
    // blocks go out as zeros the same way.  (n on the tile grid and a 16-byte aligned G; anything else takes the element-wise path below.)
    const bool whole_lines = NT <= 6 && nn == 16 * NT && !(ld & 1) && !(a.G_out_stride & 1) && (((uintptr_t)a.G_out & 15) == 0);
    if (whole_lines) {   // (NT <= 6: the 128 grid keeps the element-wise path, it has no registers left for a second store sequence)
      double* const stage = reinterpret_cast<double*>(smem);
      static_assert(D * SLOT >= 32 * 32 * 8, "the ring holds one 32 x 32 block");
#pragma unroll
      for (int sa = 0; sa < NT / 2; ++sa) {
#pragma unroll
        for (int sb = sa; sb < NT / 2; ++sb) {
          if (sa == sb) {
#pragma unroll
            for (int pq = 0; pq < 2; ++pq) {                       // tiles (2 sa + pq, 2 sa + pq): rows and columns of one parity, both triangles present
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int natr = 2 * (g + 4 * t) + pq, natc = 2 * j + pq;
                const double v = U[(2 * sa + pq) * NB + 2 * sa + pq][t] + (natr == natc ? lam : 0.0);
                stage[natr + 32 * natc] = natr >= natc ? v : 0.0;
              }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {                          // tile (2 sa, 2 sa + 1): even rows x odd columns; its mirror image is not stored
              const int natr = 2 * (g + 4 * t), natc = 2 * j + 1;
              const int hi = natr > natc ? natr : natc, lo = natr > natc ? natc : natr;
              stage[hi + 32 * lo] = U[(2 * sa) * NB + 2 * sa + 1][t];
              stage[lo + 32 * hi] = 0.0;
            }
          } else {
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {                       // G(32 sa + natr, 32 sb + natc) lies above the diagonal: it is G(row natc, column natr) of the lower block
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int natr = 2 * (g + 4 * t) + (pq >> 1), natc = 2 * j + (pq & 1);
                stage[natc + 32 * natr] = U[(2 * sa + (pq >> 1)) * NB + 2 * sb + (pq & 1)][t];
              }
            }
          }
          lds_fence();
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const d2 v = *(const d2*)(stage + it * 128 + lane * 2);
            const int col = it * 4 + (lane >> 4), row = (lane & 15) * 2;
            *(d2*)(Go + (32 * sb + row) + (size_t)(32 * sa + col) * ld) = v;
            if (sa != sb) *(d2*)(Go + (32 * sa + row) + (size_t)(32 * sb + col) * ld) = d2{0.0, 0.0};   // the strict upper triangle stays exactly zero
          }
          lds_fence();
        }
      }
    } else {
#pragma unroll
    for (int ta = 0; ta < NT; ++ta) {
#pragma unroll
      for (int tb = ta; tb < NT; ++tb) {
        const int natc = 32 * (tb >> 1) + 2 * j + (tb & 1);        // variable of tile column position 16 tb + j
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = g + 4 * t;
          const int natr = 32 * (ta >> 1) + 2 * r + (ta & 1);      // variable of tile row position 16 ta + r
          if (natr < nn && natc < nn) {
            const int hi = natr > natc ? natr : natc, lo = natr > natc ? natc : natr;
            Go[hi + (size_t)lo * ld] = U[ta * NB + tb][t] + (hi == lo ? lam : 0.0);   // lower triangle (residual.hpp:216-220)
            if (hi != lo) Go[lo + (size_t)hi * ld] = 0.0;                             // the strict upper triangle stays exactly zero
          }
        }
      }
    }
    }
    if (g == 0) stv_n<NT, false>((double*)a.c_out + p * a.c_out_stride, j, nn, cvec);
    if (lane == 0 && a.half_sq_out) ((double*)a.half_sq_out)[p * (a.half_sq_stride ? a.half_sq_stride : 1)] = half_sq;
    if (last_of_chunk) { p = uniform64(next_ticket) + ticket_base; chunk_end = p + next_chunk; } else { ++p; }
  }
}

// =====================================================================================================================
// Fused interior-point Solve: QPInteriorPointSolver::Solve (qp.cc:100-151) with ComputeInitialGuess
// (qp.cc:439-482) and Iterate (qp.cc:153-201), one wavefront per QP, per-problem early exit.  Every pass re-streams J
// (from L2 / HBM) to rebuild the G tiles the previous factorisation consumed, evaluates the KKT residual
// [G x + c - A^T y - A_i^T z ; A x + b] as a tile product K [x; -y] in registers (qp.cc:404-419), and -- unlike the
// one-shot step kernel -- solves for the DIRECTION with the residual as right-hand side, exactly the reference's system
// (qp.cc:255-268, 337-363), so the loop keeps Newton's self-correcting behaviour down to tight KKT tolerances.
// All three BarrierStrategy values; PREDICTOR_CORRECTOR pushes its second right-hand side through the first solve's factors.
template <int NT, int WPS, int MC = 1, int NY = 1> struct SolveCfg {
  static constexpr int N = 16 * NT;
  static constexpr int NH = NT / 2;
  static constexpr int SLOT = NH * 1024 + 64;
  static constexpr int MCAP = 64 * MC;                    // constraint slots: MC per lane
  // CLDS: the constraint data (s, z, a, b, variable) live in LDS between the phases of a pass instead of in 9 MC registers across the
  // factorisation.  On where the tiles alone take most of the budget: two y tiles on the 64 grid are 21 tiles = 168 of the 256 registers
  // of two waves per SIMD (round 3: 308 - 408 B of scratch per lane, 4.4 x the algorithmic HBM traffic).  The price is LDS: 36 MCAP bytes
  // less for parked tiles (one or two more of them in the L2-resident global scratch).
  static constexpr bool CLDS = NY >= 2 && WPS >= 2;
  static constexpr int CONS = CLDS ? MCAP * 36 : 0;       // cS, cZ, cA, cB (doubles), cV (int)
  static constexpr int VEC = (6 * N + 32 * NY) * 8 + CONS;  // xs, xp, azS, diagS, rhoS, tmp, ysmall[32 NY] (+ the constraint arrays)
  static constexpr int D_FIT = ((160 * 1024) / (4 * WPS) - VEC) / SLOT;
  static constexpr int D_TUNED = NT > 4 ? 4 : (NT == 4 ? (WPS >= 3 ? 4 : 6) : 8);
  static constexpr int D = MC == 1 ? D_TUNED : (D_FIT > 8 ? 8 : D_FIT);
  static_assert(D >= 2, "LDS budget");
  // Tile park: G = J^T J + lambda I and c do not change between the passes of one Solve.  After the first pass has streamed J the ring
  // is idle for the rest of the problem, so its bytes (and whatever else the wave's share of the 160 KiB leaves) hold the G tiles
  // lane-linearly -- PARK_LDS of the NTILES tiles; the rest goes to a global scratch indexed by the wave's SLOT in the persistent grid
  // (a few KB per wave that stay in L2), not by problem.  Nothing of a cached pass then comes from HBM.
  static constexpr int NTILES = NT * (NT + 1) / 2;
  static constexpr int BUDGET = ((160 * 1024) / (4 * WPS)) & ~15;
  static constexpr int PARK_FIT = (BUDGET - VEC - N * 8) / 2048;
  static constexpr int PARK_LDS = PARK_FIT > NTILES ? NTILES : PARK_FIT;
  static constexpr int PARK_GLOBAL = NTILES - PARK_LDS;
  static constexpr int AREA = D * SLOT > PARK_LDS * 2048 + N * 8 ? D * SLOT : PARK_LDS * 2048 + N * 8;  // ring, later the parked tiles + c
  static constexpr int LDS = AREA + VEC;
  static_assert(PARK_LDS >= 0 && LDS <= BUDGET, "LDS budget");
};

__device__ inline double wave_sum_f64(double v) { return cross_row_sum(row_sum(v)); }

template <int NT, int WPS, int SW, bool QPL, int MC = 1, int JMODE = JMODE_VECTOR, int NY = 1, bool PCK = true>
__global__ __launch_bounds__(256 * WPS, WPS) void kkt_fused_solve_kernel(const KernelArgs a) {
  using C = SolveCfg<NT, WPS, MC, NY>;
  constexpr int N = C::N, NB = NT + NY, LT = NB - 1, SLOT = C::SLOT, D = C::D;  // LT: the tile column that carries the right-hand side
  constexpr int YN = 16 * NY;                                                    // equality rows the y tiles hold
  constexpr int WAVES = 4 * WPS;

  __shared__ __attribute__((aligned(16))) char smem_all[WAVES * C::LDS];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* const smem = smem_all + wave * C::LDS;
  double* const xs = reinterpret_cast<double*>(smem + C::AREA);  // x, natural order
  double* const xp = xs + N;                                      // x, permuted order
  double* const azS = xp + N;                                     // sum a z per variable, natural order
  double* const diagS = azS + N;                                  // barrier diagonal per variable
  double* const rhoS = diagS + N;                                 // inequality part of r_aug per variable
  double* const tmp = rhoS + N;                                   // layout-conversion scratch (R <-> V16, natural <-> permuted)
  double* const ysm = tmp + N;                                    // [0, YN): y ; [YN, 2 YN): -r_pe
  constexpr bool CLDS = C::CLDS;
  constexpr int MCAP = C::MCAP;
  double* const cS = ysm + 2 * YN;                                // CLDS: s, z, a, b, variable of constraint ix at index ix (see SolveCfg)
  double* const cZ = cS + MCAP;
  double* const cA = cZ + MCAP;
  double* const cB = cA + MCAP;
  int* const cV = reinterpret_cast<int*>(cB + MCAP);
  const unsigned ring_base = (unsigned)(uintptr_t)smem;

  // 1 / M for ComputeMu (qp.cc:509-516): one f64 division per kernel; the quotient is wave-uniform and is moved to a scalar register pair
  // (as a VGPR pair it was spilled and reloaded twice per pass)
  double inv_m = a.m > 0 ? 1.0 / (double)a.m : 0.0;   // (k, m, n, m_r are read per problem / per pass, see below)
  inv_m = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(inv_m)), __builtin_amdgcn_readfirstlane(__double2loint(inv_m)));
  for (int i = (int)(threadIdx.x & 63); i < D * SLOT / 8; i += 64) reinterpret_cast<double*>(smem)[i] = 0.0;
  lds_fence();

  const int chunk_shift = 63 - __builtin_clzll((unsigned long long)gridDim.x * WAVES * 4);  // ~ remaining / (4 waves' worth)
  // Small launches -- at most a.static_rounds problems per wave -- are split STATICALLY, round by round, in slot-major wave order (first one wave on
  // every SIMD of every CU, then the second wave of every SIMD, ...): no ticket at all.  A wave must otherwise wait for a ticket just to
  // learn that nothing is left, and 3 072 waves asking one counter word at ~88 M atomics/s is 35 us -- as long as the whole first round of
  // BASELINE configs[1] (4 096 problems).  A partial round then also lands one wave per SIMD instead of three per SIMD on a third of the CUs.
  const long long waves_all = (long long)gridDim.x * WAVES;
  const bool st_rounds = a.static_rounds > 0 && a.batch <= (long long)a.static_rounds * waves_all;   // wave-uniform
  auto chunk_for = [&](long long observed) -> int {
    if (st_rounds) return 1;
    const long long c = (a.batch - observed) >> chunk_shift;
    return c < 1 ? 1 : (c > 8 ? 8 : (int)c);
  };
  auto take_ticket = [&](int chunk, long long p_now) -> unsigned long long {
    if (st_rounds) return (unsigned long long)p_now;   // static rounds: the next problem of this wave is p_now + waves_all (= "ticket" p_now + ticket_base)
    unsigned long long t = 0;
    if (lane_id() == 0) t = atomicAdd(a.ticket, (unsigned long long)chunk);
    return t;
  };
  auto uniform64 = [](unsigned long long v) -> long long {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  // The FIRST chunk of every wave is static (wave w of the persistent grid takes problems [w c0, (w + 1) c0)); tickets from the counter start
  // behind that part.  All waves asking one counter word for their first ticket at kernel start costs 3 072 / 88 M atomics/s = 35 us: most
  // of a small launch (BASELINE configs[1]: 4 096 problems) and 2 % of the headline one.
  int chunk = chunk_for(0);
  const long long ticket_base = (long long)gridDim.x * WAVES * chunk;
  long long p = st_rounds ? (long long)(wave >> 2) * ((long long)gridDim.x * 4) + (long long)blockIdx.x * 4 + (wave & 3) : ((long long)blockIdx.x * WAVES + wave) * chunk;
  long long chunk_end = p + chunk;

  while (p < a.batch) {
    // The argument block is re-read from the kernarg segment where it is used (scalar loads, K$ hits): held in SGPRs for the whole kernel it
    // overflows the SGPR file -- round 2's build parked ~270 SGPRs in VGPR lanes and executed hundreds of v_readlane / v_writelane VALU
    // instructions per pass to get at them (the step kernel has had this since round 1).
    KArgs ka = fresh_args();
    // (the shape, too: conditions on k / m / n would otherwise be evaluated once per kernel and their lane masks kept in SGPR pairs throughout)
    const int k = ka->k, m = ka->m, nn = ka->n;   // actual number of variables nn <= N (see the step kernel)
    const bool last_of_chunk = p + 1 >= chunk_end;
    int next_chunk = 0;
    unsigned long long next_ticket = 0;
    if (last_of_chunk) {
      next_chunk = chunk_for(p);
      next_ticket = take_ticket(next_chunk, p);
    }
    if (ka->skip && ka->skip[p * ka->skip_stride] >= 0) {  // wave-uniform: a problem the caller's outer loop has finished with
      if (last_of_chunk) { p = uniform64(next_ticket) + ticket_base; chunk_end = p + next_chunk; } else { ++p; }
      continue;
    }
    const int lane = lane_id();
    const int g = lane >> 4, j = lane & 15;

    // (per-problem addresses -- J, r, A_eq, vars, records, the parked-tile scratch -- are formed from the kernarg block WHERE they are used,
    // a few scalar instructions each: carried in SGPRs across the pass loop they were spilled to VGPR lanes, i.e. v_readlane / v_writelane
    // work on the datapath the passes are bound by)

    // ---- constants of the problem
    int cvar[MC]; double ca[MC], cb[MC];  // constraint lane + 64 ci
#pragma unroll
    for (int ci = 0; ci < MC; ++ci) {
      const int ix = lane + 64 * ci;
      cvar[ci] = 0; ca[ci] = 1.0; cb[ci] = 0.0;
      if (ix < m) {
        cvar[ci] = ka->cons_var[p * ka->cons_stride + ix];
        ca[ci] = ((const double*)ka->cons_a)[p * ka->cons_stride + ix];
        cb[ci] = ((const double*)ka->cons_b)[p * ka->cons_stride + ix];
      }
    }
    // CLDS: every phase that computes with the constraint data fetches its registers from LDS first (all of them are overwritten, so
    // nothing of them is live across the factorisation or the corrector's second solve) and writes s / z back when it changed them.
    double cs[MC], cz[MC];
#pragma unroll
    for (int ci = 0; ci < MC; ++ci) { cs[ci] = 1.0; cz[ci] = 1.0; }
    auto cons_load = [&](int lane) {
      if constexpr (CLDS) {
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) {
          const int ix = lane + 64 * ci;
          cvar[ci] = cV[ix]; ca[ci] = cA[ix]; cb[ci] = cB[ix]; cs[ci] = cS[ix]; cz[ci] = cZ[ix];
        }
      }
    };
    auto cons_store_state = [&](int lane) {
      if constexpr (CLDS) {
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) { cS[lane + 64 * ci] = cs[ci]; cZ[lane + 64 * ci] = cz[ci]; }
      }
    };
    auto cons_store_all = [&](int lane) {
      if constexpr (CLDS) {
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) { cV[lane + 64 * ci] = cvar[ci]; cA[lane + 64 * ci] = ca[ci]; cB[lane + 64 * ci] = cb[ci]; }
        cons_store_state(lane);
      }
    };

    // ---- state.  x and y LIVE IN LDS (xs: natural order, xp: permuted position order, ysm[0, YN): y, zero beyond k) -- at the top of every
    // pass and at every exit of the pass loop the LDS copies are the current iterate; registers hold them only while a phase computes with
    // them.  (Round 3 carried x, y, the residual r_d / r_pe, r_pi / r_comp and six record scalars in VGPRs across the factorisation: with the
    // 120 tile registers that was 136 B of scratch per lane at the 256-register budget of two waves per SIMD, and the spill traffic
    // competed with the parked tiles for L2.)  s / z stay in registers, one slot per constraint lane -- unless CLDS.
    // (the lane coordinates are arguments: a pass uses its own opaque copies, nothing lane-derived may stay live across a factorisation)
    auto publish_x = [&](const double (&xq)[NT], int g, int j) {   // all lanes hold the V16 copy (replicated over g); row 0 writes both orders
      if (g == 0) {
        stv<NT, QPL>(xs, j, xq);
#pragma unroll
        for (int c = 0; c < NT; ++c) xp[16 * c + j] = xq[c];
      }
    };
    auto publish_y = [&](const double (&yq)[NY], int g, int j) {
      if (g == 0) {
#pragma unroll
        for (int q = 0; q < NY; ++q) ysm[16 * q + j] = (16 * q + j < k) ? yq[q] : 0.0;
      }
    };
    // MODE_RESIDUAL: EvaluateKKTConditions + ComputeErrors (qp.cc:391-437) on the caller's state -- part A of a pass, then the outputs
    const bool residual_mode = ka->mode == MODE_RESIDUAL;
    const bool iterate_mode = ka->mode == MODE_ITERATE || residual_mode;  // one Iterate (qp.cc:153-201) on the caller's state and mu
    {
      double xq[NT], yq[NY];
#pragma unroll
      for (int q = 0; q < NY; ++q) yq[q] = 0.0;
#pragma unroll
      for (int c = 0; c < NT; ++c) xq[c] = 0.0;
      if (iterate_mode || ka->sp.initial_guess_method == MO_GUESS_USER_PROVIDED) {  // qp.cc:440-442
        const double* const vp = (const double*)ka->vars + p * ka->vars_stride;
        ldv_n<NT, QPL>(vp, j, nn, xq);
#pragma unroll
        for (int q = 0; q < NY; ++q)
          if (16 * q + j < k) yq[q] = vp[nn + m + 16 * q + j];
#pragma unroll
        for (int ci = 0; ci < MC; ++ci)
          if (lane + 64 * ci < m) { cs[ci] = vp[nn + lane + 64 * ci]; cz[ci] = vp[nn + m + k + lane + 64 * ci]; }
      }
      publish_x(xq, g, j);
      publish_y(yq, g, j);
      lds_fence();
    }
    bool lane_bad_index = false;
#pragma unroll
    for (int ci = 0; ci < MC; ++ci) lane_bad_index = lane_bad_index || ((lane + 64 * ci < m) && ((cvar[ci] < 0) || (cvar[ci] >= nn)));
    const bool bad_index = __any(lane_bad_index);
    if (bad_index) {
#pragma unroll
      for (int ci = 0; ci < MC; ++ci) cvar[ci] = 0;
    }
    cons_store_all(lane);

    int st = bad_index ? MO_STATUS_BAD_INDEX : MO_STATUS_OK;
    int term = MO_MAX_ITERATIONS, it = 0;
    double mu = iterate_mode ? (ka->mu ? ((const double*)ka->mu)[p * ka->mu_stride] : 0.0) : ka->sp.initial_mu;
    bool guess_pass = !iterate_mode && ka->sp.initial_guess_method == MO_GUESS_SOLVE_EQUALITY_CONSTRAINED;
    auto records_of = [&](KArgs kq) -> double* {   // this problem's iteration records (NULL: the caller wants none)
      return kq->iterations ? (double*)kq->iterations + (size_t)p * kq->sp.max_iterations * MO_ITER_RECORD : nullptr;
    };

    // s = max(1e-9, a x + b), z = 1/s after clamping x into the feasible region in constraint order (qp.cc:464-481)
    auto clamp_and_init_slacks = [&](int lane, int g, int j) {   // on the x that xs holds; leaves the clamped x in xs and xp
      cons_load(lane);
      for (int c = 0; c < m; ++c) {  // wave-uniform loop; one constraint at a time keeps the reference's order
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) {
          if (lane + 64 * ci == c) {
            const double x0 = xs[cvar[ci]];
            double x1;
            if (ca[ci] < 0.0) { const double lim = cb[ci] / -ca[ci]; x1 = x0 < lim ? x0 : lim; }  // ClampX, qp.hpp:43-53
            else { const double lim = -cb[ci] / ca[ci]; x1 = x0 > lim ? x0 : lim; }
            xs[cvar[ci]] = x1;
          }
        }
        lds_fence();
      }
      {
        double xq[NT];
        ldv<NT, QPL>(xs, j, xq);
        if (g == 0) {
#pragma unroll
          for (int c = 0; c < NT; ++c) xp[16 * c + j] = xq[c];
        }
      }
      double sz = 0.0;
#pragma unroll
      for (int ci = 0; ci < MC; ++ci) {
        if (lane + 64 * ci < m) {
          const double sv = ca[ci] * xs[cvar[ci]] + cb[ci];
          cs[ci] = sv > 1.0e-9 ? sv : 1.0e-9;
          cz[ci] = 1.0 / cs[ci];
          sz += cs[ci] * cz[ci];
        }
      }
      cons_store_state(lane);
      if (ka->sp.initialize_mu_with_complementarity) {  // qp.cc:115
        const double t = wave_sum_f64(sz);
        mu = t * inv_m;
      }
    };
    if (st == MO_STATUS_OK && !iterate_mode && ka->sp.initial_guess_method == MO_GUESS_NAIVE) clamp_and_init_slacks(lane, g, j);
    if (!iterate_mode && ka->sp.initial_guess_method == MO_GUESS_USER_PROVIDED && ka->sp.initialize_mu_with_complementarity) {
      double sz = 0.0;  // qp.cc:115 on the caller's state: mu = s^T z / M (0 without inequalities, qp.cc:509-516)
#pragma unroll
      for (int ci = 0; ci < MC; ++ci)
        if (lane + 64 * ci < m) sz = fma(cs[ci], cz[ci], sz);
      mu = wave_sum_f64(sz) * inv_m;
    }

    double n_rd2 = 0, n_rpe2 = 0, n_rc2 = 0, n_rc1 = 0, n_rpi2 = 0;
    // ComputeErrors (qp.cc:423-437) as SQUARED norms: the decisions compare squares (all quantities are non-negative); the four f64
    // square roots per call (~120 VALU instructions) are only taken for the iteration records, i.e. when the caller asked for them.
    auto kkt_errors_sq = [&](double mu_e, double (&o)[4]) {
      o[0] = n_rd2;
      o[2] = k > 0 ? n_rpe2 : 0.0;
      if (m > 0) {
        const double corrected = n_rc2 - 2 * (n_rc1 * mu_e) + (mu_e * mu_e) * (double)m;
        o[1] = corrected > 0.0 ? corrected : 0.0;
        o[3] = n_rpi2;
      } else { o[1] = 0.0; o[3] = 0.0; }
    };
    // G = J^T J + lambda I and c = J^T r do not change between the passes of one Solve: the first pass parks its tiles in a per-problem
    // scratch (plan-owned, ka->G_out; lane-linear, every lane reads back exactly what it wrote), later passes reload 22 KB instead of
    // re-streaming 64 KB of J and redoing 320 of the 440 MFMAs (n = 64 figures).
    constexpr int PARK_LDS = C::PARK_LDS, PARK_GLOBAL = C::PARK_GLOBAL;
    double* const park = reinterpret_cast<double*>(smem);             // PARK_LDS tiles (256 doubles each, lane-linear), then c (V16, N doubles)
    double* const cpark = park + PARK_LDS * 256;
    // the tiles that do not fit: global scratch of this wave's slot in the persistent grid (plan-owned; NULL: re-stream / re-load every pass)
    auto tile_scratch_of = [&](KArgs kq) -> double* {
      return (PARK_GLOBAL > 0 && kq->G_out) ? (double*)kq->G_out + ((size_t)blockIdx.x * WAVES + wave) * (size_t)kq->G_out_stride : nullptr;
    };
    const bool can_park = PARK_GLOBAL == 0 || ka->G_out != nullptr;
    bool tiles_cached = false;
    // Lanes whose J piece lies beyond the row never receive DMA data and must read zeros from the ring: a problem that parked tiles there
    // leaves it dirty, so streams that rely on the zeros (n below the grid, flat / gather pieces) clear it again first.
    if (!QPL && (JMODE != JMODE_VECTOR || nn != N)) {
      for (int i = lane; i < D * SLOT / 8; i += 64) reinterpret_cast<double*>(smem)[i] = 0.0;
      lds_fence();
    }
    double mu_used = mu;   // the mu handed to the previous Iterate
    // Mehrotra predictor-corrector (qp.cc:170-187): solve with mu = 0, probe alpha(tau = 1), then solve again with the second-order
    // term ds_aff dz_aff and mu = sigma mu_input on the right-hand side -- through the factors of the first solve (solve_second_rhs).
    // (PCK = false: an instantiation without the corrector's code -- its second solve keeps every factor tile alive behind the back-substitution,
    // which with two y tiles on the 64 grid is what does not fit 256 registers; the launcher picks by barrier strategy)
    const bool use_pc = PCK && (iterate_mode ? ka->barrier_strategy : ka->sp.barrier_strategy) == MO_PREDICTOR_CORRECTOR && m > 0;

    while (st == MO_STATUS_OK) {
      const bool include_ineq = !guess_pass && !(residual_mode && (ka->flags & MO_STEP_NO_INEQUALITIES));
      // lane coordinates are re-made opaque every pass: nothing derived from them may be hoisted out of the pass loop and
      // kept in VGPRs across the factorisation (see the step kernel)
      const int lane = lane_id(), g = lane >> 4, j = lane & 15;  // shadow the per-problem copies inside the pass
      ka = fresh_args();
      const int k = ka->k, m = ka->m, nn = ka->n, m_r = ka->m_r;
      // ---------------------------------------------------------------- part A: tiles, residual, norms
      JStream<NT, D, JMODE, NY> stream;
      const bool build_now = __builtin_amdgcn_readfirstlane((int)!tiles_cached) != 0;          // wave-uniform, and hipcc must know it
      const bool stream_now = !QPL && build_now;
      double* const Gt = tile_scratch_of(ka);
      if (stream_now) {
        const double* const Jp = (const double*)ka->J + p * ka->J_stride;
        const double* const rg = (const double*)ka->r + p * ka->r_stride;
        stream.init(Jp, rg, smem, ring_base, lane, g, j, m_r, nn, ka->J_row_major ? (long long)ka->J_ld : 1ll, ka->J_row_major ? 1ll : (long long)ka->J_ld);
        stream.prologue();
      }
      d4 U[NB * NB];
#pragma unroll
      for (int q = 0; q < NB * NB; ++q) U[q] = d4{0.0, 0.0, 0.0, 0.0};
      load_a_tiles<NT, QPL, NY>(k > 0 ? (const double*)ka->A + p * ka->A_stride : nullptr, ka->A_ld, k, nn, g, j, U);
      // (xs / xp / ysm hold the state, see above); zero the per-variable scatter arrays
      if (lane < N / 2) {
        azS[2 * lane] = 0.0; azS[2 * lane + 1] = 0.0;
        diagS[2 * lane] = 0.0; diagS[2 * lane + 1] = 0.0;
        rhoS[2 * lane] = 0.0; rhoS[2 * lane + 1] = 0.0;
      }
      // c (= J^T r, or the caller's c) goes to LDS (cpark: behind the parked tiles, inside the drained ring) as soon as it exists and is read
      // back where the residual needs it: nothing of it occupies registers during the tile products
      if (QPL && build_now) {
        double cvec[NT];
        load_g_tiles<NT, NY>((const double*)ka->G + p * ka->G_stride, ka->G_ld, (const double*)ka->c + p * ka->c_stride, nn, g, j, U, cvec);
        if (g == 0) {
#pragma unroll
          for (int c = 0; c < NT; ++c) cpark[16 * c + j] = cvec[c];
        }
      } else if (stream_now) {
        double cpart[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) cpart[c] = 0.0;
        stream.run(U, cpart);
        const double lam_in = ka->lambda_vec ? ((const double*)ka->lambda_vec)[p * ka->lambda_vec_stride] : ka->lambda;
        const double lam = lam_in > 0.0 ? lam_in : 0.0;   // (J-level input; a given G already carries the LM damping)
#pragma unroll
        for (int c = 0; c < NT; ++c) {  // G = J^T J + lambda I (nonlinear.cc:187-189): lambda is part of G in the residual too
#pragma unroll
          for (int t = 0; t < 4; ++t) U[c * NB + c][t] += (j == g + 4 * t) ? lam : 0.0;
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) {
          const double cv = cross_row_sum(cpart[c]);
          if (g == 0) cpark[16 * c + j] = cv;
        }
      } else {  // fetch what the first pass parked
        int ti = 0;
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) {
#pragma unroll
          for (int tb = ta; tb < NT; ++tb, ++ti) {
            if (ti < PARK_LDS) U[ta * NB + tb] = *(const d4*)(park + (ti * 64 + lane) * 4);
            else U[ta * NB + tb] = *(const d4*)(Gt + ((size_t)(ti - PARK_LDS) * 64 + lane) * 4);
          }
        }
      }
      if (build_now && can_park && !iterate_mode) {  // park the tiles and c for the following passes (the stream has drained: the ring is free)
        int ti = 0;
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) {
#pragma unroll
          for (int tb = ta; tb < NT; ++tb, ++ti) {
            if (ti < PARK_LDS) *(d4*)(park + (ti * 64 + lane) * 4) = U[ta * NB + tb];
            else *(d4*)(Gt + ((size_t)(ti - PARK_LDS) * 64 + lane) * 4) = U[ta * NB + tb];
          }
        }
        tiles_cached = true;
      }
      {  // unit diagonal for the padding variables (index >= nn): they stay at zero
        double padv[NT];
        pad_diag<NT, QPL>(j, nn, padv);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
          for (int t = 0; t < 4; ++t) U[c * NB + c][t] += (j == g + 4 * t) ? padv[c] : 0.0;
        }
      }
      lds_fence();
      cons_load(lane);
      double r_pi[MC], r_comp[MC];
#pragma unroll
      for (int ci = 0; ci < MC; ++ci) {
        r_pi[ci] = 0.0; r_comp[ci] = 0.0;
        if (include_ineq && lane + 64 * ci < m) {
          atomicAdd(&azS[cvar[ci]], ca[ci] * cz[ci]);                    // qp.cc:415
          r_pi[ci] = fma(ca[ci], xs[cvar[ci]], cb[ci]) - cs[ci];         // qp.cc:416
          r_comp[ci] = cs[ci] * cz[ci];                                  // qp.cc:417
        }
      }
      double r_d[NT], r_pe[NY];
      {
      // w = K [x; -y] as tile products: type 1 (sum over tile rows, result on lanes) over every stored tile,
      // type 2 (sum over tile columns, result on rows) over the strictly upper tiles; the latter goes through LDS once.
      double xv[NT], yv[NY];   // V16 copies of the state for the type-2 products (yv is zero on lanes beyond k)
#pragma unroll
      for (int c = 0; c < NT; ++c) xv[c] = xp[16 * c + j];
#pragma unroll
      for (int q = 0; q < NY; ++q) yv[q] = ysm[16 * q + j];
      double acc1[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc1[b] = 0.0;
#pragma unroll
      for (int ra = 0; ra < NB; ++ra) {
        // compiler-level memory barrier: the row operands of tile row ra are fetched HERE -- hipcc otherwise hoists the LDS reads of all tile
        // rows above the first one (32 registers live across the whole product at n = 64; its own pressure report: 293 with two y tiles)
        asm volatile("" ::: "memory");
        double vR[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) vR[t] = ra < NT ? xp[16 * ra + g + 4 * t] : 0.0;
#pragma unroll
        for (int b = ra; b < NB; ++b) {
          if (ra >= NT) continue;  // the y diagonal block of K is zero
#pragma unroll
          for (int t = 0; t < 4; ++t) acc1[b] = fma(U[ra * NB + b][t], vR[t], acc1[b]);
        }
        if (ra < NT) {
          double pt[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            pt[t] = 0.0;
#pragma unroll
            for (int b = ra + 1; b < NB; ++b) pt[t] = fma(U[ra * NB + b][t], b < NT ? xv[b] : -yv[b < NT ? 0 : b - NT], pt[t]);  // yv is zero on lanes beyond k
          }
          // four row sums as one transposed butterfly (27 VALU instead of 48): lane j of each row holds the sum for row g + 4 (j & 3)
          const double ws = row_sum4_scatter(pt[0], pt[1], pt[2], pt[3]);
          if (j < 4) tmp[16 * ra + g + 4 * j] = ws;
        }
        // One tile row at a time: the empty asm statements pin the partial sums of this row HERE.  hipcc otherwise sinks the whole chains of
        // the y columns (acc1[NT ..]) to their use behind the loop and keeps the row operands of ALL tile rows alive until then -- 32 registers
        // at n = 64 (its own pressure report for two y tiles: 293 live registers at that point).
#pragma unroll
        for (int b = ra; b < NB; ++b) asm volatile("" : "+v"(acc1[b]));
      }
      lds_fence();
#pragma unroll
      for (int h = 0; h < NT / 2; ++h) {   // (per pair of tiles, as ldv reads azS; two y tiles: one pair at a time, see part B)
        if constexpr (NY >= 2) asm volatile("" ::: "memory");
        double azv[2];
        if (QPL) { azv[0] = azS[32 * h + j]; azv[1] = azS[32 * h + 16 + j]; }
        else { const d2 av = *(const d2*)(azS + 32 * h + 2 * j); azv[0] = av[0]; azv[1] = av[1]; }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = 2 * h + e;
          r_d[c] = cross_row_sum(acc1[c]) + tmp[16 * c + j] + cpark[16 * c + j] - azv[e];  // qp.cc:404-406, 415
          if constexpr (NY >= 2) asm volatile("" : "+v"(r_d[c]));   // computed HERE (hipcc sinks the sums behind the r_pe branches, operands alive)
        }
      }
      if constexpr (NY >= 2) asm volatile("" ::: "memory");
      {
        const double* const bp = (const double*)ka->b + p * ka->b_stride;
#pragma unroll
        for (int q = 0; q < NY; ++q) r_pe[q] = (16 * q + j < k) ? cross_row_sum(acc1[NT + q]) + bp[16 * q + j] : 0.0;  // qp.cc:408
      }
      {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < NT; ++c) t = fma(r_d[c], r_d[c], t);
        n_rd2 = row_sum(t);
        double t_pe = 0.0;
#pragma unroll
        for (int q = 0; q < NY; ++q) t_pe = fma(r_pe[q], r_pe[q], t_pe);
        n_rpe2 = row_sum(t_pe);
        double s_rc2 = 0.0, s_rc1 = 0.0, s_rpi2 = 0.0;
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) { s_rc2 = fma(r_comp[ci], r_comp[ci], s_rc2); s_rc1 += r_comp[ci]; s_rpi2 = fma(r_pi[ci], r_pi[ci], s_rpi2); }
        n_rc2 = wave_sum_f64(s_rc2);
        n_rc1 = wave_sum_f64(s_rc1);
        n_rpi2 = wave_sum_f64(s_rpi2);
        n_rd2 = readlane_f64(n_rd2, 0); n_rpe2 = readlane_f64(n_rpe2, 0);  // uniform copies
      }
      }
      if (residual_mode) {  // r_ = [r_d | r_comp | r_pe | r_pi] (qp.cc:391-420) and the four norms of ComputeErrors (qp.cc:423-437)
        double* ro = (double*)ka->r_out + p * ka->r_out_stride;
        if (g == 0) {
          stv_n<NT, QPL>(ro, j, nn, r_d);
#pragma unroll
          for (int q = 0; q < NY; ++q)
            if (16 * q + j < k) ro[nn + m + 16 * q + j] = r_pe[q];
        }
#pragma unroll
        for (int ci = 0; ci < MC; ++ci)
          if (lane + 64 * ci < m) { ro[nn + lane + 64 * ci] = r_comp[ci]; ro[nn + m + k + lane + 64 * ci] = r_pi[ci]; }
        if (ka->kkt_out) {
          double kq[4];
          kkt_errors_sq(mu, kq);
          if (!include_ineq) { kq[1] = 0.0; kq[3] = 0.0; }
          const double e0 = sqrt(kq[0]), e1 = sqrt(kq[1]), e2 = sqrt(kq[2]), e3 = sqrt(kq[3]);
          if (lane == 0) { double* ko = (double*)ka->kkt_out + 4 * p; ko[0] = e0; ko[1] = e1; ko[2] = e2; ko[3] = e3; }
        }
        break;
      }
      if (!guess_pass && !iterate_mode) {
        // ---- the decision point of Solve (qp.cc:116-147)
        const bool have_prev = it > 0;
        bool stop = false;
        double kf[4] = {0.0, 0.0, 0.0, 0.0};
        if (have_prev) {
          kkt_errors_sq(mu_used, kf);                               // kkt_after of the previous iteration (squared), qp.cc:127
          const double cur_mu = n_rc1 * inv_m;                      // ComputeMu, qp.cc:509-516 (one f64 division per kernel, not per pass)
          double kmax2 = kf[0];                                     // KKTError::Max() squared
          kmax2 = kf[1] > kmax2 ? kf[1] : kmax2; kmax2 = kf[2] > kmax2 ? kf[2] : kmax2; kmax2 = kf[3] > kmax2 ? kf[3] : kmax2;
          // The operands are wave-uniform VALUES in vector registers; __any() turns each verdict into a scalar condition, so that the exits of
          // the pass loop are scalar branches and `it`, `term`, `st` and the loop's flags live in SGPRs (as divergent branches they cost an
          // EXEC-mask pair each and turned every one of those into a VGPR).
          if (__any(kmax2 < ka->sp.termination_kkt_tol * ka->sp.termination_kkt_tol && cur_mu < ka->sp.termination_complementarity_tol)) {  // qp.cc:132-137
            term = MO_SATISFIED_KKT_TOL;
            stop = true;
          } else if (__any(kmax2 <= mu * mu) || !ka->sp.decrease_mu_only_on_small_error) {        // qp.cc:140-146 (mu > 0)
            if (ka->sp.barrier_strategy == MO_FIXED_DECREASE) mu *= ka->sp.sigma;
            else mu = ka->sp.sigma * cur_mu;
          }
        }
        if (it >= ka->sp.max_iterations) stop = true;                    // MAX_ITERATIONS, qp.cc:149
        double* const iter_out = records_of(ka);
        if (iter_out) {                                              // wave-uniform
          // The records want NORMS: kkt_after of the previous iteration (lanes 0-3) and kkt_prev of this one (lanes 4-7, qp.cc:118) take
          // ONE lane-parallel f64 square root (~25 VALU instructions) instead of eight wave-wide ones.
          double ki[4] = {0.0, 0.0, 0.0, 0.0};
          if (!stop) kkt_errors_sq(mu, ki);
          double sq = kf[0];
          sq = lane == 1 ? kf[1] : sq; sq = lane == 2 ? kf[2] : sq; sq = lane == 3 ? kf[3] : sq;
          sq = lane == 4 ? ki[0] : sq; sq = lane == 5 ? ki[1] : sq; sq = lane == 6 ? ki[2] : sq; sq = lane == 7 ? ki[3] : sq;
          const double rt = sqrt(sq);
          if (have_prev && lane < 4) iter_out[(size_t)(it - 1) * MO_ITER_RECORD + 4 + lane] = rt;
          if (!stop && lane >= 4 && lane < 8) iter_out[(size_t)it * MO_ITER_RECORD + (lane - 4)] = rt;
          // (IPIterationOutputs of iteration it - 1 -- mu, alpha, alpha_probe, mu_affine -- were written when that iteration ended)
        }
        if (stop) break;
      }
      // ---------------------------------------------------------------- part B: right-hand side, factorisation, direction
      ka = fresh_args();
      const bool predictor_pass = use_pc && !guess_pass;  // Mehrotra: solve with mu = 0, probe, then the corrector through the same factors
      const double mu_step = m > 0 ? (predictor_pass ? 0.0 : mu) : 0.0;  // qp.cc:165-187
      double aff[MC], cs_inv[MC];
#pragma unroll
      for (int ci = 0; ci < MC; ++ci) {
        aff[ci] = 0.0;  // ds_aff dz_aff (qp.cc:341), set by the predictor
        cs_inv[ci] = 1.0;
      }
      if (include_ineq) {
        bool lane_bad_slack = false;
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) lane_bad_slack = lane_bad_slack || ((lane + 64 * ci < m) && !(cs[ci] > 0.0));
        if (__any(lane_bad_slack)) { st = MO_STATUS_NONPOSITIVE_SLACK; break; }  // qp.cc:285
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) {
          cs_inv[ci] = rcp_f64(cs[ci]);
          if (lane + 64 * ci < m) {
            const double zs = cz[ci] * cs_inv[ci];
            atomicAdd(&diagS[cvar[ci]], ca[ci] * zs * ca[ci]);                                                        // qp.cc:296
            atomicAdd(&rhoS[cvar[ci]], ca[ci] * zs * r_pi[ci] + ca[ci] * (r_comp[ci] + aff[ci] - mu_step) * cs_inv[ci]);  // qp.cc:340-341
          }
        }
      }
      lds_fence();
      // (two y tiles: 168 tile registers -- the loops below fetch their LDS operands one tile at a time, behind compiler-level memory barriers,
      // instead of all sixteen / thirty-two values up front)
#pragma unroll
      for (int h = 0; h < NT / 2; ++h) {   // tiles 2h, 2h + 1: ldv's pair layout (QPL: positions 16 (2h) + j, 16 (2h + 1) + j)
        if constexpr (NY >= 2) asm volatile("" ::: "memory");
        double dd[2], rr[2];
        if (QPL) { dd[0] = diagS[32 * h + j]; dd[1] = diagS[32 * h + 16 + j]; rr[0] = rhoS[32 * h + j]; rr[1] = rhoS[32 * h + 16 + j]; }
        else { const d2 dv = *(const d2*)(diagS + 32 * h + 2 * j), rv = *(const d2*)(rhoS + 32 * h + 2 * j); dd[0] = dv[0]; dd[1] = dv[1]; rr[0] = rv[0]; rr[1] = rv[1]; }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = 2 * h + e;
#pragma unroll
          for (int t = 0; t < 4; ++t) U[c * NB + c][t] += (j == g + 4 * t) ? dd[e] : 0.0;
          if (g == 0) tmp[16 * c + j] = -(r_d[c] + rr[e]);          // -r_aug, position order (qp.cc:337-342)
          if (g == 0) azS[16 * c + j] = r_d[c];                     // r_d for the corrector's right-hand side (azS is dead: r_d consumed it)
        }
      }
      if (g == 0) {
#pragma unroll
        for (int q = 0; q < NY; ++q) ysm[YN + 16 * q + j] = -r_pe[q];
      }
      lds_fence();
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        if constexpr (NY >= 2) asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double rv = tmp[16 * c + g + 4 * t];
          if (j == kRC) U[c * NB + LT][t] = rv;
        }
      }
      if constexpr (NY >= 2) asm volatile("" ::: "memory");
#pragma unroll
      for (int t = 0; t < 4; ++t) {  // last y diagonal tile = [0, -r_pe; -r_pe^T, 0]
        double v = 0.0;
        if (j == kRC) v = ysm[YN + 16 * (NY - 1) + g + 4 * t];
        if (g + 4 * t == kRC) v = -r_pe[NY - 1];
        U[LT * NB + LT][t] = v;
#pragma unroll
        for (int q = 0; q < NY - 1; ++q) {   // the full y tiles in front of it: their right-hand sides ride in column kRC of the tiles (NT + q, LT)
          U[(NT + q) * NB + NT + q][t] = 0.0;
          U[(NT + q) * NB + LT][t] = (j == kRC) ? ysm[YN + 16 * q + g + 4 * t] : 0.0;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#ifndef MO_SOLVE_NO_LOOKAHEAD
      // the look-ahead elimination (bit-identical results) on the 64 grid: no gain in the step kernel at three waves per SIMD (DESIGN.md
      // section 8), but the Solve kernel runs two and its cached passes are chains: 9.38 -> 9.55 M solves/s, 11.86 -> 12.03 M predictor-corrector
      bool elim_ok;
      if constexpr (NY == 1 && (NT == 4 || NT >= MO_LA_MIN_NT)) elim_ok = block_eliminate_lookahead<NT>(U, k, g, j);
      else elim_ok = block_eliminate<NT, SW, NY>(U, k, g, j);
      if (!__all(elim_ok)) { st = MO_STATUS_FACTORIZATION_FAILED; break; }   // (__all: a scalar branch, see the decision point)
#else
      if (!__all(block_eliminate<NT, SW, NY>(U, k, g, j))) { st = MO_STATUS_FACTORIZATION_FAILED; break; }
#endif
      double xb[NB];
      back_substitute<NT, NY>(U, k, j, xb);  // xb[c] = dx (permuted), xb[NT + q] = -dy
      double dyv[NY], dsv[MC], dzv[MC], ap = 1.0, ad = 1.0;
#pragma unroll
      for (int q = 0; q < NY; ++q) dyv[q] = 0.0;
      // r_pi / r_comp (qp.cc:416-417) are recomputed behind the factorisation instead of being carried across it in registers: the same
      // operations on the same operands (xs still holds this pass's x), hence the same bits.  The empty asm keeps hipcc from recognising
      // the products and holding on to the first copies.
      double r_pi2[MC], r_comp2[MC];
      auto post_cons = [&]() {   // (CLDS: also called again behind the corrector's second solve, with the registers refetched from LDS)
        cons_load(lane);
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) {
          if constexpr (!CLDS) asm volatile("" : "+v"(cs[ci]), "+v"(cz[ci]));
          if constexpr (CLDS) cs_inv[ci] = include_ineq ? rcp_f64(cs[ci]) : 1.0;   // the same reciprocal part B computed: same bits
          r_pi2[ci] = 0.0; r_comp2[ci] = 0.0;
          if (!guess_pass && lane + 64 * ci < m) {
            r_pi2[ci] = fma(ca[ci], xs[cvar[ci]], cb[ci]) - cs[ci];
            r_comp2[ci] = cs[ci] * cz[ci];
          }
        }
      };
      post_cons();
      // From a solution xb to the direction: dy, dx (natural order in LDS), ds, dz, the step lengths (qp.cc:359-363, 485-507).
      auto finish_direction = [&](double mu_s, double tau) -> bool {
        bool finite = true;
#pragma unroll
        for (int q = 0; q < NY; ++q) {
          dyv[q] = (16 * q + j < k) ? -xb[NT + q] : 0.0;
          finite = finite && (fabs(dyv[q]) < INFINITY);
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) finite = finite && (fabs(xb[c]) < INFINITY);
        if (g == 0) {
          double dxn[NT];
#pragma unroll
          for (int c = 0; c < NT; ++c) dxn[c] = xb[c];
          stv<NT, QPL>(tmp, j, dxn);  // dx, natural order
        }
        lds_fence();
        ap = 1.0; ad = 1.0;
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) {
          dsv[ci] = 0.0; dzv[ci] = 0.0;
          if (lane + 64 * ci < m) {
            dsv[ci] = ca[ci] * tmp[cvar[ci]] + r_pi2[ci];                                              // qp.cc:361
            dzv[ci] = -(cz[ci] * cs_inv[ci]) * dsv[ci] - cs_inv[ci] * (r_comp2[ci] + aff[ci] - mu_s);  // qp.cc:362
            if (cs[ci] + dsv[ci] <= 0.0 && fabs(dsv[ci]) > 0.0) ap = fmin(ap, -tau * cs[ci] * rcp_f64(dsv[ci]));  // qp.cc:498-503
            if (cz[ci] + dzv[ci] <= 0.0 && fabs(dzv[ci]) > 0.0) ad = fmin(ad, -tau * cz[ci] * rcp_f64(dzv[ci]));
            finite = finite && (fabs(dsv[ci]) < INFINITY) && (fabs(dzv[ci]) < INFINITY);
          }
        }
        if (!__all(finite)) return false;
        ap = cross_row_min(row_min(ap));
        ad = cross_row_min(row_min(ad));
        return true;
      };
      if (guess_pass) {                      // qp.cc:455-460: x, y <- the equality-constrained solution
        bool finite = true;
#pragma unroll
        for (int q = 0; q < NY; ++q) finite = finite && ((16 * q + j < k) ? (fabs(xb[NT + q]) < INFINITY) : true);
#pragma unroll
        for (int c = 0; c < NT; ++c) finite = finite && (fabs(xb[c]) < INFINITY);
        if (!__all(finite)) { st = MO_STATUS_NONFINITE; break; }
        {
          double xq[NT], yq[NY];
#pragma unroll
          for (int c = 0; c < NT; ++c) xq[c] = xb[c];
#pragma unroll
          for (int q = 0; q < NY; ++q) yq[q] = (16 * q + j < k) ? -xb[NT + q] : 0.0;
          publish_x(xq, g, j);
          publish_y(yq, g, j);
          lds_fence();
        }
        guess_pass = false;
        clamp_and_init_slacks(lane, g, j);
        lds_fence();
        continue;
      }
      if (!finish_direction(mu_step, predictor_pass ? 1.0 : 0.995)) { st = MO_STATUS_NONFINITE; break; }  // tau: qp.cc:174, 192
      double ip_mu = mu;                                                               // IPIterationOutputs::mu
      double probe_p = __builtin_nan(""), probe_d = __builtin_nan(""), mu_aff = __builtin_nan("");
      if (predictor_pass) {
        probe_p = ap; probe_d = ad;                                                    // alpha_probe, qp.cc:174
        double t_sdz = 0.0, t_zds = 0.0, t_dsdz = 0.0;
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) {
          if (lane + 64 * ci < m) {
            t_sdz = fma(cs[ci], dzv[ci], t_sdz); t_zds = fma(cz[ci], dsv[ci], t_zds); t_dsdz = fma(dsv[ci], dzv[ci], t_dsdz);
          }
          aff[ci] = dsv[ci] * dzv[ci];                                                 // delta_affine_ (qp.cc:177) enters as ds_aff dz_aff
        }
        const double sdz = wave_sum_f64(t_sdz), zds = wave_sum_f64(t_zds), dsdz = wave_sum_f64(t_dsdz);
        double ma = mu;                                                                // qp.cc:519-537
        ma += ad * sdz * inv_m;
        ma += ap * zds * inv_m;
        ma += (ad * ap) * dsdz * inv_m;
        mu_aff = ma > 0.0 ? ma : 0.0;
        const double ratio = mu_aff * rcp_f64(mu);
        const double mu_pc = (ratio * ratio * ratio) * mu;                             // qp.cc:182-183
        // The corrector solve (qp.cc:187): same matrix, right-hand side with ds_aff dz_aff and sigma mu -- pushed through the factors
        // still sitting in the tiles instead of a second factorisation.
        {   // (a zero made on the spot: hipcc hoisted the constant pair out of the problem loop and then spilled it)
          double z0;
          asm volatile("v_mov_b64 %0, 0" : "=v"(z0));
          if (lane < N / 2) { rhoS[2 * lane] = z0; rhoS[2 * lane + 1] = z0; }
        }
        lds_fence();
#pragma unroll
        for (int ci = 0; ci < MC; ++ci) {
          if (lane + 64 * ci < m) {
            const double zs = cz[ci] * cs_inv[ci];
            atomicAdd(&rhoS[cvar[ci]], ca[ci] * zs * r_pi2[ci] + ca[ci] * (r_comp2[ci] + aff[ci] - mu_pc) * cs_inv[ci]);  // qp.cc:340-341
          }
        }
        lds_fence();
        double rb[NB];
        {
          double rr[NT];
          ldv<NT, QPL>(rhoS, j, rr);
#pragma unroll
          for (int c = 0; c < NT; ++c) rb[c] = -(azS[16 * c + j] + rr[c]);   // r_d was parked in azS (position order) by part B
#pragma unroll
          for (int q = 0; q < NY; ++q) rb[NT + q] = ysm[YN + 16 * q + j];    // -r_pe, zero on lanes beyond k
        }
        solve_second_rhs<NT, NY>(U, k, g, j, rb, diagS, azS, ysm + YN, xb);  // (xp keeps x: the update below reads it)
        if constexpr (CLDS) post_cons();
        if (!finish_direction(mu_pc, 0.995)) { st = MO_STATUS_NONFINITE; break; }
        ip_mu = mu_pc;
      }
      // x,s += alpha_p (dx,ds) ; y,z += alpha_d (dy,dz), qp.cc:196-199
      {
        double xq[NT], yq[NY];
#pragma unroll
        for (int c = 0; c < NT; ++c) xq[c] = fma(xb[c], ap, xp[16 * c + j]);
#pragma unroll
        for (int q = 0; q < NY; ++q) yq[q] = fma(dyv[q], ad, ysm[16 * q + j]);
        publish_x(xq, g, j);
        publish_y(yq, g, j);
      }
#pragma unroll
      for (int ci = 0; ci < MC; ++ci) { cs[ci] = fma(dsv[ci], ap, cs[ci]); cz[ci] = fma(dzv[ci], ad, cz[ci]); }
      cons_store_state(lane);
      mu_used = mu;
      double* const iter_out = records_of(ka);
      if (iter_out && lane == 0) {   // IPIterationOutputs of this iteration (structs.hpp:53-64); its KKT norms follow at the next decision point
        double* rec = iter_out + (size_t)it * MO_ITER_RECORD;
        rec[8] = ip_mu; rec[9] = ap; rec[10] = ad;
        rec[11] = probe_p; rec[12] = probe_d; rec[13] = mu_aff;
      }
      ++it;
      lds_fence();
      if (iterate_mode) {  // outputs of Iterate: delta_ and IPIterationOutputs (structs.hpp:53-64)
        if (ka->delta) {
          double* dp = (double*)ka->delta + p * ka->delta_stride;
          for (int i = lane; i < nn; i += 64) dp[i] = tmp[i];  // dx, natural order
#pragma unroll
          for (int ci = 0; ci < MC; ++ci)
            if (lane + 64 * ci < m) { dp[nn + lane + 64 * ci] = dsv[ci]; dp[nn + m + k + lane + 64 * ci] = dzv[ci]; }
#pragma unroll
          for (int q = 0; q < NY; ++q)
            if (g == 0 && 16 * q + j < k) dp[nn + m + 16 * q + j] = dyv[q];
        }
        if (ka->ip_out && lane == 0) {
          double* ip = (double*)ka->ip_out + p * MO_IP_RECORD;
          ip[0] = ip_mu; ip[1] = ap; ip[2] = ad;  // outputs.mu = mu_input (sigma mu_input after a corrector), qp.cc:160, 183
          ip[3] = probe_p; ip[4] = probe_d; ip[5] = mu_aff;
        }
        break;
      }
    }

    // ---- outputs: state, termination, iteration count, Lagrange summary, status
    ka = fresh_args();
    {
    const int lane = lane_id(), g = lane >> 4, j = lane & 15;   // fresh copies: the per-problem ones would be carried across the whole pass loop
    double yv[NY];
#pragma unroll
    for (int q = 0; q < NY; ++q) yv[q] = ysm[16 * q + j];
    cons_load(lane);
    double* const vp = (double*)ka->vars + p * ka->vars_stride;
    if (!residual_mode) {  // the state is an input only there
      for (int i = lane; i < nn; i += 64) vp[i] = xs[i];   // x, natural order
      if (g == 0) {
#pragma unroll
        for (int q = 0; q < NY; ++q)
          if (16 * q + j < k) vp[nn + m + 16 * q + j] = yv[q];
      }
#pragma unroll
      for (int ci = 0; ci < MC; ++ci)
        if (lane + 64 * ci < m) { vp[nn + lane + 64 * ci] = cs[ci]; vp[nn + m + k + lane + 64 * ci] = cz[ci]; }
    }
    double ymin_l = INFINITY, yabs_l = INFINITY;
#pragma unroll
    for (int q = 0; q < NY; ++q) {
      ymin_l = fmin(ymin_l, (16 * q + j < k) ? yv[q] : INFINITY);
      yabs_l = fmin(yabs_l, (16 * q + j < k) ? -fabs(yv[q]) : INFINITY);
    }
    const double ymin = row_min(ymin_l), yabs = -row_min(yabs_l);
    if (lane == 0) {
      if (ka->termination) ka->termination[p] = term;
      if (ka->num_iterations) ka->num_iterations[p] = it;
      if (ka->status) ka->status[p] = st;
      if (ka->lagrange) {  // qp.cc:539-546
        ((double*)ka->lagrange)[2 * p] = k > 0 ? ymin : __builtin_nan("");
        ((double*)ka->lagrange)[2 * p + 1] = k > 0 ? yabs : __builtin_nan("");
      }
    }
    }
    lds_fence();
    if (last_of_chunk) {
      p = uniform64(next_ticket) + ticket_base;
      chunk_end = p + next_chunk;
    } else {
      ++p;
    }
  }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

#ifndef MO_FUSED_IMPL_ONLY  // kkt_fused_gather.hip includes this file for the templates only
// J-level input whose layout the 16-byte / flat streams cannot take: column-major, a leading dimension beyond n, rows that are not 16-byte
// aligned (even n), odd n beyond the flat stream's 64 -- served by the per-lane gather stream (JMODE_GATHER, kkt_fused_gather.hip).
bool fused_needs_gather(const KernelArgs& a) {
  if (!a.J) return false;
  if (!a.J_row_major || a.J_ld != a.n) return true;
  if (a.n & 1) return a.n > 64 || a.m > 64;
  return !aligned16(a.J) || (a.J_stride & 1);
}

bool fused_supported(const KernelArgs& a, int dtype) {
  if (dtype != MO_F64) return false;
  if (a.flags & ~MO_STEP_NO_INEQUALITIES) return false;
  if (a.flags && a.mode != MODE_STEP && a.mode != MODE_RESIDUAL) return false;
  if (a.mode == MODE_LINEARIZE) {  // standalone J^T J: J-level fp64; packed J on the 16-byte stream, every other layout (odd n included) on the gather stream
    return a.J && a.ticket && a.G_out && a.c_out && a.n >= 2 && a.n <= 128 && a.m_r > 0 && a.J_ld >= (a.J_row_major ? a.n : a.m_r) && a.G_out_ld >= a.n;
  }
  if (a.mode != MODE_SOLVE && a.mode != MODE_ITERATE && a.mode != MODE_STEP && a.mode != MODE_RESIDUAL) return false;
  if (a.mode == MODE_RESIDUAL && !a.r_out) return false;
  if (a.n < 2 || a.n > 128) return false;  // padded to 32 / 64 / 96 / 128 variables inside the kernel
  if (a.m < 0) return false;
  if (a.k > 31) {  // one y tile up to k = 15, two (kkt_fused_ny2.hip) up to 31, three / four (kkt_fused_ny34.hip) up to 47 / 63 on the 32 / 64 grids,
                   // and since round 4 three / four on the 96 grid (45 / 55 live tiles) and three on the 128 grid (66: it spills, and is still 2 x the generic kernel)
    if (a.k > 63) return false;                  // (m <= 256 as everywhere since round 4: four constraint slots per lane beyond 128)
    if (a.n > 96 && a.k > 47) return false;      // (the 128 grid with four y tiles would be 78 live tiles)
    if (a.mode == MODE_LINEARIZE) return false;
    // (every layout of J since round 4: the gather instantiations of kkt_fused_ny34.hip, one constraint slot per lane -- m <= 64 below)
    if (a.J && (a.n & 1) && a.m > 64) return false;
  }
  // up to four constraint slots per lane: m <= 256 (beyond 128, and beyond 64 for Solve / Iterate on the 96 / 128 grids: kkt_fused_mc4.hip)
  if (a.m > 256) return false;
  if (a.m > 128 && a.J && (a.n & 1)) return false;  // the flat-stream kernels carry one slot
  // (round 4: the two-y-tile kernels carry up to four slots on every grid, Solve / Iterate included -- a box on each of 128 variables
  // beside 16 .. 31 equalities used to fall to the generic kernel)
  if (!a.ticket || !a.vars) return false;
  if (a.mode == MODE_STEP && !a.delta) return false;
  if (a.J) {  // J-level: 16-byte pieces of a packed row-major J (even n), the flat-group stream (odd n <= 64), or the gather stream
    if (a.m_r <= 0) return false;  // m_r % 4 rows are handled after the ring stream
    if (a.J_ld < (a.J_row_major ? a.n : a.m_r)) return false;
    if (fused_needs_gather(a) && a.m > 64) return false;  // the gather instantiations carry one constraint slot per lane
  } else {    // QP-level: G, c given; no alignment requirements
    if (!a.G || !a.c || a.G_ld < a.n) return false;
  }
  return true;
}

const char* fused_name(const KernelArgs& a, int) {
#if !defined(MO_FUSED_STAMPS) && !defined(MO_GENERIC_STAMPS)
  if (fused_tiny_supported(a)) {  // the whole KKT system in one tile (kkt_fused_tiny.hip)
    const bool solve = a.mode == MODE_SOLVE || a.mode == MODE_ITERATE || a.mode == MODE_RESIDUAL;
    if (!a.J) return solve ? "fused_solve_qp_tiny_f64" : "fused_qp_tiny_f64";
    return solve ? "fused_solve_tiny_f64" : "fused_tiny_f64";
  }
#endif
  if (a.mode == MODE_SOLVE || a.mode == MODE_ITERATE || a.mode == MODE_RESIDUAL) {
    if (!a.J) return a.n > 96 ? "fused_solve_qp_f64_n128" : a.n > 64 ? "fused_solve_qp_f64_n96" : a.n > 32 ? "fused_solve_qp_f64_n64" : "fused_solve_qp_f64_n32";
    return a.n > 96 ? "fused_solve_mfma_f64_n128" : a.n > 64 ? "fused_solve_mfma_f64_n96" : a.n > 32 ? "fused_solve_mfma_f64_n64" : "fused_solve_mfma_f64_n32";
  }
  // the tile grid the problem is padded to
  if (!a.J) return a.n > 96 ? "fused_qp_f64_n128" : a.n > 64 ? "fused_qp_f64_n96" : a.n > 32 ? "fused_qp_f64_n64" : "fused_qp_f64_n32";
  return a.n > 96 ? "fused_mfma_f64_n128" : a.n > 64 ? "fused_mfma_f64_n96" : a.n > 32 ? "fused_mfma_f64_n64" : "fused_mfma_f64_n32";
}

hipError_t launch_fused(const KernelArgs& a_in, int, int num_cus, hipStream_t stream) {
  KernelArgs a = a_in;
  // static rounds up to this many problems per wave (mo_kernels.h; measured): equal-cost work (step, Iterate, residual,
  // linearisation) splits statically further than a Solve, whose problems need different numbers of passes
  if (a.static_rounds < 0) a.static_rounds = a.mode == MODE_SOLVE ? (a.n > 32 ? 2 : 6) : (a.n > 32 ? 8 : 32);
#ifdef MO_TUNING   // (A/B builds only: the product library reads no environment variable)
  static const int env_stagger = [] { const char* e = getenv("MO_FUSED_STAGGER"); return e ? atoi(e) : -1; }();
  static const int env_wps = [] { const char* e = getenv("MO_FUSED_WPS"); return e ? atoi(e) : 0; }();
  static const int env_sw = [] { const char* e = getenv("MO_FUSED_SWEEP"); return e ? atoi(e) : -1; }();
#else
  constexpr int env_stagger = -1, env_wps = 0, env_sw = -1;
#endif
  // the 64-variable grid of the step kernel with J-level input (the BASELINE configs[2] / [4] shape); measured neutral elsewhere
  const bool headline_shape = a.mode == MODE_STEP && a.J && a.n > 32 && a.n <= 64;
  a.stagger = env_stagger >= 0 ? (env_stagger & 0xff) : (headline_shape ? 4 : 0);
  a.chain_prio = env_stagger >= 0 ? ((env_stagger >> 8) & 1) : (headline_shape ? 1 : 0);   // MO_FUSED_STAGGER = units + 256 * priority
  // Waves per SIMD the kernel is register-budgeted for (defaults picked from measurements; MO_FUSED_WPS in -DMO_TUNING builds).
  // 64 grid: 3 (A/B in DESIGN.md).  32 grid: FOUR since round 4 (the kernel needs 100 VGPRs).  Round 3 had measured a fourth wave 10 % slower at
  // BASELINE configs[1] (0.136 vs 0.123 ms) -- when launches still took a ticket per problem; with static rounds it is faster at every batch
  // size: configs[1] 86.4 / 87.1 -> 94.9 / 96.4 M steps/s (4 096 problems are ONE round of 4 096 waves), batch 65 536: 140.6 -> 152.9 M.
  const int wps = a.n > 32 ? (env_wps == 2 ? 2 : 3) : (env_wps == 3 ? 3 : 4);
  const int sw = (env_sw >= 0 && env_sw <= 6) ? env_sw : 3;
  long long grid = num_cus;  // one workgroup of 4*wps waves per CU; problems are pulled from the ticket counter
  const long long blocks_needed = (a.batch + 3) / 4;
  if (grid > blocks_needed) grid = blocks_needed;
  if (grid < 1) grid = 1;
  // The work counter is zeroed on the stream in front of the kernel -- unless the launch is certain to run in static rounds, which never touch
  // it: every kernel below has at least 4 waves per workgroup and min(CUs, ceil(batch / 4)) workgroups, so batch <= rounds x 4 x workgroups
  // is static whatever the instantiation (the kernels test batch <= rounds x waves).  One enqueued operation less per small launch.
  {
    long long wgs = (a.batch + 3) / 4;
    if (wgs > num_cus) wgs = num_cus;
#if !defined(MO_FUSED_STAMPS) && !defined(MO_GENERIC_STAMPS)
    const bool tickets_always = fused_tiny_supported(a);   // the one-tile kernel hands out tickets of up to 64 problems whatever the size
#else
    const bool tickets_always = false;
#endif
    const bool surely_static = !tickets_always && a.static_rounds > 0 && a.batch <= (long long)a.static_rounds * 4 * wgs;
    if (!surely_static) {
      hipError_t e = hipMemsetAsync(a.ticket, 0, sizeof(unsigned long long), stream);
      if (e != hipSuccess) return e;
    }
  }
#if !defined(MO_FUSED_STAMPS) && !defined(MO_GENERIC_STAMPS)  // (the diagnostic builds of tools/phase_timer*.hip link this file alone)
  if (fused_tiny_supported(a)) return launch_fused_tiny(a, num_cus, stream);
  if (a.mode != MODE_LINEARIZE && a.k <= 15 && !fused_needs_gather(a) && !(a.J && (a.n & 1)) &&
      (a.m > 128 || (a.m > 64 && a.n > 64 && a.mode != MODE_STEP)))
    return launch_fused_mc4(a, num_cus, stream);
  if (a.mode != MODE_LINEARIZE && a.k > 31) return launch_fused_ny34(a, num_cus, stream);
  if (a.mode != MODE_LINEARIZE && a.k > 15) return launch_fused_ny2(a, num_cus, stream);
  if (fused_needs_gather(a) || (a.mode == MODE_LINEARIZE && (a.n & 1))) return launch_fused_gather(a, num_cus, stream);
#endif
  if (a.mode == MODE_LINEARIZE) {
    const int wl = a.n > 96 ? 1 : (a.n > 64 ? 2 : 3);
    long long lgrid = num_cus;
    const long long lneed = (a.batch + 3) / 4;
    if (lgrid > lneed) lgrid = lneed;
    if (lgrid < 1) lgrid = 1;
    const dim3 lgd((unsigned)lgrid), lbd(256 * wl);
    if (a.n > 96) hipLaunchKernelGGL((kkt_fused_linearize_kernel<8, 1>), lgd, lbd, 0, stream, a);
    else if (a.n > 64) hipLaunchKernelGGL((kkt_fused_linearize_kernel<6, 2>), lgd, lbd, 0, stream, a);
    else if (a.n > 32) hipLaunchKernelGGL((kkt_fused_linearize_kernel<4, 3>), lgd, lbd, 0, stream, a);
    else hipLaunchKernelGGL((kkt_fused_linearize_kernel<2, 3>), lgd, lbd, 0, stream, a);
    return hipGetLastError();
  }
  if (a.J && (a.n & 1)) {  // odd n: flat-group J stream
    const bool solve = a.mode == MODE_SOLVE || a.mode == MODE_ITERATE || a.mode == MODE_RESIDUAL;
    const int wf = (a.n > 32 && solve) ? 2 : 3;
    long long fgrid = num_cus;
    const long long fneed = (a.batch + 3) / 4;
    if (fgrid > fneed) fgrid = fneed;
    if (fgrid < 1) fgrid = 1;
    const dim3 fgd((unsigned)fgrid), fbd(256 * wf);
    if (solve) {
      if (a.n > 32) hipLaunchKernelGGL((kkt_fused_solve_kernel<4, 2, 3, false, 1, true>), fgd, fbd, 0, stream, a);
      else hipLaunchKernelGGL((kkt_fused_solve_kernel<2, 3, 3, false, 1, true>), fgd, fbd, 0, stream, a);
    } else {
      if (a.n > 32) hipLaunchKernelGGL((kkt_fused_f64_kernel<4, 3, 3, false, 1, true>), fgd, fbd, 0, stream, a);
      else hipLaunchKernelGGL((kkt_fused_f64_kernel<2, 3, 3, false, 1, true>), fgd, fbd, 0, stream, a);
    }
    return hipGetLastError();
  }
  if (a.mode == MODE_STEP && a.m > 64) {  // two constraint slots per lane (a box on every one of 64 variables is m = 128)
    const int wq = a.n > 96 ? 1 : (a.n > 32 ? 2 : 3);
    long long qgrid = num_cus;
    const long long qneed = (a.batch + 3) / 4;
    if (qgrid > qneed) qgrid = qneed;
    if (qgrid < 1) qgrid = 1;
    const dim3 qgd((unsigned)qgrid), qbd(256 * wq);
#define MO_FUSED_MC2(NT_, WPS_)                                                                                      \
  do {                                                                                                               \
    if (a.J) hipLaunchKernelGGL((kkt_fused_f64_kernel<NT_, WPS_, 3, false, 2>), qgd, qbd, 0, stream, a);             \
    else hipLaunchKernelGGL((kkt_fused_f64_kernel<NT_, WPS_, 3, true, 2>), qgd, qbd, 0, stream, a);                  \
  } while (0)
    if (a.n > 96) MO_FUSED_MC2(8, 1);
    else if (a.n > 64) MO_FUSED_MC2(6, 2);
    else if (a.n > 32) MO_FUSED_MC2(4, 2);
    else MO_FUSED_MC2(2, 3);
#undef MO_FUSED_MC2
    return hipGetLastError();
  }
  if (a.m > 64) {  // Solve / Iterate with two constraint slots per lane (n <= 64, checked by fused_supported)
    const int wq = a.n > 32 ? 2 : 3;
    long long qgrid = num_cus;
    const long long qneed = (a.batch + 3) / 4;
    if (qgrid > qneed) qgrid = qneed;
    if (qgrid < 1) qgrid = 1;
    const dim3 qgd((unsigned)qgrid), qbd(256 * wq);
    // (J-level on the 64 grid: the instantiation with the corrector's code needs 20 B of scratch, the one without none -- picked by strategy)
    const bool pc = (a.mode == MODE_SOLVE ? a.sp.barrier_strategy : a.barrier_strategy) == MO_PREDICTOR_CORRECTOR && a.mode != MODE_RESIDUAL;
    if (a.n > 32) {
      if (a.J && !pc) hipLaunchKernelGGL((kkt_fused_solve_kernel<4, 2, 3, false, 2, JMODE_VECTOR, 1, false>), qgd, qbd, 0, stream, a);
      else if (a.J) hipLaunchKernelGGL((kkt_fused_solve_kernel<4, 2, 3, false, 2>), qgd, qbd, 0, stream, a);
      else hipLaunchKernelGGL((kkt_fused_solve_kernel<4, 2, 3, true, 2>), qgd, qbd, 0, stream, a);
    } else {
      if (a.J) hipLaunchKernelGGL((kkt_fused_solve_kernel<2, 3, 3, false, 2>), qgd, qbd, 0, stream, a);
      else hipLaunchKernelGGL((kkt_fused_solve_kernel<2, 3, 3, true, 2>), qgd, qbd, 0, stream, a);
    }
    return hipGetLastError();
  }
  if (a.n > 64) {  // 96 / 128-variable tile grids: correctness-first instantiations (the 128 one spills), one or two waves per SIMD
    const bool big = a.n > 96, solve = a.mode == MODE_SOLVE || a.mode == MODE_ITERATE || a.mode == MODE_RESIDUAL;
    const int bw = (!big && !solve) ? 2 : 1;
    long long bgrid = num_cus;
    const long long bneed = (a.batch + 3) / 4;
    if (bgrid > bneed) bgrid = bneed;
    if (bgrid < 1) bgrid = 1;
    const dim3 bgd((unsigned)bgrid), bbd(256 * bw);
    if (solve) {
      if (big) { if (a.J) hipLaunchKernelGGL((kkt_fused_solve_kernel<8, 1, 3, false>), bgd, bbd, 0, stream, a); else hipLaunchKernelGGL((kkt_fused_solve_kernel<8, 1, 3, true>), bgd, bbd, 0, stream, a); }
      else { if (a.J) hipLaunchKernelGGL((kkt_fused_solve_kernel<6, 1, 3, false>), bgd, bbd, 0, stream, a); else hipLaunchKernelGGL((kkt_fused_solve_kernel<6, 1, 3, true>), bgd, bbd, 0, stream, a); }
    } else {
      if (big) { if (a.J) hipLaunchKernelGGL((kkt_fused_f64_kernel<8, 1, 3, false>), bgd, bbd, 0, stream, a); else hipLaunchKernelGGL((kkt_fused_f64_kernel<8, 1, 3, true>), bgd, bbd, 0, stream, a); }
      else { if (a.J) hipLaunchKernelGGL((kkt_fused_f64_kernel<6, 2, 3, false>), bgd, bbd, 0, stream, a); else hipLaunchKernelGGL((kkt_fused_f64_kernel<6, 2, 3, true>), bgd, bbd, 0, stream, a); }
    }
    return hipGetLastError();
  }
  if (a.mode == MODE_SOLVE || a.mode == MODE_ITERATE || a.mode == MODE_RESIDUAL) {  // register budget of the Solve kernel: 2 waves per SIMD at n = 64, 3 at n = 32
    const int swps = a.n > 32 ? 2 : 3;
    long long sgrid = num_cus;
    const long long need = (a.batch + 3) / 4;
    if (sgrid > need) sgrid = need;
    if (sgrid < 1) sgrid = 1;
    const dim3 sgd((unsigned)sgrid), sbd(256 * swps);
    if (a.n > 32) {
      if (a.J) hipLaunchKernelGGL((kkt_fused_solve_kernel<4, 2, 3, false>), sgd, sbd, 0, stream, a);
      else hipLaunchKernelGGL((kkt_fused_solve_kernel<4, 2, 3, true>), sgd, sbd, 0, stream, a);
    } else {
      if (a.J) hipLaunchKernelGGL((kkt_fused_solve_kernel<2, 3, 3, false>), sgd, sbd, 0, stream, a);
      else hipLaunchKernelGGL((kkt_fused_solve_kernel<2, 3, 3, true>), sgd, sbd, 0, stream, a);
    }
    return hipGetLastError();
  }
  const dim3 gd((unsigned)grid), bd(256 * wps);
#ifndef MO_TUNING
  // The product build carries ONE instantiation per tile grid and input level: three waves per SIMD (four on the 32 grid), the lean sweep (SW = 3).  The other
  // waves-per-SIMD / sweep flavours DESIGN.md section 8 measured and rejected are instantiated in -DMO_TUNING builds only.
  (void)sw;
  if (a.n > 32) {
    if (a.J) hipLaunchKernelGGL((kkt_fused_f64_kernel<4, 3, 3, false>), gd, bd, 0, stream, a);
    else hipLaunchKernelGGL((kkt_fused_f64_kernel<4, 3, 3, true>), gd, bd, 0, stream, a);
  } else {
    if (a.J) hipLaunchKernelGGL((kkt_fused_f64_kernel<2, 4, 3, false>), gd, bd, 0, stream, a);
    else hipLaunchKernelGGL((kkt_fused_f64_kernel<2, 4, 3, true>), gd, bd, 0, stream, a);
  }
  return hipGetLastError();
#else
  if (!a.J) {  // QP-level input: default flavour only
    if (a.n > 32) {
      if (wps == 3) hipLaunchKernelGGL((kkt_fused_f64_kernel<4, 3, 3, true>), gd, bd, 0, stream, a);
      else hipLaunchKernelGGL((kkt_fused_f64_kernel<4, 2, 3, true>), gd, bd, 0, stream, a);
    } else {
      if (wps == 3) hipLaunchKernelGGL((kkt_fused_f64_kernel<2, 3, 3, true>), gd, bd, 0, stream, a);
      else hipLaunchKernelGGL((kkt_fused_f64_kernel<2, 4, 3, true>), gd, bd, 0, stream, a);
    }
    return hipGetLastError();
  }
#define MO_FUSED_LAUNCH(NT_, WPS_, SW_) hipLaunchKernelGGL((kkt_fused_f64_kernel<NT_, WPS_, SW_, false>), gd, bd, 0, stream, a)
// MO_FUSED_SWEEP = 5 (fused-broadcast sweep) / 6 (look-ahead elimination): measured and rejected; they are only
// instantiated in builds with -DMO_FUSED_EXPERIMENTS (A/B runs), the product build carries neither.
#ifdef MO_FUSED_EXPERIMENTS
#define MO_FUSED_EXPERIMENTAL(NT_, WPS_)              \
    else if (sw == 5) MO_FUSED_LAUNCH(NT_, WPS_, 5);  \
    else if (sw == 6) MO_FUSED_LAUNCH(NT_, WPS_, 6);
#else
#define MO_FUSED_EXPERIMENTAL(NT_, WPS_)
#endif
#define MO_FUSED_BY_SW(NT_, WPS_)                     \
  do {                                                \
    if (sw == 1) MO_FUSED_LAUNCH(NT_, WPS_, 1);       \
    MO_FUSED_EXPERIMENTAL(NT_, WPS_)                  \
    else MO_FUSED_LAUNCH(NT_, WPS_, 3);               \
  } while (0)
  if (a.n > 32) {
    if (wps == 3) MO_FUSED_BY_SW(4, 3); else MO_FUSED_BY_SW(4, 2);
  } else {
    if (wps == 3) MO_FUSED_BY_SW(2, 3); else MO_FUSED_BY_SW(2, 4);
  }
#undef MO_FUSED_BY_SW
#undef MO_FUSED_LAUNCH
  return hipGetLastError();
#endif  // MO_TUNING
}

#endif  // MO_FUSED_IMPL_ONLY

}  // namespace mo
