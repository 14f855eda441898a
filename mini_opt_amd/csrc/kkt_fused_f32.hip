// kkt_fused_f32.hip -- fused single-wave KKT Newton step in fp32 for n = 64 / 128 (BASELINE configs[3]: n = 128, 16 equalities,
// 64 box entries, m_r = 256).  Same construction as the fp64 kernel (kkt_fused.hip): one wavefront owns one QP, the reduced
// KKT matrix lives in VGPRs as 16x16 tiles in the v_mfma_f32_16x16x4_f32 C/D layout, J streams once through an LDS-DMA ring
// straight into MFMA operands, block LDL^T with 16x16 pivot blocks inverted by symmetric sweeps, x+ formulation.
// What differs from fp64:
//   * C/D fragment layout: lane (g = l >> 4, j = l & 15), register t <-> element (row 4g + t, col j)   (fp64: row g + 4t);
//   * a 16-byte J piece carries FOUR columns, so the variable permutation is position 16c + i <-> column 64(c>>2) + 4i + (c&3);
//   * k may be 16 (all of the y tile), so the right-hand side rides in a tile column of its own (index NT + 1, column 0)
//     instead of column 15 of the [A_eq^T | rhs] tiles: (NT+1)(NT+2)/2 + NT + 1 tiles of 4 VGPRs (216 VGPRs at n = 128).
// Reference lines: residual.hpp:206-224 + nonlinear.cc:187-189 (J^T J, J^T r, lambda), qp.cc:281-298 (assembly),
// qp.cc:302-311 + 318-364 (factorisation and solve; no explicit inverse of H here), qp.cc:485-507 (alpha).
#include <stdlib.h>

#include "mo_kernels.h"

namespace mo {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// Phase stamps exist only in the diagnostic build of tools/phase_timer_f32.hip; the product kernel executes none.
#ifdef MO_F32_STAMPS
#define MO_STAMP32(i)                                                                                 \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    unsigned long long t__;                                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                       \
    stamp_acc[i] += t__ - stamp_prev;                                                                 \
    stamp_prev = t__;                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#else
#define MO_STAMP32(i) do { } while (0)
#endif
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ inline int lane_id32() {  // volatile on purpose: nothing derived from it is hoisted out of the problem loop
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
typedef const KernelArgs __attribute__((address_space(4)))* KArgs32;
__device__ inline KArgs32 fresh_args32() {
  KArgs32 p = (KArgs32)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
__device__ inline float readlane_f32(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ inline float bpermute_f32(int byte_addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v))); }
template <int CTRL> __device__ inline float dpp_f32(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, false)); }
template <int LANE_IN_ROW> __device__ inline float row_bcast_f32(float v) { return dpp_f32<0x150 + LANE_IN_ROW>(v); }
__device__ inline float row_sum_f32(float v) {  // over the 16 lanes of each row: quad_perm, quad_perm, row_half_mirror, row_mirror
  v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v); v += dpp_f32<0x140>(v);
  return v;
}
__device__ inline float row_min_f32(float v) {
  v = fminf(v, dpp_f32<0xB1>(v)); v = fminf(v, dpp_f32<0x4E>(v)); v = fminf(v, dpp_f32<0x141>(v)); v = fminf(v, dpp_f32<0x140>(v));
  return v;
}
// v_permlane16_swap(a, b) -> {[a.R0, b.R0, a.R2, b.R2], [a.R1, b.R1, a.R3, b.R3]}; v_permlane32_swap(a, b) ->
// {[a.R0, a.R1, b.R0, b.R1], [a.R2, a.R3, b.R2, b.R3]} (tools/microbench.hip)
__device__ inline float cross_row_sum_f32(float v) {
  const u2v p = __builtin_amdgcn_permlane16_swap((unsigned)__float_as_int(v), (unsigned)__float_as_int(v), false, false);
  const float s = __int_as_float((int)p[0]) + __int_as_float((int)p[1]);
  const u2v q = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(s), (unsigned)__float_as_int(s), false, false);
  return __int_as_float((int)q[0]) + __int_as_float((int)q[1]);
}
__device__ inline float cross_row_min_f32(float v) {
  const u2v p = __builtin_amdgcn_permlane16_swap((unsigned)__float_as_int(v), (unsigned)__float_as_int(v), false, false);
  const float s = fminf(__int_as_float((int)p[0]), __int_as_float((int)p[1]));
  const u2v q = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(s), (unsigned)__float_as_int(s), false, false);
  return fminf(__int_as_float((int)q[0]), __int_as_float((int)q[1]));
}
__device__ inline float rcp_f32(float d) {
  float q = __builtin_amdgcn_rcpf(d);
  q = fmaf(q, fmaf(-d, q, 1.0f), q);
  return q;
}
__device__ inline f4 mfma4_f32(const f4& a, const f4& b, f4 c) {  // c += A^T-fragment(a) * fragment(b) over the 16 tile rows
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
  return c;
}
// LDS-DMA (no VGPR destination; completion is waited for by hand, loads retire in order)
__device__ inline void dma16_f32(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ inline void dma4_f32(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// `count` consecutive 4-byte words (count <= 128, wave-uniform) global -> LDS
__device__ inline void dma_words(const void* src, unsigned lds_dst, int count, int lane) {
  const char* s4 = reinterpret_cast<const char*>(src) + 4 * lane;
  if (lane < count) dma4_f32(s4, lds_dst);
  if (lane + 64 < count) dma4_f32(s4 + 256, lds_dst + 256);
}
template <int N> __device__ inline void wait_vmcnt32() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ inline void lds_fence32() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// EXEC-masked moves with immediate lane masks (see kkt_fused.hip for why the masks are immediates, and for the pad: a DPP read of the VGPR
// just written needs 2 wait states -- the s_mov_b64 that restores EXEC and `s_nop 0`; rounds 1-2 padded with `s_nop 4`, -DMO_F32_MASKED_PAD_4
// restores that for A/B builds)
#ifdef MO_F32_MASKED_PAD_4
#define MO_F32_PAD "4"
#else
#define MO_F32_PAD "0"
#endif
#ifndef MO_F32_LOOKAHEAD
#define MO_F32_LOOKAHEAD 0   // A/B knob: 1 = look-ahead elimination (measured: -2.2 % at BASELINE configs[3])
#endif
#ifndef MO_F32_RHS_VECTOR   // A/B knob (step kernel): 0 = the right-hand side in tile column NT + 1, as in rounds 1 - 3
#define MO_F32_RHS_VECTOR (!MO_F32_LOOKAHEAD)   // round 4: a vector (registers + two LDS hops per block step): -180 MFMAs at NT = 8, +1.0 %
#endif
#if MO_F32_RHS_VECTOR && MO_F32_LOOKAHEAD
#error "the look-ahead elimination schedules the MFMAs of tile column NT + 1: build it with -DMO_F32_RHS_VECTOR=0"
#endif
template <unsigned long long MASK> __device__ inline void masked_set_f32(float& dst, float src) {
  unsigned long long save;
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_mov_b32 %[d], %[s]\n\t"
               "s_mov_b64 exec, %[sv]\n\ts_nop " MO_F32_PAD
               : [d] "+v"(dst), [sv] "=&s"(save)
               : [s] "v"(src), [lo] "i"((unsigned)(MASK & 0xffffffffull)), [hi] "i"((unsigned)(MASK >> 32)));
}
template <unsigned long long MASK> __device__ inline void masked_set_neg_f32(float& dst, float src) {
  unsigned long long save;
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_max_f32 %[d], -%[s], -%[s]\n\t"
               "s_mov_b64 exec, %[sv]\n\ts_nop " MO_F32_PAD
               : [d] "+v"(dst), [sv] "=&s"(save)
               : [s] "v"(src), [lo] "i"((unsigned)(MASK & 0xffffffffull)), [hi] "i"((unsigned)(MASK >> 32)));
}
template <unsigned long long MASK> __device__ inline void masked_zero4_f32(f4& T) {
  unsigned long long save;
  float t0 = T[0], t1 = T[1], t2 = T[2], t3 = T[3];
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_mov_b32 %[a], 0\n\tv_mov_b32 %[b], 0\n\t"
               "v_mov_b32 %[c], 0\n\tv_mov_b32 %[d], 0\n\ts_mov_b64 exec, %[sv]\n\ts_nop " MO_F32_PAD
               : [a] "+v"(t0), [b] "+v"(t1), [c] "+v"(t2), [d] "+v"(t3), [sv] "=&s"(save)
               : [lo] "i"((unsigned)(MASK & 0xffffffffull)), [hi] "i"((unsigned)(MASK >> 32)));
  T[0] = t0; T[1] = t1; T[2] = t2; T[3] = t3;
}

// One pivot of the symmetric sweep of a 16x16 tile (row K lives in lane row K >> 2, register K & 3).  Afterwards the swept
// block holds -T11^-1, the swept x unswept block T11^-1 T12, the unswept block the Schur complement.
template <int K>
__device__ inline void sweep_step_f32(f4& T, float& bad, int j) {
  constexpr int src_g = K >> 2, src_t = K & 3;
  constexpr unsigned long long mcol = 0x0001000100010001ull << K;  // the four lanes of tile column K
  constexpr unsigned long long mrow = 0xFFFFull << (16 * src_g);   // the 16 lanes of the row group that holds row K
  const float rowreg = T[src_t];
  const float d = readlane_f32(rowreg, 16 * src_g + K);
  float inv = __builtin_amdgcn_rcpf(d);
  inv = fmaf(inv, fmaf(-d, inv, 1.0f), inv);
  asm volatile("v_fma_f32 %0, %1, 0, %0" : "+v"(bad) : "v"(inv));  // NaN at a zero / NaN pivot; pinned (see kkt_fused.hip)
  float rk = bpermute_f32((16 * src_g + j) * 4, rowreg) * inv;     // T(K, j) / d in every row
  masked_set_neg_f32<mcol>(rk, inv);
  float f[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) f[t] = row_bcast_f32<K>(T[t]);        // T(4g + t, K): column K of this lane's own rows
  masked_zero4_f32<mcol>(T);
#pragma unroll
  for (int t = 0; t < 4; ++t) T[t] = fmaf(-f[t], rk, T[t]);
  float rowk_new = T[src_t];
  masked_set_f32<mrow>(rowk_new, rk);
  T[src_t] = rowk_new;
}
template <int K, int KEND> struct SweepLoop32 {
  static __device__ inline void run(f4& T, float& bad, int npiv, int j) {
    if (K < npiv) sweep_step_f32<K>(T, bad, j);  // wave-uniform
    SweepLoop32<K + 1, KEND>::run(T, bad, npiv, j);
  }
};
template <int KEND> struct SweepLoop32<KEND, KEND> {
  static __device__ inline void run(f4&, float&, int, int) {}
};
__device__ inline bool sweep_tile_f32(f4& T, int npiv, int j) {
  float bad = 0.0f;
  SweepLoop32<0, 16>::run(T, bad, npiv, j);
  return bad == 0.0f;
}

// ---- look-ahead elimination of the step kernel ----------------------------------------------------------------------------------
// Program order "sweep tile (pa, pa), then every update of block step pa" makes a wave alternate between a dependent VALU chain that
// leaves the matrix pipe idle (16 pivots of ~150 cycles each) and a run of MFMAs that leaves the VALU idle -- and the fp32 MFMA, unlike
// the fp64 one, does not share the VALU's datapath, so the two can run side by side inside ONE wave.  The update of block step pa is
// split: the column of tile (pa+1, pa+1) first, then the sweep of that tile interleaved pivot by pivot with the remaining tile products
// of step pa (they touch neither that tile nor its operands).  Work items (one item = one 16x16x16 tile product = four MFMAs) stay in
// program order -- all four MFMAs of an item before the next one: the updates of a panel need its finished -Z, the next panel product
// overwrites it -- so every tile sees the same operations in the same order as in the plain loop: bit-identical results.
// Tile columns: x blocks 0 .. NT-1, y block NT, right-hand side NR = NT + 1 (NB = NT + 2); rows pb <= NT.
template <int NT> constexpr int la32_rows(int pa, int pc) { return (pc < NT ? pc : NT) - pa; }   // tiles (pa+1 .. min(pc, NT), pc) updated at step pa
template <int NT> constexpr int la32_count(int pa) { int c = 0; for (int pc = pa + 2; pc < NT + 2; ++pc) c += 1 + la32_rows<NT>(pa, pc); return c; }
template <int NT> constexpr int la32_pc(int pa, int idx) {
  for (int pc = pa + 2; pc < NT + 2; ++pc) { const int n = 1 + la32_rows<NT>(pa, pc); if (idx < n) return pc; idx -= n; }
  return -1;
}
template <int NT> constexpr int la32_sub(int pa, int idx) {  // 0: the panel product -Z = (-T^-1) U_ac;  s >= 1: update of tile (pa + s, pc)
  for (int pc = pa + 2; pc < NT + 2; ++pc) { const int n = 1 + la32_rows<NT>(pa, pc); if (idx < n) return idx; idx -= n; }
  return -1;
}
template <int NT> constexpr int la32_per_pivot(int pa) { return (la32_count<NT>(pa) + 15) / 16; }
// the Q-th of the four MFMAs of work item IDX of block step PA
template <int NT, int PA, int IDX, int Q> __device__ inline void la32_mfma(f4 (&U)[(NT + 2) * (NT + 2)], f4& negZ) {
  constexpr int NB = NT + 2;
  if constexpr (IDX < la32_count<NT>(PA)) {
    constexpr int pc = la32_pc<NT>(PA, IDX), sub = la32_sub<NT>(PA, IDX);
    if constexpr (sub == 0) {
      if constexpr (Q == 0) negZ = f4{0.0f, 0.0f, 0.0f, 0.0f};
      negZ = __builtin_amdgcn_mfma_f32_16x16x4f32(U[PA * NB + PA][Q], U[PA * NB + pc][Q], negZ, 0, 0, 0);
    } else {
      U[(PA + sub) * NB + pc] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[PA * NB + (PA + sub)][Q], negZ[Q], U[(PA + sub) * NB + pc], 0, 0, 0);
    }
  }
}
// What goes to wait point P (0 .. 3) of pivot K's chain: with one item per pivot its P-th MFMA; with M > 1 items per pivot (pivot K owns
// items K M .. K M + M - 1) whole items, spread over the points in order.
template <int NT, int PA, int K, int P> __device__ inline void la32_point(f4 (&U)[(NT + 2) * (NT + 2)], f4& negZ) {
  constexpr int M = la32_per_pivot<NT>(PA);
  static_assert(M <= 4, "work items per pivot");
  if constexpr (M <= 1) {
    la32_mfma<NT, PA, K, P>(U, negZ);
  } else {
    constexpr int slot = M == 2 ? (P == 0 ? 0 : (P == 2 ? 1 : -1)) : (P < M ? P : -1);
    if constexpr (slot >= 0) {
      la32_mfma<NT, PA, K * M + slot, 0>(U, negZ); la32_mfma<NT, PA, K * M + slot, 1>(U, negZ);
      la32_mfma<NT, PA, K * M + slot, 2>(U, negZ); la32_mfma<NT, PA, K * M + slot, 3>(U, negZ);
    }
  }
}
template <int NT, int PA, int K>
__device__ inline void sweep_step_work_f32(f4 (&U)[(NT + 2) * (NT + 2)], f4& negZ, float& bad, bool active, int j) {
  constexpr int NB = NT + 2;
  constexpr int src_g = K >> 2, src_t = K & 3;
  constexpr unsigned long long mcol = 0x0001000100010001ull << K;
  constexpr unsigned long long mrow = 0xFFFFull << (16 * src_g);
  f4& T = U[(PA + 1) * NB + (PA + 1)];
  if (!active) {  // wave-uniform: pivots beyond the y tile's k rows -- only the work
    la32_point<NT, PA, K, 0>(U, negZ); la32_point<NT, PA, K, 1>(U, negZ); la32_point<NT, PA, K, 2>(U, negZ); la32_point<NT, PA, K, 3>(U, negZ);
    return;
  }
  const float rowreg = T[src_t];
  const float d = readlane_f32(rowreg, 16 * src_g + K);
  float inv = __builtin_amdgcn_rcpf(d);
  const float rowk = bpermute_f32((16 * src_g + j) * 4, rowreg);
  __builtin_amdgcn_sched_barrier(0);
  la32_point<NT, PA, K, 0>(U, negZ);                       // covers v_rcp_f32 and the LDS round trip of the row broadcast
  __builtin_amdgcn_sched_barrier(0);
  inv = fmaf(inv, fmaf(-d, inv, 1.0f), inv);
  asm volatile("v_fma_f32 %0, %1, 0, %0" : "+v"(bad) : "v"(inv));
  float f[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) f[t] = row_bcast_f32<K>(T[t]);
  __builtin_amdgcn_sched_barrier(0);
  la32_point<NT, PA, K, 1>(U, negZ);
  __builtin_amdgcn_sched_barrier(0);
  float rk = rowk * inv;
  masked_set_neg_f32<mcol>(rk, inv);
  masked_zero4_f32<mcol>(T);
  __builtin_amdgcn_sched_barrier(0);
  la32_point<NT, PA, K, 2>(U, negZ);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) T[t] = fmaf(-f[t], rk, T[t]);
  float rowk_new = T[src_t];
  masked_set_f32<mrow>(rowk_new, rk);
  T[src_t] = rowk_new;
  __builtin_amdgcn_sched_barrier(0);
  la32_point<NT, PA, K, 3>(U, negZ);
  __builtin_amdgcn_sched_barrier(0);
}
template <int NT, int PA, int K> struct SweepWithWork32 {
  static __device__ inline void run(f4 (&U)[(NT + 2) * (NT + 2)], f4& negZ, float& bad, int npiv, int j) {
    sweep_step_work_f32<NT, PA, K>(U, negZ, bad, K < npiv, j);
    SweepWithWork32<NT, PA, K + 1>::run(U, negZ, bad, npiv, j);
  }
};
template <int NT, int PA> struct SweepWithWork32<NT, PA, 16> {
  static __device__ inline void run(f4 (&)[(NT + 2) * (NT + 2)], f4&, float&, int, int) {
    static_assert(la32_count<NT>(PA) <= 16 * la32_per_pivot<NT>(PA), "work items per pivot");
  }
};
template <int NT, int PA> struct LookAheadSteps32 {
  static __device__ inline void run(f4 (&U)[(NT + 2) * (NT + 2)], float& bad, int k, int j) {
    constexpr int NB = NT + 2;
    // the column of the next diagonal tile first
    f4 negZ = mfma4_f32(U[PA * NB + PA], U[PA * NB + PA + 1], f4{0.0f, 0.0f, 0.0f, 0.0f});
    U[(PA + 1) * NB + PA + 1] = mfma4_f32(U[PA * NB + PA + 1], negZ, U[(PA + 1) * NB + PA + 1]);
    __builtin_amdgcn_sched_barrier(0);
    SweepWithWork32<NT, PA, 0>::run(U, negZ, bad, PA + 1 < NT ? 16 : k, j);
    if constexpr (PA + 1 < NT) LookAheadSteps32<NT, PA + 1>::run(U, bad, k, j);
  }
};
// (block step NT -- the y tile -- has no update left: the back-substitution applies its swept tile itself)
template <int NT>
__device__ inline bool block_eliminate_lookahead_f32(f4 (&U)[(NT + 2) * (NT + 2)], int k, int j) {
  float bad = 0.0f;
  SweepLoop32<0, 16>::run(U[0], bad, 16, j);
  __builtin_amdgcn_sched_barrier(0);
  LookAheadSteps32<NT, 0>::run(U, bad, k, j);
  return bad == 0.0f;
}

// natural-order array <-> value at tile position 16c + j (position 16c + i <-> variable 64(c>>2) + 4i + (c&3))
template <int NT> __device__ inline void ldv32(const float* arr, int j, float (&v)[NT]) {
#pragma unroll
  for (int h = 0; h < NT / 4; ++h) {
    const f4 t = *(const f4*)(arr + 64 * h + 4 * j);
    v[4 * h] = t[0]; v[4 * h + 1] = t[1]; v[4 * h + 2] = t[2]; v[4 * h + 3] = t[3];
  }
}
template <int NT> __device__ inline void stv32(float* arr, int j, const float (&v)[NT]) {
#pragma unroll
  for (int h = 0; h < NT / 4; ++h) {
    f4 t; t[0] = v[4 * h]; t[1] = v[4 * h + 1]; t[2] = v[4 * h + 2]; t[3] = v[4 * h + 3];
    *(f4*)(arr + 64 * h + 4 * j) = t;
  }
}

// the same on a caller's array of nn <= 16 NT entries (nn a multiple of 4: a lane's four variables are all inside or all outside)
template <int NT> __device__ inline void stv32_n(float* arr, int j, int nn, const float (&v)[NT]) {
#pragma unroll
  for (int h = 0; h < NT / 4; ++h) {
    f4 t; t[0] = v[4 * h]; t[1] = v[4 * h + 1]; t[2] = v[4 * h + 2]; t[3] = v[4 * h + 3];
    if (64 * h + 4 * j < nn) *(f4*)(arr + 64 * h + 4 * j) = t;
  }
}
template <int NT> __device__ inline void ldv32_n(const float* arr, int j, int nn, float (&v)[NT]) {
#pragma unroll
  for (int h = 0; h < NT / 4; ++h) {
    f4 t = f4{0.0f, 0.0f, 0.0f, 0.0f};
    if (64 * h + 4 * j < nn) t = *(const f4*)(arr + 64 * h + 4 * j);
    v[4 * h] = t[0]; v[4 * h + 1] = t[1]; v[4 * h + 2] = t[2]; v[4 * h + 3] = t[3];
  }
}

template <int NT, int WPS> struct Cfg32 {
  static constexpr int N = 16 * NT;
  static constexpr int NH = NT / 4;                // 16-byte J pieces per lane per 4-row group
  static constexpr int DPS = NH + 1;               // DMA instructions per group (J pieces + 16 B of r)
  static constexpr int SLOT = NH * 1024 + 64;
  static constexpr int D = 4;                      // ring depth
  static constexpr int VEC = (3 * N + 4 * 64 + 64 + 32) * 4;  // xs, diagS|rp, rhsS|dxs, cons a/b/s/z, cons var, y, b_eq
  static constexpr int LDS = D * SLOT + VEC;
};

// PAD: n is any multiple of 4 up to N = 16 NT and the system is padded to the grid inside the kernel (identity rows, zero right-hand side);
// without it n == N is a compile-time fact and every mask below folds away (BASELINE configs[3] runs that instantiation).
template <int NT, int WPS, bool PAD>
__global__ __launch_bounds__(256 * WPS, WPS) void kkt_fused_f32_kernel(const KernelArgs a) {
  using C = Cfg32<NT, WPS>;
  constexpr int N = C::N, NH = C::NH, DPS = C::DPS, SLOT = C::SLOT, D = C::D;
  constexpr int NB = NT + 2, NR = NT + 1;          // tile columns: x blocks 0..NT-1, y block NT, right-hand side NR
  constexpr int WAVES = 4 * WPS;

  __shared__ __attribute__((aligned(16))) char smem_all[WAVES * C::LDS];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* const smem = smem_all + wave * C::LDS;
  float* const xs = reinterpret_cast<float*>(smem + D * SLOT);
  float* const diagS = xs + N;
  float* const rhsS = diagS + N;
  float* const rp = diagS;    // right-hand side, permuted order (diagS is dead by then)
  float* const dxs = rhsS;    // dx, natural order (rhsS is dead by then)
  float* const cA = rhsS + N;
  float* const cB = cA + 64;
  float* const cS = cB + 64;
  float* const cZ = cS + 64;
  int* const cV = reinterpret_cast<int*>(cZ + 64);
  float* const yb = cZ + 128;
  float* const bb = yb + 16;
  const unsigned ring_base = (unsigned)(uintptr_t)smem;
  const unsigned vec_base = ring_base + D * SLOT;

  // MO_STEP_NO_INEQUALITIES (SolveForUpdateNoInequalities, qp.cc:366-386): the constraints take no part (m = 0 below); the state and direction
  // vectors keep their [x | s(m_lay) | y | z(m_lay)] layout, ds = dz = 0 and both step lengths are 1.
  const int m_lay = a.m;
  const bool no_ineq = (a.flags & MO_STEP_NO_INEQUALITIES) != 0;
  const int k = a.k, m = no_ineq ? 0 : a.m, m_r = a.m_r;
  const int nn = PAD ? a.n : N;   // variables (PAD: a multiple of 4, N - 63 .. N)
  const int nsteps = m_r >> 2;

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
    if (lane_id32() == 0) t = atomicAdd(a.ticket, (unsigned long long)chunk);
    return t;
  };
  auto uniform64 = [](unsigned long long v) -> long long {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  if (PAD && nn < N) {  // the lanes beyond a row of J never write their ring bytes: zeros there, once
    for (int i = lane_id32() * 16; i < D * SLOT; i += 64 * 16) *(f4*)(smem + i) = f4{0.0f, 0.0f, 0.0f, 0.0f};
    lds_fence32();
  }
  if (a.stagger > 0 && !st_rounds) {  // start stagger between the waves that share a SIMD (see kkt_fused.hip): equal-cost problems keep waves in lockstep
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
#ifdef MO_F32_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif

  while (p < a.batch) {
    const bool last_of_chunk = p + 1 >= chunk_end;
    int next_chunk = 0;
    unsigned long long next_ticket = 0;
    if (last_of_chunk) {
      next_chunk = chunk_for(p);
      next_ticket = take_ticket(next_chunk, p);
    }
    const int lane = lane_id32();
    const int g = lane >> 4, j = lane & 15;

    // ---- J stream set-up: lane (g, j) of 4-row group s fetches J(4s + g, 64h + 4j .. +3), h < NH; lane 0 fetches r[4s .. 4s+3]
    const char* jsrc = reinterpret_cast<const char*>((const float*)a.J + p * a.J_stride + (size_t)g * nn + 4 * j);
    const char* rsrc = reinterpret_cast<const char*>((const float*)a.r + p * a.r_stride);
    const char* const lane_piece = smem + lane * 16;
    const char* const r_elem = smem + NH * 1024 + 4 * g;
    auto issue = [&](int slot) {
      const unsigned dst = ring_base + slot * SLOT;
#pragma unroll
      for (int h = 0; h < NH; ++h)
        if (!PAD || 64 * h + 4 * j < nn) dma16_f32(jsrc + 256 * h, dst + h * 1024);   // (lane 0 of every piece is inside the row: the grid is the smallest that holds nn)
      if (lane < 4) dma4_f32(rsrc + 4 * lane, dst + NH * 1024);   // r[4s .. 4s+3] as four dwords: no alignment asked of r
      jsrc += 4 * nn * 4;
      rsrc += 16;
    };
#pragma unroll
    for (int u = 0; u < D; ++u)
      if (u < nsteps) issue(u);

    // ---- P0: small vectors global -> LDS by DMA (no VGPRs held while J streams)
    KArgs32 ka = fresh_args32();
    const float* vp = (const float*)ka->vars + p * ka->vars_stride;
    f4 U[NB * NB];
#pragma unroll
    for (int q = 0; q < NB * NB; ++q) U[q] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    dma_words(vp, vec_base, nn, lane);
    for (int i = nn + lane; i < N; i += 64) xs[i] = 0.0f;
    if (m > 0) {
      const long long coff = p * ka->cons_stride;
      dma_words(ka->cons_var + coff, vec_base + (3 * N + 256) * 4, m, lane);
      dma_words((const float*)ka->cons_a + coff, vec_base + (3 * N) * 4, m, lane);
      dma_words((const float*)ka->cons_b + coff, vec_base + (3 * N + 64) * 4, m, lane);
      dma_words(vp + nn, vec_base + (3 * N + 128) * 4, m, lane);
      dma_words(vp + nn + m + k, vec_base + (3 * N + 192) * 4, m, lane);
    }
    if (k > 0) {
      dma_words(vp + nn + m_lay, vec_base + (3 * N + 320) * 4, k, lane);
      dma_words((const float*)ka->b + p * ka->b_stride, vec_base + (3 * N + 336) * 4, k, lane);
    }

    MO_STAMP32(0);
    // ---- P1: G = J^T J on the matrix cores (upper block triangle), c = J^T r on the VALU
    float cpart[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) cpart[c] = 0.0f;
    for (int q0 = 0; q0 < nsteps; q0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int q = q0 + u;
        if (q < nsteps) {
          const int younger = nsteps - 1 - q;  // groups that may stay in flight (DPS DMAs each)
          if (younger >= D - 1) wait_vmcnt32<(D - 1) * DPS>();
          else if (younger == 2) wait_vmcnt32<2 * DPS>();
          else if (younger == 1) wait_vmcnt32<1 * DPS>();
          else wait_vmcnt32<0>();
          float ops[NT];
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            const f4 v = *(const f4*)(lane_piece + u * SLOT + h * 1024);
            ops[4 * h] = v[0]; ops[4 * h + 1] = v[1]; ops[4 * h + 2] = v[2]; ops[4 * h + 3] = v[3];
          }
          const float rq = *(const float*)(r_elem + u * SLOT);
          lds_fence32();  // the slot's bytes are in registers before the slot is handed back to the DMA engine
          if (q + D < nsteps) issue(u);
#pragma unroll
          for (int ta = 0; ta < NT; ++ta) {
            cpart[ta] = fmaf(ops[ta], rq, cpart[ta]);
#pragma unroll
            for (int tb = ta; tb < NT; ++tb)
              U[ta * NB + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ops[ta], ops[tb], U[ta * NB + tb], 0, 0, 0);
          }
        }
      }
    }
    if (m_r & 3) {  // wave-uniform: up to three rows behind the last whole group -- plain loads, the lanes of the missing rows feed zeros
      float ops[NT];
#pragma unroll
      for (int c = 0; c < NT; ++c) ops[c] = 0.0f;
      float rq = 0.0f;
      if (g < (m_r & 3)) {
        const float* row = (const float*)a.J + p * a.J_stride + (size_t)(4 * nsteps + g) * nn + 4 * j;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          if (!PAD || 64 * h + 4 * j < nn) {
            const f4 v = *(const f4*)(row + 64 * h);
            ops[4 * h] = v[0]; ops[4 * h + 1] = v[1]; ops[4 * h + 2] = v[2]; ops[4 * h + 3] = v[3];
          }
        }
        rq = ((const float*)a.r + p * a.r_stride)[4 * nsteps + g];
      }
#pragma unroll
      for (int ta = 0; ta < NT; ++ta) {
        cpart[ta] = fmaf(ops[ta], rq, cpart[ta]);
#pragma unroll
        for (int tb = ta; tb < NT; ++tb)
          U[ta * NB + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ops[ta], ops[tb], U[ta * NB + tb], 0, 0, 0);
      }
    }
    float cvec[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) cvec[c] = cross_row_sum_f32(cpart[c]);

    MO_STAMP32(1);
    // ---- P3: barrier terms scattered per variable through LDS (duplicates on one variable accumulate)
    wait_vmcnt32<0>();
    ka = fresh_args32();
    const float mu = ka->mu ? ((const float*)ka->mu)[p * ka->mu_stride] : 0.0f;
    for (int i = lane; i < N; i += 64) { diagS[i] = 0.0f; rhsS[i] = 0.0f; }
    int cvar = 0; float ca = 1.0f, cb = 0.0f, cs = 1.0f, cz = 0.0f;
    if (lane < m) { cvar = cV[lane]; ca = cA[lane]; cb = cB[lane]; cs = cS[lane]; cz = cZ[lane]; }
    lds_fence32();
    bool bad_index = (lane < m) && ((cvar < 0) || (cvar >= nn));
    if (bad_index) cvar = 0;
    const bool slack_bad = __any((lane < m) && !(cs > 0.0f));
    bool any_bad_index = __any(bad_index);
    if (no_ineq && m_lay > 0) {  // the index check of Setup (qp.cc:70-72) does not depend on the flag
      bool bad = false;
      const int* cvp = ka->cons_var + p * ka->cons_stride;
      for (int ix = lane; ix < m_lay; ix += 64) { const int v = cvp[ix]; bad = bad || v < 0 || v >= nn; }
      any_bad_index = __any(bad);
    }
    const float cs_inv = rcp_f32(cs);
    if (lane < m) {
      const float zs = cz * cs_inv;
      atomicAdd(&diagS[cvar], ca * zs * ca);                         // qp.cc:296
      atomicAdd(&rhsS[cvar], ca * (cz * (cs - cb) + mu) * cs_inv);    // x+ form of qp.cc:340-341
    }
    lds_fence32();
    float dS[NT], rS[NT];
    ldv32<NT>(diagS, j, dS);
    ldv32<NT>(rhsS, j, rS);
    if (g == 0) {
#pragma unroll
      for (int c = 0; c < NT; ++c) rp[16 * c + j] = rS[c] - cvec[c];
    }
    lds_fence32();

    // ---- P2/P4: lambda + Sigma on the diagonal tiles, [A_eq^T] tile column, right-hand side tile column
    const float lam_in = ka->lambda_vec ? ((const float*)ka->lambda_vec)[p * ka->lambda_vec_stride] : (float)ka->lambda;
    const float lam = lam_in > 0.0f ? lam_in : 0.0f;                 // nonlinear.cc:187-189
    {
      const float* Ap = k > 0 ? (const float*)ka->A + p * ka->A_stride : nullptr;
      const int A_ld = ka->A_ld;
#pragma unroll
      for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int natcol = 64 * (c >> 2) + 16 * g + 4 * t + (c & 3);  // variable at position 16c + 4g + t
          U[c * NB + c][t] += (j == 4 * g + t) ? ((!PAD || natcol < nn) ? lam + dS[c] : 1.0f) : 0.0f;   // padding: identity rows, zero right-hand side
          U[c * NB + NT][t] = (j < k && (!PAD || natcol < nn)) ? Ap[j + (size_t)natcol * A_ld] : 0.0f;
#if !MO_F32_RHS_VECTOR
          const float rv = rp[16 * c + 4 * g + t];
          U[c * NB + NR][t] = (j == 0) ? rv : 0.0f;
#endif
        }
      }
#if !MO_F32_RHS_VECTOR
#pragma unroll
      for (int t = 0; t < 4; ++t) U[NT * NB + NR][t] = (j == 0 && 4 * g + t < k) ? -bb[4 * g + t] : 0.0f;  // -b_eq
#endif
    }
#if MO_F32_RHS_VECTOR
    // The right-hand side as a VECTOR: rj[c] = its entry at position 16 c + j (every lane group holds a copy).  Tile column NT + 1 costs
    // 4 MFMAs per tile product for one useful column in sixteen (180 MFMAs at NT = 8); the vector pays 4 FMAs + a cross-row sum per product
    // and two LDS hops per block step (lane j <-> rows 4 g .. 4 g + 3, the layout a tile product wants its operand in).  rp (= diagS) holds
    // the blocks 0 .. NT - 1 the back-substitution reads; block NT and the hop buffer zS follow it in rhsS (dead until dxs is written).
    float rj[NT + 1];
#pragma unroll
    for (int c = 0; c < NT; ++c) rj[c] = rS[c] - cvec[c];
    rj[NT] = (j < k) ? -bb[j] : 0.0f;                                 // -b_eq
    float* const zS = rhsS + 16;
#endif

    // ---- P5: block elimination with 16x16 pivot blocks (pivot blocks 0..NT; column NR only rides along)
    __builtin_amdgcn_sched_barrier(0);
    MO_STAMP32(2);
    bool ok = true;
#if MO_F32_LOOKAHEAD
    ok = block_eliminate_lookahead_f32<NT>(U, k, j);
#else
#pragma unroll
    for (int pa = 0; pa <= NT; ++pa) {
#if MO_F32_RHS_VECTOR
      if (g == 0) rp[16 * pa + j] = rj[pa];                           // block pa of the right-hand side is final: to LDS (the sweep covers the hop)
#endif
      ok = sweep_tile_f32(U[pa * NB + pa], pa < NT ? 16 : k, j) && ok;
      __builtin_amdgcn_sched_barrier(0);
      MO_STAMP32(3);
#if MO_F32_RHS_VECTOR
      {
        lds_fence32();
        const f4 ra4 = *(const f4*)(rp + 16 * pa + 4 * g);            // rows 4 g .. 4 g + 3 of block pa
        float zs = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) zs = fmaf(U[pa * NB + pa][t], ra4[t], zs);
        zs = cross_row_sum_f32(zs);                                   // (-T^-1) r_a at position j
        if (g == 0) zS[j] = zs;
      }
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int pc = pa + 1; pc < (MO_F32_RHS_VECTOR ? NB - 1 : NB); ++pc) {
        const f4 negZ = mfma4_f32(U[pa * NB + pa], U[pa * NB + pc], f4{0.0f, 0.0f, 0.0f, 0.0f});  // (-T^-1) U_ac
#pragma unroll
        for (int pb = pa + 1; pb <= (pc < NT ? pc : NT); ++pb) U[pb * NB + pc] = mfma4_f32(U[pa * NB + pb], negZ, U[pb * NB + pc]);
        __builtin_amdgcn_sched_barrier(0);
      }
#if MO_F32_RHS_VECTOR
      if (pa < NT) {
        lds_fence32();
        const f4 z4 = *(const f4*)(zS + 4 * g);
#pragma unroll
        for (int pb = pa + 1; pb <= NT; ++pb) {
          float us = 0.0f;
#pragma unroll
          for (int t = 0; t < 4; ++t) us = fmaf(U[pa * NB + pb][t], z4[t], us);
          rj[pb] += cross_row_sum_f32(us);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      MO_STAMP32(4);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    MO_STAMP32(4);

    // ---- P6: backward substitution; xb[c] = solution at permuted position 16c + j (replicated over g); xb[NT] = -y+
    float xb[NT + 1];
#pragma unroll
    for (int pa = NT; pa >= 0; --pa) {
      float vt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float pt = 0.0f;
#pragma unroll
        for (int pb = pa + 1; pb <= NT; ++pb) pt = fmaf(U[pa * NB + pb][t], xb[pb], pt);
        pt = row_sum_f32(pt);
#if MO_F32_RHS_VECTOR
        vt[t] = rp[16 * pa + 4 * g + t] - pt;
#else
        vt[t] = row_bcast_f32<0>(U[pa * NB + NR][t]) - pt;
#endif
      }
      float q = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) q = fmaf(U[pa * NB + pa][t], vt[t], q);
      xb[pa] = -cross_row_sum_f32(q);
      __builtin_amdgcn_sched_barrier(0);
    }

    MO_STAMP32(5);
    // ---- P7: direction, step lengths, status
    ka = fresh_args32();
    float dxv[NT];
    {
      float xn[NT];
      ldv32<NT>(xs, j, xn);
#pragma unroll
      for (int c = 0; c < NT; ++c) dxv[c] = xb[c] - xn[c];
    }
    bool finite = true;
#pragma unroll
    for (int c = 0; c < NT; ++c) finite = finite && (fabsf(dxv[c]) < INFINITY);
    if (g == 0) stv32<NT>(dxs, j, dxv);
    lds_fence32();
    float dsv = 0.0f, dzv = 0.0f, ap = 1.0f, ad = 1.0f;
    if (lane < m) {
      const float ca2 = cA[lane], cb2 = cB[lane], cs2 = cS[lane], cz2 = cZ[lane];
      const int cvar2 = bad_index ? 0 : cV[lane];
      const float r_pi = ca2 * xs[cvar2] + cb2 - cs2;                               // qp.cc:416
      dsv = ca2 * dxs[cvar2] + r_pi;                                                // qp.cc:361
      dzv = -(cz2 * cs_inv) * dsv - cs_inv * (cs2 * cz2 - mu);                      // qp.cc:362
      const float tau = (float)ka->tau;
      if (cs2 + dsv <= 0.0f && fabsf(dsv) > 0.0f) ap = -tau * cs2 * rcp_f32(dsv);   // qp.cc:498-503
      if (cz2 + dzv <= 0.0f && fabsf(dzv) > 0.0f) ad = -tau * cz2 * rcp_f32(dzv);
      finite = finite && (fabsf(dsv) < INFINITY) && (fabsf(dzv) < INFINITY);
    }
    ap = cross_row_min_f32(row_min_f32(ap));
    ad = cross_row_min_f32(row_min_f32(ad));
    const float dyv = (j < k) ? (-xb[NT] - yb[j]) : 0.0f;
    finite = finite && (fabsf(dyv) < INFINITY);
    int st = MO_STATUS_OK;
    if (!__all(finite)) st = MO_STATUS_NONFINITE;
    if (!ok) st = MO_STATUS_FACTORIZATION_FAILED;
    if (slack_bad) st = MO_STATUS_NONPOSITIVE_SLACK;
    if (any_bad_index) st = MO_STATUS_BAD_INDEX;
    const float nanv = __builtin_nanf("");
    float* dp = (float*)ka->delta + p * ka->delta_stride;
    if (g == 0) {
      float outv[NT];
#pragma unroll
      for (int c = 0; c < NT; ++c) outv[c] = st == MO_STATUS_OK ? dxv[c] : nanv;
      if (((uintptr_t)dp & 15) == 0) {   // (wave-uniform) whole 16-byte pieces where the caller's stride allows them
        if (PAD) stv32_n<NT>(dp, j, nn, outv); else stv32<NT>(dp, j, outv);
      } else {
#pragma unroll
        for (int c = 0; c < NT; ++c)
          if (!PAD || 64 * (c >> 2) + 4 * j < nn) dp[64 * (c >> 2) + 4 * j + (c & 3)] = outv[c];
      }
      if (j < k) dp[nn + m_lay + j] = st == MO_STATUS_OK ? dyv : nanv;
    }
    if (no_ineq) {  // ds = dz = 0 (qp.cc:366-386 writes only dx, dy)
      for (int ix = lane; ix < m_lay; ix += 64) { dp[nn + ix] = st == MO_STATUS_OK ? 0.0f : nanv; dp[nn + m_lay + k + ix] = st == MO_STATUS_OK ? 0.0f : nanv; }
    }
    if (lane < m) {
      dp[nn + lane] = st == MO_STATUS_OK ? dsv : nanv;
      dp[nn + m + k + lane] = st == MO_STATUS_OK ? dzv : nanv;
    }
    if (lane == 0) {
      if (ka->alpha) {
        ((float*)ka->alpha)[2 * p] = st == MO_STATUS_OK ? ap : nanv;
        ((float*)ka->alpha)[2 * p + 1] = st == MO_STATUS_OK ? ad : nanv;
      }
      if (ka->status) ka->status[p] = st;
    }
    lds_fence32();  // the LDS vectors are re-initialised by the next problem
    MO_STAMP32(6);
    if (last_of_chunk) {
      p = uniform64(next_ticket) + ticket_base;
      chunk_end = p + next_chunk;
    } else {
      ++p;
    }
  }
#ifdef MO_F32_STAMPS
  if ((threadIdx.x & 63) == 0 && a.debug) {
    for (int i = 0; i < 8; ++i) atomicAdd(a.debug + i, stamp_acc[i]);
    atomicAdd(a.debug + 8, 1ull);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// QPInteriorPointSolver::Solve (qp.cc:100-151), one Iterate (qp.cc:153-201) and EvaluateKKTConditions + ComputeErrors
// (qp.cc:391-437) in fp32, one wavefront per QP -- the fp32 sibling of kkt_fused_solve_kernel (kkt_fused.hip): every pass
// rebuilds the G tiles (first pass: J through the LDS-DMA ring; later passes: the parked tiles), evaluates the KKT residual
// as tile products K [x; -y] in registers (qp.cc:404-419), and solves for the DIRECTION with the residual as right-hand
// side (qp.cc:255-268, 337-363).  All three BarrierStrategy values; PREDICTOR_CORRECTOR pushes its second right-hand side
// through the factors of the predictor solve.  Tile layout and variable permutation as in the step kernel above (row 4g + t,
// position 16c + i <-> variable 64(c>>2) + 4i + (c&3)); the right-hand side rides in column 0 of tile column NR = NT + 1.
template <int NT> struct SolveCfg32 {
  static constexpr int N = 16 * NT;
  static constexpr int NH = NT / 4;
  static constexpr int DPS = NH + 1;
  static constexpr int SLOT = NH * 1024 + 64;
  static constexpr int D = 4;
  static constexpr int VEC = (6 * N + 32) * 4;  // xs, xp, azS, diagS, rhoS, tmp, ysmall[32]
  // Tile park (as in the fp64 Solve kernel): after the first pass has streamed J the ring is idle, and in fp32 ALL G tiles (1 KiB each,
  // lane-linear) + c fit the wave's share of the LDS (n = 128: 36.5 KiB of 40 at one wave per SIMD; n = 64: 10.3 KiB of 13.6 at three)
  static constexpr int NTILES = NT * (NT + 1) / 2;
  static constexpr int AREA = D * SLOT > NTILES * 1024 + N * 4 ? D * SLOT : NTILES * 1024 + N * 4;
  static constexpr int LDS = AREA + VEC;
};

__device__ inline float wave_sum_f32(float v) { return cross_row_sum_f32(row_sum_f32(v)); }
__device__ inline int natvar32(int c, int i) { return 64 * (c >> 2) + 4 * i + (c & 3); }  // variable at tile position 16c + i

// A second right-hand side through the factors the elimination left in the tiles (diagonal tiles: -T^-1, row panels: their
// forward-eliminated values).  Vectors are V16 (value at lane j, replicated over g); rb[] is consumed; the forward-eliminated
// blocks are parked in rbuf_x / rbuf_y for the substitution.  xb[c] = solution at permuted position 16c + j, xb[NT] = the y block.
template <int NT>
__device__ inline void solve_second_rhs_f32(const f4 (&U)[(NT + 2) * (NT + 2)], int g, int j, float (&rb)[NT + 1], float* hop, float* rbuf_x,
                                            float* rbuf_y, float (&xb)[NT + 1]) {
  constexpr int NB = NT + 2;
#pragma unroll
  for (int pa = 0; pa <= NT; ++pa) {  // forward: r_b += U_ab^T (-T_a^-1 r_a), b > a
    if (g == 0) { hop[j] = rb[pa]; (pa < NT ? rbuf_x + 16 * pa : rbuf_y)[j] = rb[pa]; }
    lds_fence32();
    if (pa < NT) {
      float q = 0.0f;
      {
        const f4 h4 = *(const f4*)(hop + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) q = fmaf(U[pa * NB + pa][t], h4[t], q);
      }
      const float w = cross_row_sum_f32(q);  // (-T_a^-1 r_a)(j)
      lds_fence32();                         // hop has been read
      if (g == 0) hop[j] = w;
      lds_fence32();
      const f4 wr = *(const f4*)(hop + 4 * g);
#pragma unroll
      for (int pb = pa + 1; pb <= NT; ++pb) {
        float q2 = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) q2 = fmaf(U[pa * NB + pb][t], wr[t], q2);
        rb[pb] += cross_row_sum_f32(q2);
      }
      lds_fence32();                         // hop has been read before the next block overwrites it
    }
  }
#pragma unroll
  for (int pa = NT; pa >= 0; --pa) {  // backward
    const float* rsrc = pa < NT ? rbuf_x + 16 * pa : rbuf_y;
    const f4 r4 = *(const f4*)(rsrc + 4 * g);
    float vt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float pt = 0.0f;
#pragma unroll
      for (int pb = pa + 1; pb <= NT; ++pb) pt = fmaf(U[pa * NB + pb][t], xb[pb], pt);
      if (pa < NT) pt = row_sum_f32(pt);
      vt[t] = r4[t] - pt;
    }
    float q = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) q = fmaf(U[pa * NB + pa][t], vt[t], q);
    xb[pa] = -cross_row_sum_f32(q);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NT, int WPS, bool PAD>
__global__ __launch_bounds__(256 * WPS, WPS) void kkt_fused_f32_solve_kernel(const KernelArgs a) {
  using C = SolveCfg32<NT>;
  constexpr int N = C::N, NH = C::NH, DPS = C::DPS, SLOT = C::SLOT, D = C::D;
  constexpr int NB = NT + 2, NR = NT + 1;          // tile columns: x blocks 0..NT-1, y block NT, right-hand side NR
  constexpr int WAVES = 4 * WPS;

  __shared__ __attribute__((aligned(16))) char smem_all[WAVES * C::LDS];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* const smem = smem_all + wave * C::LDS;
  static_assert(WAVES * C::LDS <= 160 * 1024, "LDS budget");
  float* const xs = reinterpret_cast<float*>(smem + C::AREA);   // x, natural order
  float* const xp = xs + N;                                      // x, position order
  float* const azS = xp + N;                                     // sum a z per variable, natural order
  float* const diagS = azS + N;                                  // barrier diagonal per variable
  float* const rhoS = diagS + N;                                 // inequality part of r_aug per variable
  float* const tmp = rhoS + N;                                   // layout-conversion scratch
  float* const ysm = tmp + N;                                    // [0,16): y ; [16,32): -r_pe
  const unsigned ring_base = (unsigned)(uintptr_t)smem;

  const int m = a.m, m_r = a.m_r;
  const int nsteps = m_r >> 2;
  const float inv_m = m > 0 ? 1.0f / (float)m : 0.0f;
  const bool qpl = a.J == nullptr;  // wave-uniform: QP-level input (G, c given)

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
    if (lane_id32() == 0) t = atomicAdd(a.ticket, (unsigned long long)chunk);
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
    // the argument block and the shape are re-read from the kernarg segment where they are used (see kkt_fused_solve_kernel): held in SGPRs
    // for the whole kernel they were spilled into VGPR lanes (226 v_writelane / 534 v_readlane in round 2's build)
    KArgs32 ka = fresh_args32();
    const int k = ka->k, m = ka->m, nn = PAD ? ka->n : N;   // nn variables (PAD: a multiple of 4) on the N = 16 NT grid
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
    const int lane = lane_id32();
    const int g = lane >> 4, j = lane & 15;

    float* vp = (float*)ka->vars + p * ka->vars_stride;
    const float lam_in = ka->lambda_vec ? ((const float*)ka->lambda_vec)[p * ka->lambda_vec_stride] : (float)ka->lambda;
    const float lam = (!qpl && lam_in > 0.0f) ? lam_in : 0.0f;  // a given G already carries the LM damping

    // ---- constants of the problem
    int cvar = 0; float ca = 1.0f, cb = 0.0f;
    if (lane < m) {
      cvar = ka->cons_var[p * ka->cons_stride + lane];
      ca = ((const float*)ka->cons_a)[p * ka->cons_stride + lane];
      cb = ((const float*)ka->cons_b)[p * ka->cons_stride + lane];
    }
    float b_col = 0.0f;
    if (j < k) b_col = ((const float*)ka->b + p * ka->b_stride)[j];
    const float* const Ap = k > 0 ? (const float*)ka->A + p * ka->A_stride : nullptr;

    // ---- state: x in the V16 layout (position 16c + j, replicated over g), y in lanes j < k, s / z per constraint lane
    float xv[NT], yv = 0.0f, cs = 1.0f, cz = 1.0f;
#pragma unroll
    for (int c = 0; c < NT; ++c) xv[c] = 0.0f;
    const bool residual_mode = ka->mode == MODE_RESIDUAL;
    const bool iterate_mode = ka->mode == MODE_ITERATE || residual_mode;
    if (iterate_mode || ka->sp.initial_guess_method == MO_GUESS_USER_PROVIDED) {  // qp.cc:440-442
#pragma unroll
      for (int c = 0; c < NT; ++c) xv[c] = (!PAD || natvar32(c, j) < nn) ? vp[natvar32(c, j)] : 0.0f;
      if (j < k) yv = vp[nn + m + j];
      if (lane < m) { cs = vp[nn + lane]; cz = vp[nn + m + k + lane]; }
    }
    const bool bad_index = __any((lane < m) && ((cvar < 0) || (cvar >= nn)));
    if (bad_index) cvar = 0;

    int st = bad_index ? MO_STATUS_BAD_INDEX : MO_STATUS_OK;
    int term = MO_MAX_ITERATIONS, it = 0;
    float mu = iterate_mode ? (ka->mu ? ((const float*)ka->mu)[p * ka->mu_stride] : 0.0f) : (float)ka->sp.initial_mu;
    bool guess_pass = !iterate_mode && ka->sp.initial_guess_method == MO_GUESS_SOLVE_EQUALITY_CONSTRAINED;
    float* iter_out = ka->iterations ? (float*)ka->iterations + (size_t)p * ka->sp.max_iterations * MO_ITER_RECORD : nullptr;

    // s = max(1e-9, a x + b), z = 1/s after clamping x into the feasible region in constraint order (qp.cc:464-481)
    auto clamp_and_init_slacks = [&]() {
      if (g == 0) stv32<NT>(xs, j, xv);
      lds_fence32();
      for (int c = 0; c < m; ++c) {  // wave-uniform loop; one constraint at a time keeps the reference's order
        if (lane == c) {
          const float x0 = xs[cvar];
          float x1;
          if (ca < 0.0f) { const float lim = cb / -ca; x1 = x0 < lim ? x0 : lim; }  // ClampX, qp.hpp:43-53
          else { const float lim = -cb / ca; x1 = x0 > lim ? x0 : lim; }
          xs[cvar] = x1;
        }
        lds_fence32();
      }
      ldv32<NT>(xs, j, xv);
      float sz = 0.0f;
      if (lane < m) {
        const float sv = ca * xs[cvar] + cb;
        cs = sv > 1.0e-9f ? sv : 1.0e-9f;
        cz = 1.0f / cs;
        sz = cs * cz;
      }
      if (ka->sp.initialize_mu_with_complementarity) mu = wave_sum_f32(sz) * inv_m;  // qp.cc:115
    };
    if (st == MO_STATUS_OK && !iterate_mode && ka->sp.initial_guess_method == MO_GUESS_NAIVE) clamp_and_init_slacks();
    if (!iterate_mode && ka->sp.initial_guess_method == MO_GUESS_USER_PROVIDED && ka->sp.initialize_mu_with_complementarity)
      mu = wave_sum_f32(lane < m ? cs * cz : 0.0f) * inv_m;  // qp.cc:115 on the caller's state (0 without inequalities, qp.cc:509-516)

    float n_rd2 = 0, n_rpe2 = 0, n_rc2 = 0, n_rc1 = 0, n_rpi2 = 0;
    // ComputeErrors (qp.cc:423-437) as SQUARED norms (the decisions compare squares; square roots only for the records)
    auto kkt_errors_sq = [&](float mu_e, float (&o)[4]) {
      o[0] = n_rd2;
      o[2] = k > 0 ? n_rpe2 : 0.0f;
      if (m > 0) {
        const float corrected = n_rc2 - 2 * (n_rc1 * mu_e) + (mu_e * mu_e) * (float)m;
        o[1] = corrected > 0.0f ? corrected : 0.0f;
        o[3] = n_rpi2;
      } else { o[1] = 0.0f; o[3] = 0.0f; }
    };
    // G = J^T J + lambda I and c = J^T r do not change between the passes: the first pass parks its tiles in a per-problem scratch
    // (plan-owned, ka->G_out; lane-linear), later passes reload them instead of re-streaming J
    constexpr int NTILES = C::NTILES;
    float* const park = reinterpret_cast<float*>(smem);            // NTILES tiles (256 floats each, lane-linear), then c (V16, N floats)
    float* const cpark = park + NTILES * 256;
    bool tiles_cached = false;
    float mu_used = mu;
    float ip_alpha_p = 1.0f, ip_alpha_d = 1.0f;
    const bool use_pc = (iterate_mode ? ka->barrier_strategy : ka->sp.barrier_strategy) == MO_PREDICTOR_CORRECTOR && m > 0;
    const float nanf32 = __builtin_nanf("");
    float ip_mu = mu, probe_p = nanf32, probe_d = nanf32, mu_aff = nanf32, mu_pc = 0.0f;

    while (st == MO_STATUS_OK) {
      const bool include_ineq = !guess_pass && !(residual_mode && (ka->flags & MO_STEP_NO_INEQUALITIES));
      const int lane = lane_id32(), g = lane >> 4, j = lane & 15;  // re-made opaque every pass (nothing lane-derived is kept across the factorisation)
      ka = fresh_args32();
      const int k = ka->k, m = ka->m, nn = PAD ? ka->n : N;
      // ---------------------------------------------------------------- part A: tiles, residual, norms
      const bool stream_now = !qpl && __builtin_amdgcn_readfirstlane((int)!tiles_cached) != 0;
      const char* jsrc = reinterpret_cast<const char*>((const float*)ka->J + p * ka->J_stride + (size_t)g * nn + 4 * j);
      const char* rsrc = reinterpret_cast<const char*>((const float*)ka->r + p * ka->r_stride);
      auto issue = [&](int slot) {
        const unsigned dst = ring_base + slot * SLOT;
#pragma unroll
        for (int h = 0; h < NH; ++h)
          if (!PAD || 64 * h + 4 * j < nn) dma16_f32(jsrc + 256 * h, dst + h * 1024);   // (lane 0 of every piece is inside the row)
        if (lane < 4) dma4_f32(rsrc + 4 * lane, dst + NH * 1024);   // r[4s .. 4s+3] as four dwords: no alignment asked of r
        jsrc += 4 * nn * 4;
        rsrc += 16;
      };
      if (stream_now) {
        if (PAD && nn < N) {  // the pieces beyond the row are never written by the DMAs, and the ring held the previous problem's parked tiles: zeros there
#pragma unroll
          for (int u = 0; u < D; ++u)
#pragma unroll
            for (int h = 0; h < NH; ++h)
              if (PAD && 64 * h + 4 * j >= nn) *(f4*)(smem + lane * 16 + u * SLOT + h * 1024) = f4{0.0f, 0.0f, 0.0f, 0.0f};
          lds_fence32();
        }
#pragma unroll
        for (int u = 0; u < D; ++u)
          if (u < nsteps) issue(u);
      }
      f4 U[NB * NB];
#pragma unroll
      for (int q = 0; q < NB * NB; ++q) U[q] = f4{0.0f, 0.0f, 0.0f, 0.0f};
      // publish the state for the layout conversions below; zero the per-variable scatter arrays
      if (g == 0) {
        stv32<NT>(xs, j, xv);
#pragma unroll
        for (int c = 0; c < NT; ++c) xp[16 * c + j] = xv[c];
      }
      for (int i = lane; i < N; i += 64) { azS[i] = 0.0f; diagS[i] = 0.0f; rhoS[i] = 0.0f; }
      float cvec[NT];
      if (qpl) {  // only the lower triangle of G is read (qp.cc:289, 404)
        const float* Gp = (const float*)ka->G + p * ka->G_stride;
        const float* cp = (const float*)ka->c + p * ka->c_stride;
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) {
#pragma unroll
          for (int tb = ta; tb < NT; ++tb) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int vr = natvar32(ta, 4 * g + t), vc = natvar32(tb, j);
              const int hi = vr > vc ? vr : vc, lo = vr > vc ? vc : vr;
              U[ta * NB + tb][t] = (!PAD || hi < nn) ? Gp[hi + (size_t)lo * ka->G_ld] : 0.0f;
            }
          }
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) cvec[c] = (!PAD || natvar32(c, j) < nn) ? cp[natvar32(c, j)] : 0.0f;
      } else if (stream_now) {
        const char* const lane_piece = smem + lane * 16;
        const char* const r_elem = smem + NH * 1024 + 4 * g;
        float cpart[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) cpart[c] = 0.0f;
        for (int q0 = 0; q0 < nsteps; q0 += D) {
#pragma unroll
          for (int u = 0; u < D; ++u) {
            const int q = q0 + u;
            if (q < nsteps) {
              const int younger = nsteps - 1 - q;  // groups that may stay in flight (DPS DMAs each)
              if (younger >= D - 1) wait_vmcnt32<(D - 1) * DPS>();
              else if (younger == 2) wait_vmcnt32<2 * DPS>();
              else if (younger == 1) wait_vmcnt32<1 * DPS>();
              else wait_vmcnt32<0>();
              float ops[NT];
#pragma unroll
              for (int h = 0; h < NH; ++h) {
                const f4 v = *(const f4*)(lane_piece + u * SLOT + h * 1024);
                ops[4 * h] = v[0]; ops[4 * h + 1] = v[1]; ops[4 * h + 2] = v[2]; ops[4 * h + 3] = v[3];
              }
              const float rq = *(const float*)(r_elem + u * SLOT);
              lds_fence32();  // the slot's bytes are in registers before the slot is handed back to the DMA engine
              if (q + D < nsteps) issue(u);
#pragma unroll
              for (int ta = 0; ta < NT; ++ta) {
                cpart[ta] = fmaf(ops[ta], rq, cpart[ta]);
#pragma unroll
                for (int tb = ta; tb < NT; ++tb)
                  U[ta * NB + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ops[ta], ops[tb], U[ta * NB + tb], 0, 0, 0);
              }
            }
          }
        }
        wait_vmcnt32<0>();
        if (m_r & 3) {  // wave-uniform: up to three rows behind the last whole group -- plain loads, the lanes of the missing rows feed zeros
          float ops[NT];
#pragma unroll
          for (int c = 0; c < NT; ++c) ops[c] = 0.0f;
          float rq = 0.0f;
          if (g < (m_r & 3)) {
            const float* row = (const float*)ka->J + p * ka->J_stride + (size_t)(4 * nsteps + g) * nn + 4 * j;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
              if (!PAD || 64 * h + 4 * j < nn) {
                const f4 v = *(const f4*)(row + 64 * h);
                ops[4 * h] = v[0]; ops[4 * h + 1] = v[1]; ops[4 * h + 2] = v[2]; ops[4 * h + 3] = v[3];
              }
            }
            rq = ((const float*)ka->r + p * ka->r_stride)[4 * nsteps + g];
          }
#pragma unroll
          for (int ta = 0; ta < NT; ++ta) {
            cpart[ta] = fmaf(ops[ta], rq, cpart[ta]);
#pragma unroll
            for (int tb = ta; tb < NT; ++tb)
              U[ta * NB + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ops[ta], ops[tb], U[ta * NB + tb], 0, 0, 0);
          }
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) {  // G = J^T J + lambda I (nonlinear.cc:187-189): lambda is part of G in the residual too
#pragma unroll
          for (int t = 0; t < 4; ++t) U[c * NB + c][t] += (j == 4 * g + t) ? lam : 0.0f;
          cvec[c] = cross_row_sum_f32(cpart[c]);
        }
        if (!iterate_mode) {  // park the tiles and c in the (now idle) ring for the following passes
          int ti = 0;
#pragma unroll
          for (int ta = 0; ta < NT; ++ta) {
#pragma unroll
            for (int tb = ta; tb < NT; ++tb, ++ti) *(f4*)(park + (ti * 64 + lane) * 4) = U[ta * NB + tb];
          }
          if (g == 0) {
#pragma unroll
            for (int c = 0; c < NT; ++c) cpark[16 * c + j] = cvec[c];
          }
          tiles_cached = true;
        }
      } else {  // fetch what the first pass parked
        int ti = 0;
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) {
#pragma unroll
          for (int tb = ta; tb < NT; ++tb, ++ti) U[ta * NB + tb] = *(const f4*)(park + (ti * 64 + lane) * 4);
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) cvec[c] = cpark[16 * c + j];
      }
#pragma unroll
      for (int c = 0; c < NT; ++c) {  // [A_eq^T] tile column
#pragma unroll
        for (int t = 0; t < 4; ++t) U[c * NB + NT][t] = (j < k && (!PAD || natvar32(c, 4 * g + t) < nn)) ? Ap[j + (size_t)natvar32(c, 4 * g + t) * ka->A_ld] : 0.0f;
      }
      lds_fence32();
      float r_pi = 0.0f, r_comp = 0.0f;
      if (include_ineq && lane < m) {
        atomicAdd(&azS[cvar], ca * cz);               // qp.cc:415
        r_pi = ca * xs[cvar] + cb - cs;               // qp.cc:416
        r_comp = cs * cz;                             // qp.cc:417
      }
      float r_d[NT], r_pe;
      {
        // w = K [x; -y] as tile products: type 1 (sum over tile rows, result on lanes) over every stored tile,
        // type 2 (sum over tile columns, result on rows) over the strictly upper tiles; the latter goes through LDS once.
        float acc1[NT + 1];
#pragma unroll
        for (int b = 0; b <= NT; ++b) acc1[b] = 0.0f;
#pragma unroll
        for (int ra = 0; ra < NT; ++ra) {
          const f4 vR = *(const f4*)(xp + 16 * ra + 4 * g);  // x at the tile's rows 4g .. 4g + 3
#pragma unroll
          for (int b = ra; b <= NT; ++b) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc1[b] = fmaf(U[ra * NB + b][t], vR[t], acc1[b]);
          }
          f4 pt4;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float pt = 0.0f;
#pragma unroll
            for (int b = ra + 1; b <= NT; ++b) pt = fmaf(U[ra * NB + b][t], b < NT ? xv[b < NT ? b : 0] : -yv, pt);  // yv is zero beyond k
            pt4[t] = row_sum_f32(pt);
          }
          if (j == 0) *(f4*)(tmp + 16 * ra + 4 * g) = pt4;
        }
        lds_fence32();
        {
          float azv[NT];
          ldv32<NT>(azS, j, azv);
#pragma unroll
          for (int c = 0; c < NT; ++c) r_d[c] = cross_row_sum_f32(acc1[c]) + tmp[16 * c + j] + cvec[c] - azv[c];  // qp.cc:404-406, 415
        }
        r_pe = (j < k) ? cross_row_sum_f32(acc1[NT]) + b_col : 0.0f;                                          // qp.cc:408
        float t = 0.0f;
#pragma unroll
        for (int c = 0; c < NT; ++c) t = fmaf(r_d[c], r_d[c], t);
        n_rd2 = readlane_f32(row_sum_f32(t), 0);
        n_rpe2 = readlane_f32(row_sum_f32(r_pe * r_pe), 0);
        n_rc2 = wave_sum_f32(r_comp * r_comp);
        n_rc1 = wave_sum_f32(r_comp);
        n_rpi2 = wave_sum_f32(r_pi * r_pi);
      }
      if (residual_mode) {  // r_ = [r_d | r_comp | r_pe | r_pi] (qp.cc:391-420) and the four norms of ComputeErrors (qp.cc:423-437)
        float* ro = (float*)ka->r_out + p * ka->r_out_stride;
        if (g == 0) {
#pragma unroll
          for (int c = 0; c < NT; ++c)
            if (!PAD || natvar32(c, j) < nn) ro[natvar32(c, j)] = r_d[c];
          if (j < k) ro[nn + m + j] = r_pe;
        }
        if (lane < m) { ro[nn + lane] = r_comp; ro[nn + m + k + lane] = r_pi; }
        if (ka->kkt_out) {
          float kq[4];
          kkt_errors_sq(mu, kq);
          if (!include_ineq) { kq[1] = 0.0f; kq[3] = 0.0f; }
          const float e0 = sqrtf(kq[0]), e1 = sqrtf(kq[1]), e2 = sqrtf(kq[2]), e3 = sqrtf(kq[3]);
          if (lane == 0) { float* ko = (float*)ka->kkt_out + 4 * p; ko[0] = e0; ko[1] = e1; ko[2] = e2; ko[3] = e3; }
        }
        break;
      }
      if (!guess_pass && !iterate_mode) {
        // ---- the decision point of Solve (qp.cc:116-147)
        if (it > 0) {
          float kf[4];
          kkt_errors_sq(mu_used, kf);                               // kkt_after of the previous iteration (squared), qp.cc:127
          const float cur_mu = n_rc1 * inv_m;                       // ComputeMu, qp.cc:509-516
          if (iter_out) {
            const float r4 = sqrtf(kf[0]), r5 = sqrtf(kf[1]), r6 = sqrtf(kf[2]), r7 = sqrtf(kf[3]);
            if (lane == 0) {
              float* rec = iter_out + (size_t)(it - 1) * MO_ITER_RECORD;
              rec[4] = r4; rec[5] = r5; rec[6] = r6; rec[7] = r7;
              rec[8] = ip_mu; rec[9] = ip_alpha_p; rec[10] = ip_alpha_d;
              rec[11] = probe_p; rec[12] = probe_d; rec[13] = mu_aff;
            }
          }
          float kmax2 = kf[0];                                      // KKTError::Max() squared
          kmax2 = kf[1] > kmax2 ? kf[1] : kmax2; kmax2 = kf[2] > kmax2 ? kf[2] : kmax2; kmax2 = kf[3] > kmax2 ? kf[3] : kmax2;
          const float tol = (float)ka->sp.termination_kkt_tol;
          if (kmax2 < tol * tol && cur_mu < (float)ka->sp.termination_complementarity_tol) {  // qp.cc:132-137
            term = MO_SATISFIED_KKT_TOL;
            break;
          }
          if (kmax2 <= mu * mu || !ka->sp.decrease_mu_only_on_small_error) {                   // qp.cc:140-146 (mu > 0)
            if (ka->sp.barrier_strategy == MO_FIXED_DECREASE) mu *= (float)ka->sp.sigma;
            else mu = (float)ka->sp.sigma * cur_mu;
          }
        }
        if (it >= ka->sp.max_iterations) break;                          // MAX_ITERATIONS, qp.cc:149
        if (iter_out) {                                              // kkt_prev is only ever recorded, qp.cc:118
          float ki[4];
          kkt_errors_sq(mu, ki);
          const float r0 = sqrtf(ki[0]), r1 = sqrtf(ki[1]), r2 = sqrtf(ki[2]), r3 = sqrtf(ki[3]);
          if (lane == 0) {
            float* rec = iter_out + (size_t)it * MO_ITER_RECORD;
            rec[0] = r0; rec[1] = r1; rec[2] = r2; rec[3] = r3;
          }
        }
      }
      // ---------------------------------------------------------------- part B: right-hand side, factorisation, direction
      ka = fresh_args32();
      const bool predictor_pass = use_pc && !guess_pass;
      const float mu_step = m > 0 ? (predictor_pass ? 0.0f : mu) : 0.0f;  // qp.cc:165-187
      float aff = 0.0f, cs_inv = 1.0f;  // aff = ds_aff dz_aff (qp.cc:341), set by the predictor
      if (include_ineq) {
        if (__any((lane < m) && !(cs > 0.0f))) { st = MO_STATUS_NONPOSITIVE_SLACK; break; }  // qp.cc:285
        cs_inv = rcp_f32(cs);
        if (lane < m) {
          const float zs = cz * cs_inv;
          atomicAdd(&diagS[cvar], ca * zs * ca);                                            // qp.cc:296
          atomicAdd(&rhoS[cvar], ca * zs * r_pi + ca * (r_comp + aff - mu_step) * cs_inv);  // qp.cc:340-341
        }
      }
      lds_fence32();
      {
        float dd[NT], rr[NT];
        ldv32<NT>(diagS, j, dd);
        ldv32<NT>(rhoS, j, rr);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
          for (int t = 0; t < 4; ++t) U[c * NB + c][t] += (j == 4 * g + t) ? ((!PAD || natvar32(c, j) < nn) ? dd[c] : 1.0f) : 0.0f;   // padding: identity rows (their right-hand side is zero)
          if (g == 0) tmp[16 * c + j] = -(r_d[c] + rr[c]);          // -r_aug, position order (qp.cc:337-342)
        }
      }
      if (g == 0) ysm[16 + j] = -r_pe;
      lds_fence32();
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const f4 rv = *(const f4*)(tmp + 16 * c + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) U[c * NB + NR][t] = (j == 0) ? rv[t] : 0.0f;
      }
      {
        const f4 rv = *(const f4*)(ysm + 16 + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) U[NT * NB + NR][t] = (j == 0) ? rv[t] : 0.0f;  // -r_pe (zero beyond k)
      }
      __builtin_amdgcn_sched_barrier(0);
      bool ok = true;
#pragma unroll
      for (int pa = 0; pa <= NT; ++pa) {
        ok = sweep_tile_f32(U[pa * NB + pa], pa < NT ? 16 : k, j) && ok;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pc = pa + 1; pc < NB; ++pc) {
          const f4 negZ = mfma4_f32(U[pa * NB + pa], U[pa * NB + pc], f4{0.0f, 0.0f, 0.0f, 0.0f});  // (-T^-1) U_ac
#pragma unroll
          for (int pb = pa + 1; pb <= (pc < NT ? pc : NT); ++pb) U[pb * NB + pc] = mfma4_f32(U[pa * NB + pb], negZ, U[pb * NB + pc]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!ok) { st = MO_STATUS_FACTORIZATION_FAILED; break; }
      float xb[NT + 1];  // xb[c] = dx at position 16c + j, xb[NT] = -dy
#pragma unroll
      for (int pa = NT; pa >= 0; --pa) {
        float vt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float pt = 0.0f;
#pragma unroll
          for (int pb = pa + 1; pb <= NT; ++pb) pt = fmaf(U[pa * NB + pb][t], xb[pb < NT + 1 ? pb : 0], pt);
          pt = row_sum_f32(pt);
          vt[t] = row_bcast_f32<0>(U[pa * NB + NR][t]) - pt;
        }
        float q = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) q = fmaf(U[pa * NB + pa][t], vt[t], q);
        xb[pa] = -cross_row_sum_f32(q);
        __builtin_amdgcn_sched_barrier(0);
      }
      float dyv = 0.0f, dsv = 0.0f, dzv = 0.0f, ap = 1.0f, ad = 1.0f;
      // From a solution xb to the direction: dy, dx (natural order in LDS), ds, dz, the step lengths (qp.cc:359-363, 485-507).
      auto finish_direction = [&](float mu_s, float tau) -> bool {
        dyv = (j < k) ? -xb[NT] : 0.0f;
        bool finite = fabsf(dyv) < INFINITY;
#pragma unroll
        for (int c = 0; c < NT; ++c) finite = finite && (fabsf(xb[c]) < INFINITY);
        if (g == 0) {
          float dxn[NT];
#pragma unroll
          for (int c = 0; c < NT; ++c) dxn[c] = xb[c];
          stv32<NT>(tmp, j, dxn);  // dx, natural order
        }
        lds_fence32();
        ap = 1.0f; ad = 1.0f; dsv = 0.0f; dzv = 0.0f;
        if (lane < m) {
          dsv = ca * tmp[cvar] + r_pi;                                         // qp.cc:361
          dzv = -(cz * cs_inv) * dsv - cs_inv * (r_comp + aff - mu_s);         // qp.cc:362
          if (cs + dsv <= 0.0f && fabsf(dsv) > 0.0f) ap = -tau * cs * rcp_f32(dsv);  // qp.cc:498-503
          if (cz + dzv <= 0.0f && fabsf(dzv) > 0.0f) ad = -tau * cz * rcp_f32(dzv);
          finite = finite && (fabsf(dsv) < INFINITY) && (fabsf(dzv) < INFINITY);
        }
        if (!__all(finite)) return false;
        ap = cross_row_min_f32(row_min_f32(ap));
        ad = cross_row_min_f32(row_min_f32(ad));
        return true;
      };
      if (guess_pass) {                      // qp.cc:455-460: x, y <- the equality-constrained solution
        bool finite = (j < k) ? (fabsf(xb[NT]) < INFINITY) : true;
#pragma unroll
        for (int c = 0; c < NT; ++c) finite = finite && (fabsf(xb[c]) < INFINITY);
        if (!__all(finite)) { st = MO_STATUS_NONFINITE; break; }
#pragma unroll
        for (int c = 0; c < NT; ++c) xv[c] = xb[c];
        yv = (j < k) ? -xb[NT] : 0.0f;
        guess_pass = false;
        clamp_and_init_slacks();
        continue;
      }
      if (!finish_direction(mu_step, predictor_pass ? 1.0f : 0.995f)) { st = MO_STATUS_NONFINITE; break; }  // tau: qp.cc:174, 192
      ip_mu = mu;
      if (predictor_pass) {
        probe_p = ap; probe_d = ad;                                                    // alpha_probe, qp.cc:174
        const float sdz = wave_sum_f32(lane < m ? cs * dzv : 0.0f), zds = wave_sum_f32(lane < m ? cz * dsv : 0.0f),
                    dsdz = wave_sum_f32(lane < m ? dsv * dzv : 0.0f);
        aff = dsv * dzv;                                                               // delta_affine_ (qp.cc:177)
        float ma = mu;                                                                 // qp.cc:519-537
        ma += ad * sdz * inv_m;
        ma += ap * zds * inv_m;
        ma += (ad * ap) * dsdz * inv_m;
        mu_aff = ma > 0.0f ? ma : 0.0f;
        const float ratio = mu_aff * rcp_f32(mu);
        mu_pc = (ratio * ratio * ratio) * mu;                                          // qp.cc:182-183
        // The corrector solve (qp.cc:187): same matrix, right-hand side with ds_aff dz_aff and sigma mu -- through the factors
        for (int i = lane; i < N; i += 64) rhoS[i] = 0.0f;
        lds_fence32();
        if (lane < m) {
          const float zs = cz * cs_inv;
          atomicAdd(&rhoS[cvar], ca * zs * r_pi + ca * (r_comp + aff - mu_pc) * cs_inv);  // qp.cc:340-341
        }
        lds_fence32();
        float rb[NT + 1];
        {
          float rr[NT];
          ldv32<NT>(rhoS, j, rr);
#pragma unroll
          for (int c = 0; c < NT; ++c) rb[c] = -(r_d[c] + rr[c]);
          rb[NT] = -r_pe;  // zero beyond k
        }
        solve_second_rhs_f32<NT>(U, g, j, rb, diagS, xp, ysm + 16, xb);
        if (!finish_direction(mu_pc, 0.995f)) { st = MO_STATUS_NONFINITE; break; }
        ip_mu = mu_pc;
      }
      // x,s += alpha_p (dx,ds) ; y,z += alpha_d (dy,dz), qp.cc:196-199
#pragma unroll
      for (int c = 0; c < NT; ++c) xv[c] = fmaf(xb[c], ap, xv[c]);
      yv = fmaf(dyv, ad, yv);
      cs = fmaf(dsv, ap, cs); cz = fmaf(dzv, ad, cz);
      mu_used = mu; ip_alpha_p = ap; ip_alpha_d = ad;
      ++it;
      if (iterate_mode) {  // outputs of Iterate: delta_ and IPIterationOutputs (structs.hpp:53-64)
        if (ka->delta) {
          float* dp = (float*)ka->delta + p * ka->delta_stride;
          for (int i = lane; i < nn; i += 64) dp[i] = tmp[i];  // dx, natural order
          if (lane < m) { dp[nn + lane] = dsv; dp[nn + m + k + lane] = dzv; }
          if (g == 0 && j < k) dp[nn + m + j] = dyv;
        }
        if (ka->ip_out && lane == 0) {
          float* ip = (float*)ka->ip_out + p * MO_IP_RECORD;
          ip[0] = ip_mu; ip[1] = ap; ip[2] = ad;
          ip[3] = probe_p; ip[4] = probe_d; ip[5] = mu_aff;
        }
        break;
      }
    }

    // ---- outputs: state, termination, iteration count, Lagrange summary, status
    if (!residual_mode) {
      if (g == 0) {
#pragma unroll
        for (int c = 0; c < NT; ++c)
          if (!PAD || natvar32(c, j) < nn) vp[natvar32(c, j)] = xv[c];
        if (j < k) vp[nn + m + j] = yv;
      }
      if (lane < m) { vp[nn + lane] = cs; vp[nn + m + k + lane] = cz; }
    }
    const float ymin = row_min_f32((j < k) ? yv : INFINITY), yabs = -row_min_f32((j < k) ? -fabsf(yv) : INFINITY);
    if (lane == 0) {
      if (ka->termination) ka->termination[p] = term;
      if (ka->num_iterations) ka->num_iterations[p] = it;
      if (ka->status) ka->status[p] = st;
      if (ka->lagrange) {  // qp.cc:539-546
        ((float*)ka->lagrange)[2 * p] = k > 0 ? ymin : nanf32;
        ((float*)ka->lagrange)[2 * p + 1] = k > 0 ? yabs : nanf32;
      }
    }
    wait_vmcnt32<0>();  // nothing of this problem's ring traffic is left in flight (a pass may leave through a break)
    lds_fence32();
    if (last_of_chunk) {
      p = uniform64(next_ticket) + ticket_base;
      chunk_end = p + next_chunk;
    } else {
      ++p;
    }
  }
}

// Standalone linearisation in fp32 (mo_linearize, mo_fill_qp; nonlinear.cc:182-189, residual.hpp:186-226): J streams once through the
// ring into the G tiles, which are written out as the lower triangle of the column-major n x n matrix (strict upper triangle exactly
// zero, residual.hpp:216-220), with c = J^T r and 0.5 |r|^2.
template <int NT, int WPS>
__global__ __launch_bounds__(256 * WPS, WPS) void kkt_fused_f32_linearize_kernel(const KernelArgs a) {
  using C = SolveCfg32<NT>;
  constexpr int N = C::N, NH = C::NH, DPS = C::DPS, SLOT = C::SLOT, D = C::D;
  constexpr int WAVES = 4 * WPS;
  constexpr int WAVE_LDS = D * SLOT > 8192 ? D * SLOT : 8192;   // the ring; after the stream: a 64 x 32 staging block of G (see below)
  __shared__ __attribute__((aligned(16))) char smem_all[WAVES * WAVE_LDS];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* const smem = smem_all + wave * WAVE_LDS;
  const unsigned ring_base = (unsigned)(uintptr_t)smem;
  const int nsteps = a.m_r >> 2;
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
    if (lane_id32() == 0) t = atomicAdd(a.ticket, (unsigned long long)chunk);
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
    const int lane = lane_id32();
    const int g = lane >> 4, j = lane & 15;
    const char* jsrc = reinterpret_cast<const char*>((const float*)a.J + p * a.J_stride + (size_t)g * N + 4 * j);
    const char* rsrc = reinterpret_cast<const char*>((const float*)a.r + p * a.r_stride);
    auto issue = [&](int slot) {
      const unsigned dst = ring_base + slot * SLOT;
#pragma unroll
      for (int h = 0; h < NH; ++h) dma16_f32(jsrc + 256 * h, dst + h * 1024);
      if (lane < 1) dma16_f32(rsrc, dst + NH * 1024);
      jsrc += 4 * N * 4;
      rsrc += 16;
    };
#pragma unroll
    for (int u = 0; u < D; ++u)
      if (u < nsteps) issue(u);
    f4 U[NT * NT];
#pragma unroll
    for (int q = 0; q < NT * NT; ++q) U[q] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    const char* const lane_piece = smem + lane * 16;
    const char* const r_elem = smem + NH * 1024 + 4 * g;
    float cpart[NT], rsq = 0.0f;
#pragma unroll
    for (int c = 0; c < NT; ++c) cpart[c] = 0.0f;
    for (int q0 = 0; q0 < nsteps; q0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int q = q0 + u;
        if (q < nsteps) {
          const int younger = nsteps - 1 - q;
          if (younger >= D - 1) wait_vmcnt32<(D - 1) * DPS>();
          else if (younger == 2) wait_vmcnt32<2 * DPS>();
          else if (younger == 1) wait_vmcnt32<1 * DPS>();
          else wait_vmcnt32<0>();
          float ops[NT];
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            const f4 v = *(const f4*)(lane_piece + u * SLOT + h * 1024);
            ops[4 * h] = v[0]; ops[4 * h + 1] = v[1]; ops[4 * h + 2] = v[2]; ops[4 * h + 3] = v[3];
          }
          const float rq = *(const float*)(r_elem + u * SLOT);
          lds_fence32();
          if (q + D < nsteps) issue(u);
          rsq = fmaf(rq, rq, rsq);
#pragma unroll
          for (int ta = 0; ta < NT; ++ta) {
            cpart[ta] = fmaf(ops[ta], rq, cpart[ta]);
#pragma unroll
            for (int tb = ta; tb < NT; ++tb)
              U[ta * NT + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ops[ta], ops[tb], U[ta * NT + tb], 0, 0, 0);
          }
        }
      }
    }
    wait_vmcnt32<0>();
    const float half_sq = 0.5f * cross_row_sum_f32(rsq);  // every lane of row g holds the sum over its rows 4s + g
    const float lam_in = a.lambda_vec ? ((const float*)a.lambda_vec)[p * a.lambda_vec_stride] : (float)a.lambda;
    const float lam = lam_in > 0.0f ? lam_in : 0.0f;  // nonlinear.cc:187-189
    float* Go = (float*)a.G_out + p * a.G_out_stride;
    const int ld = a.G_out_ld;
    // Whole-line stores (as the fp64 kernel's): the 4 x 4 tiles (4 sa + p, 4 sb + q) ARE the natural block rows [64 sa, +64) x columns
    // [64 sb, +64) (position i of tile c is variable 4 i + (c & 3) of its 64).  Each natural block of the LOWER triangle is staged
    // through the idle ring in two halves of 32 columns (64 x 32 floats = 8 KB, column-major) and leaves as eight 16-byte-per-lane
    // stores, every line of G written whole; the strict upper blocks go out as zeros the same way.
    const bool whole_lines = !(ld & 3) && !(a.G_out_stride & 3) && (((uintptr_t)a.G_out & 15) == 0);
    if (whole_lines) {
      float* const stage = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int sa = 0; sa < NT / 4; ++sa) {
#pragma unroll
        for (int sb = sa; sb < NT / 4; ++sb) {
#pragma unroll 1
          for (int half = 0; half < 2; ++half) {   // (a runtime loop: it indexes no tile register, and unrolling it doubles the code for nothing)
            // Two lane bases carry every staging address, the rest are immediates: element (row natr = 16 g + 4 t + p, column natc = 4 j + q)
            // sits at  natr + 64 (natc & 31) = LA + 4 t + p + 64 q  when staged as it is (columns of this half: lanes with j >> 3 == half)
            // and at   natc + 64 (natr & 31) = LB + q + 64 (4 t + p) when staged transposed (lanes with g >> 1 == half).
            const int LA = 16 * g + 256 * (j & 7), LB = 4 * j + 1024 * (g & 1);
            const bool in_a = (j >> 3) == half, in_b = (g >> 1) == half;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
              for (int qq = (sa == sb ? pp : 0); qq < 4; ++qq) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                  const float v = U[(4 * sa + pp) * NT + 4 * sb + qq][t];
                  const int natr = 16 * g + 4 * t + pp, natc = 4 * j + qq;      // inside their 64-blocks
                  if (sa != sb) {                 // above the diagonal: G(row natc, column natr) of the lower block
                    if (in_b) stage[LB + qq + 64 * (4 * t + pp)] = v;
                  } else if (pp == qq) {          // a symmetric tile: both triangles present
                    if (in_a) stage[LA + 4 * t + pp + 64 * qq] = natr >= natc ? v + (natr == natc ? lam : 0.0f) : 0.0f;
                  } else {                        // its mirror image is not stored: the value goes below the diagonal, a zero above
                    if (in_a) stage[LA + 4 * t + pp + 64 * qq] = natr > natc ? v : 0.0f;
                    if (in_b) stage[LB + qq + 64 * (4 * t + pp)] = natr > natc ? 0.0f : v;
                  }
                }
              }
            }
            lds_fence32();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
              const int e = it * 256 + lane * 4;
              const f4 v = *(const f4*)(stage + e);
              const int col = 32 * half + (e >> 6), row = e & 63;
              *(f4*)(Go + (64 * sb + row) + (size_t)(64 * sa + col) * ld) = v;
              if (sa != sb) *(f4*)(Go + (64 * sa + row) + (size_t)(64 * sb + col) * ld) = f4{0.0f, 0.0f, 0.0f, 0.0f};   // strict upper triangle: exactly zero
            }
            lds_fence32();
          }
        }
      }
    } else {
#pragma unroll
    for (int ta = 0; ta < NT; ++ta) {
#pragma unroll
      for (int tb = ta; tb < NT; ++tb) {
        const int natc = natvar32(tb, j);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int natr = natvar32(ta, 4 * g + t);
          const int hi = natr > natc ? natr : natc, lo = natr > natc ? natc : natr;
          Go[hi + (size_t)lo * ld] = U[ta * NT + tb][t] + (hi == lo ? lam : 0.0f);   // lower triangle (residual.hpp:216-220)
          if (hi != lo) Go[lo + (size_t)hi * ld] = 0.0f;                             // the strict upper triangle stays exactly zero
        }
      }
    }
    }
    float cvec[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) cvec[c] = cross_row_sum_f32(cpart[c]);  // (all lanes take part in the cross-row swaps)
    if (g == 0) {
      float* co = (float*)a.c_out + p * a.c_out_stride;
#pragma unroll
      for (int c = 0; c < NT; ++c) co[natvar32(c, j)] = cvec[c];
    }
    if (lane == 0 && a.half_sq_out) ((float*)a.half_sq_out)[p * (a.half_sq_stride ? a.half_sq_stride : 1)] = half_sq;
    lds_fence32();
    if (last_of_chunk) { p = uniform64(next_ticket) + ticket_base; chunk_end = p + next_chunk; } else { ++p; }
  }
}

bool aligned16_f32(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

bool fused_f32_supported(const KernelArgs& a, int dtype) {
  if (dtype != MO_F32) return false;
  if (a.mode == MODE_LINEARIZE && a.n != 128 && a.n != 64) return false;
  if (a.n < 4 || a.n > 128 || (a.n & 3)) return false;   // step / Solve / Iterate / residual: any multiple of 4, padded inside the kernels to the 64 / 128 grid
  if (a.mode == MODE_LINEARIZE) {  // kkt_fused_f32_linearize_kernel: packed row-major J, rows in whole 4-row groups
    return a.J && a.ticket && a.G_out && a.c_out && a.J_row_major && a.J_ld == a.n && a.m_r > 0 && !(a.m_r & 3) && aligned16_f32(a.J) &&
           !(a.J_stride & 3) && aligned16_f32(a.r) && !(a.r_stride & 3) && a.G_out_ld >= a.n;
  }
  if (a.k > 16 || a.m > 64 || a.m < 0) return false;
  if (!a.ticket || !a.vars) return false;
  if (a.mode == MODE_SOLVE || a.mode == MODE_ITERATE || a.mode == MODE_RESIDUAL) {  // kkt_fused_f32_solve_kernel
    if (a.mode == MODE_RESIDUAL ? ((a.flags & ~MO_STEP_NO_INEQUALITIES) != 0 || !a.r_out) : a.flags != 0) return false;
    if (a.J) {
      if (!a.J_row_major || a.J_ld != a.n || a.m_r <= 0) return false;
      if (!aligned16_f32(a.J) || (a.J_stride & 3)) return false;
    } else if (!a.G || !a.c || a.G_ld < a.n) {
      return false;
    }
    return true;
  }
  if ((a.flags & ~MO_STEP_NO_INEQUALITIES) != 0 || a.mode != MODE_STEP) return false;
  if (!a.delta || !a.J) return false;
  if (!a.J_row_major || a.J_ld != a.n || a.m_r <= 0) return false;
  if (!aligned16_f32(a.J) || (a.J_stride & 3)) return false;
  return true;
}

const char* fused_f32_name(const KernelArgs& a) {
  if (a.mode == MODE_LINEARIZE) return a.n > 64 ? "fused_linearize_f32_n128" : "fused_linearize_f32_n64";
  if (a.mode == MODE_STEP) return a.n > 64 ? "fused_mfma_f32_n128" : "fused_mfma_f32_n64";
  if (!a.J) return a.n > 64 ? "fused_solve_qp_f32_n128" : "fused_solve_qp_f32_n64";
  return a.n > 64 ? "fused_solve_mfma_f32_n128" : "fused_solve_mfma_f32_n64";
}

hipError_t launch_fused_f32(const KernelArgs& a_in, int num_cus, hipStream_t stream) {
  KernelArgs a = a_in;
  // static rounds up to this many problems per wave (mo_kernels.h; measured): equal-cost work (step, Iterate, residual,
  // linearisation) splits statically further than a Solve, whose problems need different numbers of passes
  if (a.static_rounds < 0) a.static_rounds = a.mode == MODE_SOLVE ? (a.n > 32 ? 2 : 6) : (a.n > 32 ? 8 : 32);
#ifdef MO_TUNING   // (A/B builds only: the product library reads no environment variable)
  static const int env_stagger = [] { const char* e = getenv("MO_FUSED_F32_STAGGER"); return e ? atoi(e) : -1; }();
  static const int env_wps = [] { const char* e = getenv("MO_FUSED_F32_WPS"); return e ? atoi(e) : 0; }();
#else
  constexpr int env_stagger = -1, env_wps = 0;
#endif
  a.stagger = env_stagger >= 0 ? env_stagger : 0;
  // The work counter is zeroed on the stream in front of the kernel -- unless the launch is certain to run in static rounds, which never touch
  // it: every kernel below has at least 4 waves per workgroup and min(CUs, ceil(batch / 4)) workgroups, so batch <= rounds x 4 x workgroups
  // is static whatever the instantiation (the kernels test batch <= rounds x waves).  One enqueued operation less per small launch.
  {
    long long wgs = (a.batch + 3) / 4;
    if (wgs > num_cus) wgs = num_cus;
    const bool surely_static = a.static_rounds > 0 && a.batch <= (long long)a.static_rounds * 4 * wgs;
    if (!surely_static) {
      hipError_t e = hipMemsetAsync(a.ticket, 0, sizeof(unsigned long long), stream);
      if (e != hipSuccess) return e;
    }
  }
  if (a.mode == MODE_LINEARIZE) {
    long long grid = num_cus;   // (two waves per SIMD at n = 128, three at n = 64)
    const long long need = (a.batch + 3) / 4;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    if (a.n > 64) hipLaunchKernelGGL((kkt_fused_f32_linearize_kernel<8, 2>), dim3((unsigned)grid), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((kkt_fused_f32_linearize_kernel<4, 3>), dim3((unsigned)grid), dim3(768), 0, stream, a);
    return hipGetLastError();
  }
  if (a.mode != MODE_STEP) {  // Solve / Iterate / KKT residual: one wave per SIMD at n = 128 (216 tile registers + the state), three at n = 64
    long long grid = num_cus;
    const long long need = (a.batch + 3) / 4;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    if (a.n > 64) { if (a.n == 128) hipLaunchKernelGGL((kkt_fused_f32_solve_kernel<8, 1, false>), dim3((unsigned)grid), dim3(256), 0, stream, a); else hipLaunchKernelGGL((kkt_fused_f32_solve_kernel<8, 1, true>), dim3((unsigned)grid), dim3(256), 0, stream, a); }
    else { if (a.n == 64) hipLaunchKernelGGL((kkt_fused_f32_solve_kernel<4, 3, false>), dim3((unsigned)grid), dim3(768), 0, stream, a); else hipLaunchKernelGGL((kkt_fused_f32_solve_kernel<4, 3, true>), dim3((unsigned)grid), dim3(768), 0, stream, a); }
    return hipGetLastError();
  }
#ifdef MO_TUNING
  if (a.n > 64 && env_wps == 1) {
    constexpr int WPS = 1;
    long long grid = num_cus;
    const long long need = (a.batch + 3) / 4;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    if (a.n == 128) hipLaunchKernelGGL((kkt_fused_f32_kernel<8, WPS, false>), dim3((unsigned)grid), dim3(256 * WPS), 0, stream, a);
    else hipLaunchKernelGGL((kkt_fused_f32_kernel<8, WPS, true>), dim3((unsigned)grid), dim3(256 * WPS), 0, stream, a);
  } else
#else
  (void)env_wps;
#endif
  if (a.n > 64) {
    constexpr int WPS = 2;  // 255 VGPRs, no scratch: the 216 accumulator registers + operands just fit two waves per SIMD
    long long grid = num_cus;
    const long long need = (a.batch + 3) / 4;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    if (a.n == 128) hipLaunchKernelGGL((kkt_fused_f32_kernel<8, WPS, false>), dim3((unsigned)grid), dim3(256 * WPS), 0, stream, a);
    else hipLaunchKernelGGL((kkt_fused_f32_kernel<8, WPS, true>), dim3((unsigned)grid), dim3(256 * WPS), 0, stream, a);
  } else {
    constexpr int WPS = 3;
    long long grid = num_cus;
    const long long need = (a.batch + 3) / 4;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    if (a.n == 64) hipLaunchKernelGGL((kkt_fused_f32_kernel<4, WPS, false>), dim3((unsigned)grid), dim3(256 * WPS), 0, stream, a);
    else hipLaunchKernelGGL((kkt_fused_f32_kernel<4, WPS, true>), dim3((unsigned)grid), dim3(256 * WPS), 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace mo
