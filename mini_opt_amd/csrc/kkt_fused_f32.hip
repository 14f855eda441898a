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

// EXEC-masked moves with immediate lane masks (see kkt_fused.hip for why the masks are immediates)
template <unsigned long long MASK> __device__ inline void masked_set_f32(float& dst, float src) {
  unsigned long long save;
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_mov_b32 %[d], %[s]\n\t"
               "s_mov_b64 exec, %[sv]\n\ts_nop 4"
               : [d] "+v"(dst), [sv] "=&s"(save)
               : [s] "v"(src), [lo] "i"((unsigned)(MASK & 0xffffffffull)), [hi] "i"((unsigned)(MASK >> 32)));
}
template <unsigned long long MASK> __device__ inline void masked_set_neg_f32(float& dst, float src) {
  unsigned long long save;
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_max_f32 %[d], -%[s], -%[s]\n\t"
               "s_mov_b64 exec, %[sv]\n\ts_nop 4"
               : [d] "+v"(dst), [sv] "=&s"(save)
               : [s] "v"(src), [lo] "i"((unsigned)(MASK & 0xffffffffull)), [hi] "i"((unsigned)(MASK >> 32)));
}
template <unsigned long long MASK> __device__ inline void masked_zero4_f32(f4& T) {
  unsigned long long save;
  float t0 = T[0], t1 = T[1], t2 = T[2], t3 = T[3];
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\tv_mov_b32 %[a], 0\n\tv_mov_b32 %[b], 0\n\t"
               "v_mov_b32 %[c], 0\n\tv_mov_b32 %[d], 0\n\ts_mov_b64 exec, %[sv]\n\ts_nop 4"
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

template <int NT, int WPS> struct Cfg32 {
  static constexpr int N = 16 * NT;
  static constexpr int NH = NT / 4;                // 16-byte J pieces per lane per 4-row group
  static constexpr int DPS = NH + 1;               // DMA instructions per group (J pieces + 16 B of r)
  static constexpr int SLOT = NH * 1024 + 64;
  static constexpr int D = 4;                      // ring depth
  static constexpr int VEC = (3 * N + 4 * 64 + 64 + 32) * 4;  // xs, diagS|rp, rhsS|dxs, cons a/b/s/z, cons var, y, b_eq
  static constexpr int LDS = D * SLOT + VEC;
};

template <int NT, int WPS>
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

  const int k = a.k, m = a.m, m_r = a.m_r;
  const int nsteps = m_r >> 2;

  const int chunk_shift = 63 - __builtin_clzll((unsigned long long)gridDim.x * WAVES * 4);
  auto chunk_for = [&](long long observed) -> int {
    const long long c = (a.batch - observed) >> chunk_shift;
    return c < 1 ? 1 : (c > 8 ? 8 : (int)c);
  };
  auto take_ticket = [&](int chunk) -> unsigned long long {
    unsigned long long t = 0;
    if (lane_id32() == 0) t = atomicAdd(a.ticket, (unsigned long long)chunk);
    return t;
  };
  auto uniform64 = [](unsigned long long v) -> long long {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  if (a.stagger > 0) {  // start stagger between the waves that share a SIMD (see kkt_fused.hip): equal-cost problems keep waves in lockstep
    const int slot = wave >> 2;
    for (int i = 0; i < slot * a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  int chunk = chunk_for(0);
  long long p = uniform64(take_ticket(chunk));
  long long chunk_end = p + chunk;

  while (p < a.batch) {
    const bool last_of_chunk = p + 1 >= chunk_end;
    int next_chunk = 0;
    unsigned long long next_ticket = 0;
    if (last_of_chunk) {
      next_chunk = chunk_for(p);
      next_ticket = take_ticket(next_chunk);
    }
    const int lane = lane_id32();
    const int g = lane >> 4, j = lane & 15;

    // ---- J stream set-up: lane (g, j) of 4-row group s fetches J(4s + g, 64h + 4j .. +3), h < NH; lane 0 fetches r[4s .. 4s+3]
    const char* jsrc = reinterpret_cast<const char*>((const float*)a.J + p * a.J_stride + (size_t)g * N + 4 * j);
    const char* rsrc = reinterpret_cast<const char*>((const float*)a.r + p * a.r_stride);
    const char* const lane_piece = smem + lane * 16;
    const char* const r_elem = smem + NH * 1024 + 4 * g;
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

    // ---- P0: small vectors global -> LDS by DMA (no VGPRs held while J streams)
    KArgs32 ka = fresh_args32();
    const float* vp = (const float*)ka->vars + p * ka->vars_stride;
    f4 U[NB * NB];
#pragma unroll
    for (int q = 0; q < NB * NB; ++q) U[q] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    dma_words(vp, vec_base, N, lane);
    if (m > 0) {
      const long long coff = p * ka->cons_stride;
      dma_words(ka->cons_var + coff, vec_base + (3 * N + 256) * 4, m, lane);
      dma_words((const float*)ka->cons_a + coff, vec_base + (3 * N) * 4, m, lane);
      dma_words((const float*)ka->cons_b + coff, vec_base + (3 * N + 64) * 4, m, lane);
      dma_words(vp + N, vec_base + (3 * N + 128) * 4, m, lane);
      dma_words(vp + N + m + k, vec_base + (3 * N + 192) * 4, m, lane);
    }
    if (k > 0) {
      dma_words(vp + N + m, vec_base + (3 * N + 320) * 4, k, lane);
      dma_words((const float*)ka->b + p * ka->b_stride, vec_base + (3 * N + 336) * 4, k, lane);
    }

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
    float cvec[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) cvec[c] = cross_row_sum_f32(cpart[c]);

    // ---- P3: barrier terms scattered per variable through LDS (duplicates on one variable accumulate)
    wait_vmcnt32<0>();
    ka = fresh_args32();
    const float mu = ka->mu ? ((const float*)ka->mu)[p * ka->mu_stride] : 0.0f;
    for (int i = lane; i < N; i += 64) { diagS[i] = 0.0f; rhsS[i] = 0.0f; }
    int cvar = 0; float ca = 1.0f, cb = 0.0f, cs = 1.0f, cz = 0.0f;
    if (lane < m) { cvar = cV[lane]; ca = cA[lane]; cb = cB[lane]; cs = cS[lane]; cz = cZ[lane]; }
    lds_fence32();
    bool bad_index = (lane < m) && ((cvar < 0) || (cvar >= N));
    if (bad_index) cvar = 0;
    const bool slack_bad = __any((lane < m) && !(cs > 0.0f));
    const bool any_bad_index = __any(bad_index);
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
          U[c * NB + c][t] += (j == 4 * g + t) ? (lam + dS[c]) : 0.0f;
          const int natcol = 64 * (c >> 2) + 16 * g + 4 * t + (c & 3);  // variable at position 16c + 4g + t
          U[c * NB + NT][t] = (j < k) ? Ap[j + (size_t)natcol * A_ld] : 0.0f;
          const float rv = rp[16 * c + 4 * g + t];
          U[c * NB + NR][t] = (j == 0) ? rv : 0.0f;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) U[NT * NB + NR][t] = (j == 0 && 4 * g + t < k) ? -bb[4 * g + t] : 0.0f;  // -b_eq
    }

    // ---- P5: block elimination with 16x16 pivot blocks (pivot blocks 0..NT; column NR only rides along)
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
    __builtin_amdgcn_sched_barrier(0);

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
        vt[t] = row_bcast_f32<0>(U[pa * NB + NR][t]) - pt;
      }
      float q = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) q = fmaf(U[pa * NB + pa][t], vt[t], q);
      xb[pa] = -cross_row_sum_f32(q);
      __builtin_amdgcn_sched_barrier(0);
    }

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
      stv32<NT>(dp, j, outv);
      if (j < k) dp[N + m + j] = st == MO_STATUS_OK ? dyv : nanv;
    }
    if (lane < m) {
      dp[N + lane] = st == MO_STATUS_OK ? dsv : nanv;
      dp[N + m + k + lane] = st == MO_STATUS_OK ? dzv : nanv;
    }
    if (lane == 0) {
      if (ka->alpha) {
        ((float*)ka->alpha)[2 * p] = st == MO_STATUS_OK ? ap : nanv;
        ((float*)ka->alpha)[2 * p + 1] = st == MO_STATUS_OK ? ad : nanv;
      }
      if (ka->status) ka->status[p] = st;
    }
    lds_fence32();  // the LDS vectors are re-initialised by the next problem
    if (last_of_chunk) {
      p = uniform64(next_ticket);
      chunk_end = p + next_chunk;
    } else {
      ++p;
    }
  }
}

bool aligned16_f32(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

bool fused_f32_supported(const KernelArgs& a, int dtype) {
  if (dtype != MO_F32 || a.flags != 0 || a.mode != MODE_STEP) return false;
  if (a.n != 128 && a.n != 64) return false;
  if (a.k > 16 || a.m > 64 || a.m < 0) return false;
  if (!a.ticket || !a.vars || !a.delta || !a.J) return false;
  if (!a.J_row_major || a.J_ld != a.n || a.m_r <= 0 || (a.m_r & 3)) return false;
  if (!aligned16_f32(a.J) || (a.J_stride & 3)) return false;
  if (!aligned16_f32(a.r) || (a.r_stride & 3)) return false;
  if (!aligned16_f32(a.delta) || (a.delta_stride & 3)) return false;
  return true;
}

const char* fused_f32_name(const KernelArgs& a) { return a.n == 128 ? "fused_mfma_f32_n128" : "fused_mfma_f32_n64"; }

hipError_t launch_fused_f32(const KernelArgs& a_in, int num_cus, hipStream_t stream) {
  KernelArgs a = a_in;
  static const int env_stagger = [] { const char* e = getenv("MO_FUSED_F32_STAGGER"); return e ? atoi(e) : -1; }();  // A/B knob
  a.stagger = env_stagger >= 0 ? env_stagger : 0;
  hipError_t e = hipMemsetAsync(a.ticket, 0, sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  static const int env_wps = [] { const char* e = getenv("MO_FUSED_F32_WPS"); return e ? atoi(e) : 0; }();  // tuning knob
  if (a.n == 128 && env_wps == 1) {
    constexpr int WPS = 1;
    long long grid = num_cus;
    const long long need = (a.batch + 4 * WPS - 1) / (4 * WPS);
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((kkt_fused_f32_kernel<8, WPS>), dim3((unsigned)grid), dim3(256 * WPS), 0, stream, a);
  } else if (a.n == 128) {
    constexpr int WPS = 2;  // 255 VGPRs, no scratch: the 216 accumulator registers + operands just fit two waves per SIMD
    long long grid = num_cus;
    const long long need = (a.batch + 4 * WPS - 1) / (4 * WPS);
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((kkt_fused_f32_kernel<8, WPS>), dim3((unsigned)grid), dim3(256 * WPS), 0, stream, a);
  } else {
    constexpr int WPS = 3;
    long long grid = num_cus;
    const long long need = (a.batch + 4 * WPS - 1) / (4 * WPS);
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((kkt_fused_f32_kernel<4, WPS>), dim3((unsigned)grid), dim3(256 * WPS), 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace mo
