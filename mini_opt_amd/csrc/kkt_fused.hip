// kkt_fused.hip -- fused single-wavefront Newton-step kernel for gfx950 (MI355X), fp64, n = 32 or 64.
//
// One 64-lane wavefront (= one workgroup) owns one QP at a time and keeps the WHOLE reduced KKT matrix in registers
// as 16x16 tiles in the v_mfma_f64_16x16x4_f64 C/D fragment layout (lane (g = l>>4, j = l&15), register t holds
// element (row g + 4t, column j)).  Nothing but a few small vectors ever touches LDS.
//
//   P1  J (m_r x n, row-major) is streamed from HBM exactly once with 16-byte loads straight into MFMA operand
//       registers (no LDS staging): G = J^T J is accumulated on the matrix cores as the upper block triangle of tiles,
//       c = J^T r on the VALU.  A lane's 16-byte load holds two adjacent columns, which induces a fixed permutation of
//       the variables (position 16c + i  <->  column 32(c>>1) + 2i + (c&1)); the KKT system is solved in that order.
//       [residual.hpp:206-224, nonlinear.cc:182-189]
//   P2-4 lambda and the barrier diagonal Sigma (qp.cc:293-298) go onto the diagonal tiles; A_eq^T and the right-hand
//       side form one more tile column [A_eq^T | rhs].
//   P5  Block LDL^T with 16x16 pivot blocks: each diagonal tile is inverted in registers by a symmetric sweep (wave
//       broadcasts only), the panel Z = T^-1 U and the trailing update U_bc -= U_ab^T Z_c run on the matrix cores.
//       Because the right-hand side rides along as a tile column, the forward substitution is free.  [qp.cc:302]
//   P6  Backward substitution on the VALU, arranged so that no fragment-layout conversion is needed.
//   P7  ds, dz (qp.cc:359-363), alpha (qp.cc:485-507), status, coalesced 16-byte stores of delta.
//
// The system is solved for the NEW iterate (x+, -y+): [G+Sigma, A^T; A, 0] [x+; -y+] = [rhs_x; -b_eq] with
// rhs_x[v] = -c[v] + sum_{i on v} a_i (z_i (s_i - b_i) + mu) / s_i, which is the reference's reduced system
// (qp.cc:255-268) with K [x; -y] added to both sides: identical direction delta = (x+ - x, ds, y+ - y, dz) without the
// G x, A x and A^T y products of EvaluateKKTConditions (qp.cc:404-408).
//
// Roofline: per problem 73.6 KB of algorithmic HBM traffic and 440 f64 MFMAs (320 for J^T J at n = 64); see DESIGN.md.
#include <math.h>

#include "mo_kernels.h"

namespace mo {
namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

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
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + LANE_IN_ROW, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + LANE_IN_ROW, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }
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

// Symmetric sweep of pivots 0..NPIV-1 of a symmetric 16x16 tile held in the C/D layout.  Afterwards the swept block holds
// -T11^-1, the swept x unswept block T11^-1 T12 (the solution for an augmented right-hand-side column) and the unswept
// block the Schur complement.  Returns false if a pivot is zero or not finite.
template <int NPIV_MAX>
__device__ inline bool sweep_tile(d4& T, int npiv, int g, int j, int addr_j, const int (&addr_r)[4]) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < NPIV_MAX; ++k) {
    if (k < npiv) {  // wave-uniform
      constexpr int dummy = 0; (void)dummy;
      const int src_g = k & 3, src_t = k >> 2;
      const double rowreg = T[src_t];  // every broadcast below is taken before any register of T is modified
      const double d = readlane_f64(rowreg, 16 * src_g + k);
      ok = ok && (fabs(d) > 0.0) && (fabs(d) < INFINITY);
      const double inv = rcp_f64(d);
      const double rowk = bpermute_f64(addr_j + 64 * src_g, rowreg);   // T(k, j) for this lane's column j
      const double rk = (j == k) ? -inv : rowk * inv;
      double f[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) f[t] = bpermute_f64(addr_r[t] + 64 * src_g, rowreg);  // T(k, g+4t) = T(g+4t, k)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double a = (j == k) ? 0.0 : T[t];
        double nv = fma(-f[t], rk, a);
        if (t == src_t) nv = (g == src_g) ? rk : nv;
        T[t] = nv;
      }
    }
  }
  return ok;
}

// ---- the kernel ------------------------------------------------------------------------------------------------
// NT = n / 16 (2 or 4).  k <= 14, m <= 64, m_r % 4 == 0 are checked by fused_supported().
template <int NT>
__global__ __launch_bounds__(64, 2) void kkt_fused_f64_kernel(const KernelArgs a) {
  constexpr int N = 16 * NT;
  constexpr int NB = NT + 1;  // tile blocks incl. the [A_eq^T | rhs] column / y row
  constexpr int NH = NT / 2;  // 16-byte loads per J row per lane
  constexpr int PF = NT >= 4 ? 4 : 8;  // J row-groups (4 rows each) in flight per wave (register budget: 256 VGPRs)

  __shared__ __attribute__((aligned(16))) double xs[N];     // x, natural order
  __shared__ __attribute__((aligned(16))) double diagS[N];  // barrier diagonal per variable, natural order
  __shared__ __attribute__((aligned(16))) double rhsS[N];   // inequality part of the right-hand side per variable, natural order
  __shared__ __attribute__((aligned(16))) double rp[N];     // right-hand side, permuted order
  __shared__ __attribute__((aligned(16))) double dxs[N];    // dx, natural order

  const int k = a.k, m = a.m, m_r = a.m_r;

  for (long long p = blockIdx.x; p < a.batch; p += gridDim.x) {
    // Lane coordinates are made opaque once per problem so that nothing derived from them (gather addresses, masks,
    // bpermute addresses) is hoisted out of the problem loop and kept live for the whole kernel: the tile registers
    // need the room (2 waves per SIMD = 256 VGPRs).
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, j = lane & 15;
    const int addr_j = j * 4;
    int addr_r[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) addr_r[t] = (g + 4 * t) * 4;
    const double* Jp = (const double*)a.J + p * a.J_stride;
    const double* rp_g = (const double*)a.r + p * a.r_stride;
    const double* vp = (const double*)a.vars + p * a.vars_stride;
    const double mu = a.mu ? ((const double*)a.mu)[p * a.mu_stride] : 0.0;

    // ---- small loads issued first so that they overlap the J stream
    if (lane < N / 2) {
      const d2 xv = *(const d2*)(vp + 2 * lane);
      xs[2 * lane] = xv[0]; xs[2 * lane + 1] = xv[1];
      diagS[2 * lane] = 0.0; diagS[2 * lane + 1] = 0.0;
      rhsS[2 * lane] = 0.0; rhsS[2 * lane + 1] = 0.0;
    }
    int cvar = 0; double ca = 1.0, cb = 0.0, cs = 1.0, cz = 0.0;
    bool bad_index = false;
    if (lane < m) {
      cvar = a.cons_var[p * a.cons_stride + lane];
      ca = ((const double*)a.cons_a)[p * a.cons_stride + lane];
      cb = ((const double*)a.cons_b)[p * a.cons_stride + lane];
      cs = vp[N + lane];
      cz = vp[N + m + k + lane];
      bad_index = (cvar < 0) || (cvar >= N);
      if (bad_index) cvar = 0;
    }

    // ---- P1: stream J once; G = J^T J on the matrix cores (upper block triangle), c = J^T r on the VALU
    d4 U[NB * NB];
#pragma unroll
    for (int q = 0; q < NB * NB; ++q) U[q] = d4{0.0, 0.0, 0.0, 0.0};
    double cpart[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) cpart[c] = 0.0;
    {
      const int nsteps = m_r >> 2;
      const double* Jl = Jp + (size_t)g * N + 2 * j;  // row g of a 4-row group, this lane's column pair
      const double* rl = rp_g + g;
      d2 buf[PF][NH];
      double rbuf[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (u < nsteps) {
#pragma unroll
          for (int h = 0; h < NH; ++h) buf[u][h] = *(const d2*)(Jl + (size_t)u * 4 * N + 32 * h);
          rbuf[u] = rl[4 * u];
        }
      }
      for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          if (s0 + u < nsteps) {  // wave-uniform
            double ops[NT];
#pragma unroll
            for (int h = 0; h < NH; ++h) { ops[2 * h] = buf[u][h][0]; ops[2 * h + 1] = buf[u][h][1]; }
            const double rq = rbuf[u];
            const int sn = s0 + u + PF;
            if (sn < nsteps) {
#pragma unroll
              for (int h = 0; h < NH; ++h) buf[u][h] = *(const d2*)(Jl + (size_t)sn * 4 * N + 32 * h);
              rbuf[u] = rl[4 * sn];
            }
#pragma unroll
            for (int ta = 0; ta < NT; ++ta) {
              cpart[ta] = fma(ops[ta], rq, cpart[ta]);
#pragma unroll
              for (int tb = ta; tb < NT; ++tb)
                U[ta * NB + tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ops[ta], ops[tb], U[ta * NB + tb], 0, 0, 0);
            }
          }
        }
      }
    }
    double cvec[NT];  // c = J^T r at permuted position 16c + j (replicated over g)
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      double v = cpart[c];
      v += xor_f64(v, 16);
      v += xor_f64(v, 32);
      cvec[c] = v;
    }

    __builtin_amdgcn_sched_barrier(0);
    // ---- P3: per-constraint barrier terms, scattered per variable through LDS (duplicates on one variable accumulate)
    __syncthreads();  // xs / diagS / rhsS initialised
    const bool slack_bad = __any((lane < m) && !(cs > 0.0));
    const bool any_bad_index = __any(bad_index);
    if (lane < m) {
      const double zs = cz / cs;
      atomicAdd(&diagS[cvar], ca * zs * ca);                         // qp.cc:296
      atomicAdd(&rhsS[cvar], ca * (cz * (cs - cb) + mu) / cs);        // x+ form of qp.cc:340-341
    }
    __syncthreads();
    double dS[NT], rS[NT];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const d2 dd = *(const d2*)(&diagS[32 * h + 2 * j]);
      const d2 rr = *(const d2*)(&rhsS[32 * h + 2 * j]);
      dS[2 * h] = dd[0]; dS[2 * h + 1] = dd[1];
      rS[2 * h] = rr[0]; rS[2 * h + 1] = rr[1];
    }
    if (g == 0) {
#pragma unroll
      for (int c = 0; c < NT; ++c) rp[16 * c + j] = rS[c] - cvec[c];
    }
    __syncthreads();

    // ---- P2: lambda + Sigma on the diagonal tiles (position (r, r): lanes with j == g + 4t)
    const double lam = a.lambda > 0.0 ? a.lambda : 0.0;  // nonlinear.cc:187-189
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
      for (int t = 0; t < 4; ++t) U[c * NB + c][t] += (j == g + 4 * t) ? (lam + dS[c]) : 0.0;
    }

    // ---- P4: tile column NT = [A_eq^T | rhs]; y diagonal tile = [0, -b_eq; -b_eq^T, 0]
    {
      // keep the 16 gather addresses of this phase from being hoisted out of the problem loop (they would occupy
      // 32 VGPRs for the whole kernel): make the lane coordinates opaque to loop-invariant code motion
      const int gq = g, jq = j;
      const double* Ap = k > 0 ? (const double*)a.A + p * a.A_stride : nullptr;
      const double* bp = k > 0 ? (const double*)a.b + p * a.b_stride : nullptr;
#pragma unroll
      for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = gq + 4 * t;                                 // row of the tile = permuted variable 16c + r
          const int col = 32 * (c >> 1) + 2 * r + (c & 1);          // its original column
          double v = 0.0;
          if (jq < k) v = Ap[jq + (size_t)col * a.A_ld];
          if (jq == kRC) v = rp[16 * c + r];
          U[c * NB + NT][t] = v;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = gq + 4 * t;
        double v = 0.0;
        if (r < k && jq == kRC) v = -bp[r];
        if (r == kRC && jq < k) v = -bp[jq];
        U[NT * NB + NT][t] = v;
      }
    }

    // ---- P5: block elimination with 16x16 pivot blocks
    __builtin_amdgcn_sched_barrier(0);
    bool ok = true;
#pragma unroll
    for (int pa = 0; pa < NB; ++pa) {
      ok = sweep_tile<16>(U[pa * NB + pa], pa < NT ? 16 : k, g, j, addr_j, addr_r) && ok;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pc = pa + 1; pc < NB; ++pc) {
        d4 negZ = mfma4(U[pa * NB + pa], U[pa * NB + pc], d4{0.0, 0.0, 0.0, 0.0});  // (-T^-1) U_ac  (T^-1 is symmetric)
#pragma unroll
        for (int pb = pa + 1; pb <= pc; ++pb) U[pb * NB + pc] = mfma4(U[pa * NB + pb], negZ, U[pb * NB + pc]);
        __builtin_amdgcn_sched_barrier(0);  // one panel tile at a time: keeps a single -Z tile live (register pressure)
      }
    }

    __builtin_amdgcn_sched_barrier(0);
    // ---- P6: backward substitution; xb[c] = solution at permuted position 16c + j (replicated over g)
    double xb[NB];
    {
      double v = 0.0;  // -y+ sits in column kRC of the swept y tile: element (q, kRC) at lane (q & 3, kRC), register q >> 2
      const int src = (16 * (j & 3) + kRC) * 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double w = bpermute_f64(src, U[NT * NB + NT][t]);
        if ((j >> 2) == t) v = w;
      }
      xb[NT] = (j < k) ? v : 0.0;
    }
#pragma unroll
    for (int pa = NT - 1; pa >= 0; --pa) {
      double vt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        double pt = 0.0;
#pragma unroll
        for (int pb = pa + 1; pb < NB; ++pb) pt = fma(U[pa * NB + pb][t], xb[pb], pt);
        pt += xor_f64(pt, 1); pt += xor_f64(pt, 2); pt += xor_f64(pt, 4); pt += xor_f64(pt, 8);  // sum over the row's 16 lanes
        vt[t] = row_bcast<kRC>(U[pa * NB + NT][t]) - pt;  // forward-eliminated rhs minus the already solved blocks
      }
      double q = 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) q = fma(U[pa * NB + pa][t], vt[t], q);  // (-T^-1) v, summed over this lane's 4 rows
      q += xor_f64(q, 16);
      q += xor_f64(q, 32);
      xb[pa] = -q;
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- P7: direction, step lengths, status
    double dxv[NT];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const d2 xv = *(const d2*)(&xs[32 * h + 2 * j]);
      dxv[2 * h] = xb[2 * h] - xv[0];
      dxv[2 * h + 1] = xb[2 * h + 1] - xv[1];
    }
    bool finite = true;
#pragma unroll
    for (int c = 0; c < NT; ++c) finite = finite && (fabs(dxv[c]) < INFINITY);
    if (g == 0) {
#pragma unroll
      for (int h = 0; h < NH; ++h) { dxs[32 * h + 2 * j] = dxv[2 * h]; dxs[32 * h + 2 * j + 1] = dxv[2 * h + 1]; }
    }
    __syncthreads();
    double dsv = 0.0, dzv = 0.0, ap = 1.0, ad = 1.0;
    if (lane < m) {
      const double r_pi = ca * xs[cvar] + cb - cs;                                  // qp.cc:416
      dsv = ca * dxs[cvar] + r_pi;                                                  // qp.cc:361
      dzv = -(cz / cs) * dsv - (1.0 / cs) * (cs * cz - mu);                         // qp.cc:362
      const double tau = a.tau;
      if (cs + dsv <= 0.0 && fabs(dsv) > 0.0) ap = -tau * cs / dsv;                 // qp.cc:498-503
      if (cz + dzv <= 0.0 && fabs(dzv) > 0.0) ad = -tau * cz / dzv;
      finite = finite && (fabs(dsv) < INFINITY) && (fabs(dzv) < INFINITY);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double u1 = xor_f64(ap, o), u2 = xor_f64(ad, o);
      ap = u1 < ap ? u1 : ap;
      ad = u2 < ad ? u2 : ad;
    }
    const double dyv = (j < k) ? (-xb[NT] - vp[N + m + j]) : 0.0;                   // y+ - y
    finite = finite && (fabs(dyv) < INFINITY);
    int st = MO_STATUS_OK;
    if (!__all(finite)) st = MO_STATUS_NONFINITE;
    if (!ok) st = MO_STATUS_FACTORIZATION_FAILED;
    if (slack_bad) st = MO_STATUS_NONPOSITIVE_SLACK;
    if (any_bad_index) st = MO_STATUS_BAD_INDEX;
    const double nanv = __builtin_nan("");
    double* dp = (double*)a.delta + p * a.delta_stride;
    if (g == 0) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        d2 o;
        o[0] = st == MO_STATUS_OK ? dxv[2 * h] : nanv;
        o[1] = st == MO_STATUS_OK ? dxv[2 * h + 1] : nanv;
        *(d2*)(dp + 32 * h + 2 * j) = o;
      }
      if (j < k) dp[N + m + j] = st == MO_STATUS_OK ? dyv : nanv;
    }
    if (lane < m) {
      dp[N + lane] = st == MO_STATUS_OK ? dsv : nanv;
      dp[N + m + k + lane] = st == MO_STATUS_OK ? dzv : nanv;
    }
    if (lane == 0) {
      if (a.alpha) {
        ((double*)a.alpha)[2 * p] = st == MO_STATUS_OK ? ap : nanv;
        ((double*)a.alpha)[2 * p + 1] = st == MO_STATUS_OK ? ad : nanv;
      }
      if (a.status) a.status[p] = st;
    }
    __syncthreads();  // LDS vectors are re-initialised by the next problem
  }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

bool fused_supported(const KernelArgs& a, int dtype) {
  if (dtype != MO_F64 || a.mode != MODE_STEP || a.flags != 0) return false;
  if (a.n != 32 && a.n != 64) return false;
  if (a.k > 14 || a.m > 64 || a.m < 0) return false;
  if (!a.J || !a.J_row_major || a.J_ld != a.n || a.m_r <= 0 || (a.m_r & 3)) return false;
  if (!aligned16(a.J) || (a.J_stride & 1)) return false;
  if (!aligned16(a.vars) || (a.vars_stride & 1)) return false;
  if (!aligned16(a.delta) || (a.delta_stride & 1)) return false;
  if (!a.delta) return false;
  return true;
}

const char* fused_name(const KernelArgs& a, int) { return a.n == 64 ? "fused_mfma_f64_n64" : "fused_mfma_f64_n32"; }

hipError_t launch_fused(const KernelArgs& a, int, int num_cus, hipStream_t stream) {
  long long grid = (long long)num_cus * 8;  // 2 waves per SIMD
  if (grid > a.batch) grid = a.batch;
  if (grid < 1) grid = 1;
  if (a.n == 64) {
    hipLaunchKernelGGL(kkt_fused_f64_kernel<4>, dim3((unsigned)grid), dim3(64), 0, stream, a);
  } else {
    hipLaunchKernelGGL(kkt_fused_f64_kernel<2>, dim3((unsigned)grid), dim3(64), 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace mo
