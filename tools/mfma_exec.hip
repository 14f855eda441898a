// mfma_exec.hip -- does a VGPR-destination f64 MFMA on gfx950 look at EXEC?  Variants (R = 8 accumulating MFMAs each, result compared with variant 0):
//   0  EXEC = -1 throughout (reference)
//   1  EXEC = 0x00000000ffffffff while the MFMAs issue (half the lanes off), restored behind them
//   2  EXEC narrowed, then restored by s_or_b64 IMMEDIATELY in front of each MFMA (the pattern behind an `if (lane-condition) { DMA }`)
//   3  as 1 with an AGPR destination
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma_exec.hip -o tools/mfma_exec
#include <hip/hip_runtime.h>
#include <cstdio>

#define M_V "v_mfma_f64_16x16x4_f64 v[16:23], %4, %5, v[16:23]\n\t"
#define M_A "v_mfma_f64_16x16x4_f64 a[16:23], %4, %5, a[16:23]\n\t"
#define M_V2 "s_mov_b64 exec, %6\n\ts_or_b64 exec, exec, %7\n\t" M_V
#define R8(P) P P P P P P P P
#define ZERO_V "v_mov_b64 v[16:17], 0\n\tv_mov_b64 v[18:19], 0\n\tv_mov_b64 v[20:21], 0\n\tv_mov_b64 v[22:23], 0\n\t"
#define ZERO_A "v_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\t"
#define READ_V "s_nop 15\n\ts_nop 15\n\tv_mov_b64 %0, v[16:17]\n\tv_mov_b64 %1, v[18:19]\n\tv_mov_b64 %2, v[20:21]\n\tv_mov_b64 %3, v[22:23]\n\t"
#define READ_A "s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 v16, a16\n\tv_accvgpr_read_b32 v17, a17\n\tv_accvgpr_read_b32 v18, a18\n\tv_accvgpr_read_b32 v19, a19\n\tv_accvgpr_read_b32 v20, a20\n\tv_accvgpr_read_b32 v21, a21\n\tv_accvgpr_read_b32 v22, a22\n\tv_accvgpr_read_b32 v23, a23\n\ts_nop 1\n\tv_mov_b64 %0, v[16:17]\n\tv_mov_b64 %1, v[18:19]\n\tv_mov_b64 %2, v[20:21]\n\tv_mov_b64 %3, v[22:23]\n\t"
#define CLOB "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23"

template <int VARIANT>
__global__ void __launch_bounds__(64, 1) k(const double* in, double* out) {
  const int lane = threadIdx.x;
  const double a = in[lane], b = in[64 + lane];
  double v0, v1, v2, v3;
  const unsigned long long half = 0x00000000ffffffffull, rest = 0xffffffff00000000ull;
  if (VARIANT == 0)
    asm volatile(ZERO_V "s_nop 7\n\t" R8(M_V) READ_V : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a), "v"(b), "s"(half), "s"(rest) : CLOB);
  else if (VARIANT == 1)
    asm volatile(ZERO_V "s_nop 7\n\ts_mov_b64 exec, %6\n\ts_nop 7\n\t" R8(M_V) "s_mov_b64 exec, -1\n\t" READ_V
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a), "v"(b), "s"(half), "s"(rest) : CLOB);
  else if (VARIANT == 2)
    asm volatile(ZERO_V "s_nop 7\n\t" R8(M_V2) READ_V : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a), "v"(b), "s"(half), "s"(rest) : CLOB);
  else
    asm volatile(ZERO_A "s_nop 7\n\ts_mov_b64 exec, %6\n\ts_nop 7\n\t" R8(M_A) "s_mov_b64 exec, -1\n\t" READ_A
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a), "v"(b), "s"(half), "s"(rest) : CLOB);
  double* o = out + ((size_t)blockIdx.x * 64 + lane) * 4;
  o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3;
}

int main() {
  double h[128];
  for (int i = 0; i < 128; ++i) h[i] = 0.37 * ((i * 7919) % 101) - 11.0;
  const int blocks = 512;
  double *din, *dout; (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dout, 4 * blocks * 64 * 4 * 8);
  (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  static double r[4][512 * 64 * 4];
  hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, din, dout);
  hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, din, dout + 1 * blocks * 64 * 4);
  hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, din, dout + 2 * blocks * 64 * 4);
  hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, din, dout + 3 * blocks * 64 * 4);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
  for (int v = 1; v < 4; ++v) {
    size_t bad_lo = 0, bad_hi = 0;
    for (size_t i = 0; i < (size_t)blocks * 64; ++i)
      for (int e = 0; e < 4; ++e) {
        const bool bad = r[v][i * 4 + e] != r[0][i * 4 + e];
        if ((i & 63) < 32) bad_lo += bad; else bad_hi += bad;
      }
    printf("variant %d: mismatches in lanes 0-31: %zu, lanes 32-63: %zu (of %d each);  lane 40 value %.6g, reference %.6g\n", v, bad_lo, bad_hi, blocks * 32 * 4,
           r[v][40 * 4], r[0][40 * 4]);
  }
  return 0;
}
