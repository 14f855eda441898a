// mfma_mix.hip -- does gfx950 keep a VGPR-destination f64 MFMA and an AGPR-destination f64 MFMA apart when their register INDICES coincide
// (v[16:23] then a[16:23])?  Three instruction streams of R pairs each, results compared with a stream that separates every MFMA by nops:
//   0  v[16:23] / a[16:23] back to back   1  v[16:23] / a[32:39] back to back   2  as 0 with 2 x s_nop 15 between all MFMAs (reference)
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma_mix.hip -o tools/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

#define PAIR_SAME "v_mfma_f64_16x16x4_f64 v[16:23], %8, %9, v[16:23]\n\tv_mfma_f64_16x16x4_f64 a[16:23], %9, %8, a[16:23]\n\t"
#define PAIR_DIFF "v_mfma_f64_16x16x4_f64 v[16:23], %8, %9, v[16:23]\n\tv_mfma_f64_16x16x4_f64 a[32:39], %9, %8, a[32:39]\n\t"
#define PAIR_SLOW "v_mfma_f64_16x16x4_f64 v[16:23], %8, %9, v[16:23]\n\ts_nop 15\n\ts_nop 15\n\tv_mfma_f64_16x16x4_f64 a[16:23], %9, %8, a[16:23]\n\ts_nop 15\n\ts_nop 15\n\t"
#define R8(P) P P P P P P P P
#define ZERO_V "v_mov_b64 v[16:17], 0\n\tv_mov_b64 v[18:19], 0\n\tv_mov_b64 v[20:21], 0\n\tv_mov_b64 v[22:23], 0\n\t"
#define ZERO_A(b) "v_accvgpr_write_b32 a" #b ", 0\n\t"
#define READ_V "s_nop 15\n\ts_nop 15\n\tv_mov_b64 %0, v[16:17]\n\tv_mov_b64 %1, v[18:19]\n\tv_mov_b64 %2, v[20:21]\n\tv_mov_b64 %3, v[22:23]\n\t"
#define CLOB "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39"

template <int VARIANT>
__global__ void __launch_bounds__(64, 1) mix(const double* in, double* out) {
  const int lane = threadIdx.x;
  const double a = in[lane], b = in[64 + lane];
  double v0, v1, v2, v3;
  int q[8];
  if (VARIANT == 0) {
    asm volatile(ZERO_V ZERO_A(16) ZERO_A(17) ZERO_A(18) ZERO_A(19) ZERO_A(20) ZERO_A(21) ZERO_A(22) ZERO_A(23) "s_nop 7\n\t" R8(PAIR_SAME) READ_V
                 "v_accvgpr_read_b32 %4, a16\n\tv_accvgpr_read_b32 %5, a17\n\tv_accvgpr_read_b32 %6, a18\n\tv_accvgpr_read_b32 %7, a19\n\t"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(a), "v"(b) : CLOB);
  } else if (VARIANT == 1) {
    asm volatile(ZERO_V ZERO_A(32) ZERO_A(33) ZERO_A(34) ZERO_A(35) ZERO_A(36) ZERO_A(37) ZERO_A(38) ZERO_A(39) "s_nop 7\n\t" R8(PAIR_DIFF) READ_V
                 "v_accvgpr_read_b32 %4, a32\n\tv_accvgpr_read_b32 %5, a33\n\tv_accvgpr_read_b32 %6, a34\n\tv_accvgpr_read_b32 %7, a35\n\t"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(a), "v"(b) : CLOB);
  } else {
    asm volatile(ZERO_V ZERO_A(16) ZERO_A(17) ZERO_A(18) ZERO_A(19) ZERO_A(20) ZERO_A(21) ZERO_A(22) ZERO_A(23) "s_nop 7\n\t" R8(PAIR_SLOW) READ_V
                 "v_accvgpr_read_b32 %4, a16\n\tv_accvgpr_read_b32 %5, a17\n\tv_accvgpr_read_b32 %6, a18\n\tv_accvgpr_read_b32 %7, a19\n\t"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(a), "v"(b) : CLOB);
  }
  double* o = out + (size_t)blockIdx.x * 64 * 8 + lane * 8;
  o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3;
  o[4] = __hiloint2double(q[1], q[0]); o[5] = __hiloint2double(q[3], q[2]); o[6] = 0; o[7] = 0;
}

int main() {
  double h[128];
  for (int i = 0; i < 128; ++i) h[i] = 0.37 * ((i * 7919) % 101) - 11.0;
  double *din, *dout; hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 3 * 1024 * 64 * 8 * 8);
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  const int blocks = 1024;
  static double r[3][1024 * 64 * 8];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(mix<0>, dim3(blocks), dim3(64), 0, 0, din, dout);
    hipLaunchKernelGGL(mix<1>, dim3(blocks), dim3(64), 0, 0, din, dout + blocks * 64 * 8);
    hipLaunchKernelGGL(mix<2>, dim3(blocks), dim3(64), 0, 0, din, dout + 2 * blocks * 64 * 8);
    hipDeviceSynchronize();
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    for (int v = 0; v < 2; ++v) {
      size_t bad_v = 0, bad_a = 0;
      for (size_t i = 0; i < (size_t)blocks * 64; ++i) {
        for (int e = 0; e < 4; ++e) bad_v += r[v][i * 8 + e] != r[2][i * 8 + e];
        for (int e = 4; e < 6; ++e) bad_a += r[v][i * 8 + e] != r[2][i * 8 + e];
      }
      printf("rep %d variant %d (%s): VGPR-tile mismatches %zu, AGPR-tile mismatches %zu of %d lanes x (4 | 2) values;  sample v %.6g a %.6g ref v %.6g a %.6g\n",
             rep, v, v == 0 ? "same index" : "different index", bad_v, bad_a, blocks * 64, r[v][0], r[v][4], r[2][0], r[2][4]);
    }
  }
  return 0;
}
