// valu_cost.hip -- issue cost (cycles per instruction, one wave alone on its SIMD, independent instructions) of the f64 VALU forms the
// sweep is built from on gfx950.  build: hipcc --offload-arch=gfx950 -O2 tools/valu_cost.hip -o tools/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BODY(name, ASM)                                                                                         \
  __global__ void name(double* out, unsigned long long* cyc) {                                                  \
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    double b = a0 * 0.5;                                                                                        \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                       \
    for (int i = 0; i < 256; ++i) { asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); } \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                       \
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                   \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                            \
  }
BODY(k_fma, "v_fma_f64 %0, %0, %8, %1\n\tv_fma_f64 %2, %2, %8, %3\n\tv_fma_f64 %4, %4, %8, %5\n\tv_fma_f64 %6, %6, %8, %7\n\t")
BODY(k_mul, "v_mul_f64 %0, %0, %8\n\tv_mul_f64 %2, %2, %8\n\tv_mul_f64 %4, %4, %8\n\tv_mul_f64 %6, %6, %8\n\t")
BODY(k_mov64, "v_mov_b64 %0, %1\n\tv_mov_b64 %2, %3\n\tv_mov_b64 %4, %5\n\tv_mov_b64 %6, %7\n\t")
BODY(k_movdpp, "v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %6, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")
BODY(k_fmacdpp, "v_fmac_f64_dpp %0, -%1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, -%3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, -%5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, -%7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")
BODY(k_fmac, "v_fmac_f64 %0, %1, %8\n\tv_fmac_f64 %2, %3, %8\n\tv_fmac_f64 %4, %5, %8\n\tv_fmac_f64 %6, %7, %8\n\t")
BODY(k_rcp, "v_rcp_f64 %0, %1\n\tv_rcp_f64 %2, %3\n\tv_rcp_f64 %4, %5\n\tv_rcp_f64 %6, %7\n\t")
template <typename K> void run(const char* name, K k) {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 64 * 8 * 4096); (void)hipMalloc(&cyc, 8 * 4096);
  (void)hipMemset(out, 0, 64 * 8 * 4096);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, cyc);   // one wave per SIMD
  (void)hipDeviceSynchronize();
  unsigned long long h[1024]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  printf("%-28s %6.2f cycles per instruction (s_memtime ticks)\n", name, s / 1024 / (256.0 * 8 * 4));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run("v_fma_f64", k_fma); run("v_fmac_f64", k_fmac); run("v_mul_f64", k_mul); run("v_mov_b64", k_mov64); run("v_mov_b64_dpp newbcast", k_movdpp);
  run("v_fmac_f64_dpp newbcast", k_fmacdpp); run("v_rcp_f64", k_rcp);
  return 0;
}
