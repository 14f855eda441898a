// coissue.hip -- how much VALU work fits beside back-to-back f64 MFMAs on one SIMD (gfx950)?
// One workgroup of 512 threads on one CU: waves 0-3 (one per SIMD) issue MFMAs, waves 4-7 issue VALU ops of one kind.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void k(double* out, int iters, int mode, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mode & 1) && wave < 4, do_valu = (mode & 2) && wave >= 4;
  long long t0 = __builtin_amdgcn_s_memtime();
  double r = 0;
  if (do_mfma) {
    d4 acc[4] = {d4{0,0,0,0}, d4{0,0,0,0}, d4{0,0,0,0}, d4{0,0,0,0}};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  }
  if (do_valu) {
    if (KIND == 0) {  // independent f64 FMAs
      double x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], 1.0000001, 1e-9);
      }
      for (int i = 0; i < 8; ++i) r += x[i];
    } else if (KIND == 1) {  // independent 32-bit ops
      int x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (x[i] ^ (x[i] >> 3)) + 0x9e3779b9;
      }
      for (int i = 0; i < 8; ++i) r += x[i];
    } else {  // dependent f64 FMA chain
      double x = threadIdx.x;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x = fma(x, 1.0000001, 1e-9);
      }
      r = x;
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int KIND> int run(const char* name, double* d, long long* dc) {
  const int iters = 20000; long long h[8];
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(512), 0, 0, d, iters, mode, dc); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(512), 0, 0, d, iters, mode, dc); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, dc, 64, hipMemcpyDeviceToHost));
    printf("%-22s mode %d (%s): mfma wave %.1f cyc/MFMA, valu wave %.2f cyc/op\n", name, mode, mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both     ",
           h[0] / (4.0 * iters), h[4] / (8.0 * iters * (KIND == 1 ? 3 : 1)));
  }
  return 0;
}
int main() {
  double* d; long long* dc; CK(hipMalloc(&d, 4096)); CK(hipMalloc(&dc, 64));
  run<0>("indep v_fma_f64", d, dc); run<1>("indep 32-bit alu", d, dc); run<2>("dependent v_fma_f64", d, dc);
  return 0;
}
