// coissue_f32.hip -- how much VALU work of one wave fits beside back-to-back fp32 MFMAs (v_mfma_f32_16x16x4_f32, 32 cycles) of ANOTHER wave on the
// same SIMD of gfx950?  Does wave age or s_setprio change it?  One workgroup of 512 threads on one CU; `swap` decides which four waves (one per
// SIMD) stream MFMAs -- the older (0-3) or the younger (4-7) -- the other four issue VALU ops.  The MFMA waves run 4x as long as the VALU waves
// would need alone, so the VALU waves' time IS their rate under load.
// build: hipcc --offload-arch=gfx950 -O3 tools/coissue_f32.hip -o tools/coissue_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int KIND, int NACC, int GAP>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int prio, int swap, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const bool first = swap ? wave >= 4 : wave < 4;
  const bool do_mfma = (mode & 1) && first, do_valu = (mode & 2) && !first;
  if (do_valu && prio) __builtin_amdgcn_s_setprio(3);
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  float r = 0;
  if (do_mfma) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters * (8 / NACC); ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if (GAP == 1) asm volatile("s_nop 7");
        if (GAP == 2) asm volatile("s_nop 7\n\ts_nop 7");
        if (GAP == 3) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
        if (GAP == 4) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3");
        if (GAP == 5) asm volatile("s_sleep 0");
        if (GAP == 6) asm volatile("s_setprio 0");
      }
    }
    for (int i = 0; i < NACC; ++i) r += acc[i][i & 3];
  }
  if (do_valu) {
    const int vit = iters / 2;
    if (KIND == 0) {  // independent f32 FMAs
      float x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
      for (int it = 0; it < vit; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 1.0000001f, 1e-9f);
      }
      for (int i = 0; i < 8; ++i) r += x[i];
    } else {  // dependent f32 FMA chain
      float x = threadIdx.x;
      for (int it = 0; it < vit; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x = fmaf(x, 1.0000001f, 1e-9f);
      }
      r = x;
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int KIND, int NACC, int GAP> int run(const char* name, float* d, long long* dc) {
  const int iters = 20000; long long h[8];
  for (int swap = 0; swap < 2; ++swap)
    for (int prio = 0; prio < 2; ++prio)
      for (int mode = 1; mode <= 3; mode += 2) {
        hipLaunchKernelGGL((k<KIND, NACC, GAP>), dim3(1), dim3(512), 0, 0, d, iters, mode, prio, swap, dc); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL((k<KIND, NACC, GAP>), dim3(1), dim3(512), 0, 0, d, iters, mode, prio, swap, dc); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, dc, 64, hipMemcpyDeviceToHost));
        printf("%-20s acc %d gap %d  mfma waves %s, valu prio %d, %s: %.1f cyc/MFMA, %.2f cyc/VALU op\n", name, NACC, GAP, swap ? "YOUNGER" : "older  ", prio,
               mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both     ", h[swap ? 4 : 0] / (8.0 * iters), h[swap ? 0 : 4] / (8.0 * (iters / 2)));
      }
  return 0;
}
int main() {
  float* d; long long* dc; CK(hipMalloc(&d, 4096)); CK(hipMalloc(&dc, 64));
  run<0, 8, 0>("indep v_fma_f32", d, dc); run<0, 8, 1>("indep v_fma_f32", d, dc); run<0, 8, 2>("indep v_fma_f32", d, dc); run<0, 8, 3>("indep v_fma_f32", d, dc);
  run<0, 8, 4>("indep v_fma_f32", d, dc); run<0, 8, 5>("indep v_fma_f32", d, dc); run<0, 8, 6>("indep v_fma_f32", d, dc);
  run<1, 8, 0>("dependent v_fma_f32", d, dc); run<1, 8, 2>("dependent v_fma_f32", d, dc); run<0, 1, 0>("indep v_fma_f32", d, dc); run<1, 1, 0>("dependent v_fma_f32", d, dc);
  return 0;
}
