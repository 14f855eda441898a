// mfma_shadow_f32.hip -- inside ONE wave on gfx950: what do K independent VALU / LDS / SALU instructions placed behind each v_mfma_f32_16x16x4_f32
// cost?  (Is there an "MFMA shadow" for the VALU?)  One wave per SIMD, 8 independent accumulators.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_shadow_f32.hip -o tools/mfma_shadow_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int K, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  __shared__ float sm[1024];
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        if (KIND == 0) x[(i + q) & 7] = fmaf(x[(i + q) & 7], 1.0000001f, 1e-9f);
        else if (KIND == 1) x[(i + q) & 7] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[(i + q) & 7]), 0x155, 0xf, 0xf, false));
        else if (KIND == 2) asm volatile("s_nop 0");
        else if (KIND == 3) x[(i + q) & 7] += sm[(threadIdx.x + 64 * q + it) & 1023];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][i & 3] + x[i];
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int K, int KIND> int run(const char* name, float* d, long long* dc) {
  const int iters = 20000; long long h;
  hipLaunchKernelGGL((k<K, KIND>), dim3(1), dim3(256), 0, 0, d, iters, dc); CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<K, KIND>), dim3(1), dim3(256), 0, 0, d, iters, dc); CK(hipDeviceSynchronize());
  CK(hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost));
  printf("%-14s x %d behind each MFMA: %.1f cycles per MFMA\n", name, K, h / (8.0 * iters));
  return 0;
}
int main() {
  float* d; long long* dc; CK(hipMalloc(&d, 4096)); CK(hipMalloc(&dc, 64));
  run<0, 0>("v_fma_f32", d, dc); run<1, 0>("v_fma_f32", d, dc); run<2, 0>("v_fma_f32", d, dc); run<4, 0>("v_fma_f32", d, dc); run<6, 0>("v_fma_f32", d, dc); run<8, 0>("v_fma_f32", d, dc);
  run<1, 1>("dpp mov", d, dc); run<4, 1>("dpp mov", d, dc); run<6, 1>("dpp mov", d, dc);
  run<1, 2>("s_nop 0", d, dc); run<4, 2>("s_nop 0", d, dc); run<7, 2>("s_nop 0", d, dc);
  run<1, 3>("ds_read_b32", d, dc); run<4, 3>("ds_read_b32", d, dc);
  return 0;
}
