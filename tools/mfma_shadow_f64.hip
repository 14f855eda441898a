// mfma_shadow_f64.hip -- inside ONE wave on gfx950: what do K independent instructions of one kind placed behind each v_mfma_f64_16x16x4_f64 cost?
// One wave per SIMD, 8 independent accumulators.   build: hipcc --offload-arch=gfx950 -O3 tools/mfma_shadow_f64.hip -o tools/mfma_shadow_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int K, int KIND, bool MFMA>
__global__ __launch_bounds__(256) void k(double* out, int iters, long long* cyc) {
  __shared__ double sm[1024];
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MFMA) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        double& v = x[(i + q) & 7];
        if (KIND == 0) v = fma(v, 1.0000001, 1e-9);
        else if (KIND == 1) v = __longlong_as_double(__builtin_amdgcn_mov_dpp(__double_as_longlong(v), 0x155, 0xf, 0xf, false));
        else if (KIND == 2) { asm volatile("v_mov_b64 %0, %1" : "=v"(v) : "v"(x[(i + q + 3) & 7])); }
        else if (KIND == 3) v += sm[(threadIdx.x + 64 * q + it) & 1023];
        else if (KIND == 4) { int lo = __builtin_amdgcn_readlane(__double2loint(v), 5); asm volatile("" :: "s"(lo)); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][i & 3] + x[i];
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int K, int KIND, bool MFMA> int run(const char* name, double* d, long long* dc) {
  const int iters = 10000; long long h;
  hipLaunchKernelGGL((k<K, KIND, MFMA>), dim3(1), dim3(256), 0, 0, d, iters, dc); CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<K, KIND, MFMA>), dim3(1), dim3(256), 0, 0, d, iters, dc); CK(hipDeviceSynchronize());
  CK(hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost));
  printf("%-16s x %d %s: %.1f cycles per slot\n", name, K, MFMA ? "behind each MFMA" : "alone           ", h / (8.0 * iters));
  return 0;
}
#define BOTH(K, KIND, NAME) run<K, KIND, true>(NAME, d, dc); run<K, KIND, false>(NAME, d, dc)
int main() {
  double* d; long long* dc; CK(hipMalloc(&d, 8192)); CK(hipMalloc(&dc, 64));
  run<0, 0, true>("(nothing)", d, dc);
  BOTH(1, 0, "v_fma_f64"); BOTH(2, 0, "v_fma_f64"); BOTH(4, 0, "v_fma_f64"); BOTH(8, 0, "v_fma_f64");
  BOTH(1, 1, "v_mov_b64 dpp"); BOTH(4, 1, "v_mov_b64 dpp");
  BOTH(1, 2, "v_mov_b64"); BOTH(4, 2, "v_mov_b64");
  BOTH(1, 3, "ds_read_b64+add"); BOTH(4, 3, "ds_read_b64+add");
  BOTH(1, 4, "v_readlane"); BOTH(4, 4, "v_readlane");
  return 0;
}
