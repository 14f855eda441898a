// microbench.hip -- gfx950 facts the fused kernel relies on: f64 MFMA fragment maps, MFMA f64 issue rate,
// DPP row_newbcast semantics, v_rcp_f64 + one Newton step accuracy.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void mfma_layout(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  d4 acc = {0, 0, 0, 0};
  // A is 16x4 row-major [i][k], B is 4x16 row-major [k][j]
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC>
__global__ void mfma_rate(double* out, int iters, long long* cycles) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

__global__ void fma_rate(double* out, int iters, long long* cycles) {
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
  double a = 1.0000001, b = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int K> __device__ inline double row_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ void dpp_test(double* out, double* rcp_out, const double* x) {
  double v = 100.0 * threadIdx.x + 7.0;
  out[threadIdx.x] = row_bcast<5>(v);
  out[64 + threadIdx.x] = row_bcast<15>(v);
  double d = x[threadIdx.x];
  double q = __builtin_amdgcn_rcp(d);
  q = fma(q, fma(-d, q, 1.0), q);
  rcp_out[threadIdx.x] = q;
}

typedef unsigned u2v __attribute__((ext_vector_type(2)));
__global__ void permlane_test(unsigned* out) {
  const unsigned x = threadIdx.x;
  u2v r = __builtin_amdgcn_permlane16_swap(x, x + 1000, false, false);
  u2v q = __builtin_amdgcn_permlane32_swap(x, x + 1000, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1]; out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
  // DPP row ops on lane ids
  out[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  out[320 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  out[384 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);  // row_half_mirror
  out[448 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false);  // row_mirror
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
  std::vector<double> A(64), B(64), D(256), ref(256);
  for (int i = 0; i < 64; ++i) { A[i] = sin(1.0 + i * 0.37); B[i] = cos(0.5 + i * 0.91); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD; long long* dc;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 1 << 22)); CK(hipMalloc(&dc, 8));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  mfma_layout<<<1, 64>>>(dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
  double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - ref[i]));
  printf("mfma_f64_16x16x4 layout check: max err %.3e (%s)\n", err, err < 1e-14 ? "OK" : "MISMATCH");
  long long cyc; const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, int nacc, int blocks, int threads, const char* name) {
    kern<<<blocks, threads>>>(dD, iters, dc); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); kern<<<blocks, threads>>>(dD, iters, dc); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
    double n = (double)iters * nacc;
    printf("%-28s blocks=%5d threads=%4d: %.1f memtime-ticks/op (wave 0), wall %.3f ms -> %.2f ns per op per wave\n", name, blocks, threads, cyc / n, ms, ms * 1e6 / n);
  };
  run(mfma_rate<1>, 1, 1, 64, "mfma f64 dep chain x1");
  run(mfma_rate<2>, 2, 1, 64, "mfma f64 2 acc");
  run(mfma_rate<10>, 10, 1, 64, "mfma f64 10 acc 1 wave");
  run(mfma_rate<10>, 10, 1024, 256, "mfma f64 10 acc 1w/SIMD all");
  run(mfma_rate<10>, 10, 2048, 256, "mfma f64 10 acc 2w/SIMD all");
  run(fma_rate, 8, 1, 64, "v_fma_f64 8 acc 1 wave");
  run(fma_rate, 8, 1024, 256, "v_fma_f64 8 acc 1w/SIMD all");
  std::vector<double> x(64), o(128), rq(64);
  for (int i = 0; i < 64; ++i) x[i] = (i % 2 ? -1 : 1) * (0.37 + i * 13.1);
  double* dx; CK(hipMalloc(&dx, 512)); CK(hipMemcpy(dx, x.data(), 512, hipMemcpyHostToDevice));
  dpp_test<<<1, 64>>>(dD, dD + 128, dx);
  CK(hipMemcpy(o.data(), dD, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(rq.data(), dD + 128, 512, hipMemcpyDeviceToHost));
  int bad = 0; for (int l = 0; l < 64; ++l) { if (o[l] != 100.0 * ((l & ~15) + 5) + 7.0) bad++; if (o[64 + l] != 100.0 * ((l & ~15) + 15) + 7.0) bad++; }
  printf("row_newbcast semantics: %s\n", bad ? "MISMATCH" : "OK");
  double re = 0; for (int i = 0; i < 64; ++i) re = fmax(re, fabs(rq[i] * x[i] - 1.0));
  printf("rcp+1 newton max |q*x-1| = %.3e\n", re);
  unsigned* du; CK(hipMalloc(&du, 512 * 4));
  permlane_test<<<1, 64>>>(du);
  std::vector<unsigned> u(512); CK(hipMemcpy(u.data(), du, 2048, hipMemcpyDeviceToHost));
  const char* nm[8] = {"permlane16_swap(a=x,b=x+1000).ret0", "permlane16_swap.ret1", "permlane32_swap.ret0", "permlane32_swap.ret1",
                       "dpp quad_perm[1,0,3,2]", "dpp quad_perm[2,3,0,1]", "dpp row_half_mirror", "dpp row_mirror"};
  for (int v = 0; v < 8; ++v) {
    printf("%-36s rows(first lane of each 16-row): ", nm[v]);
    for (int r = 0; r < 4; ++r) printf("%4u ", u[v * 64 + 16 * r]);
    printf(" | lanes 0..7: ");
    for (int l = 0; l < 8; ++l) printf("%u ", u[v * 64 + l]);
    printf("\n");
  }
  return 0;
}
