// dpp_fmac.hip -- gfx950 fact check for the sweep: v_fmac_f64_dpp with row_newbcast, the broadcast source being the destination register
// itself and negated:  T[l] <- T[l] - T[row(l) * 16 + K] * rk[l]   in ONE instruction (instead of v_mov_b64_dpp + v_fma_f64).
// build: hipcc --offload-arch=gfx950 -O2 tools/dpp_fmac.hip -o tools/dpp_fmac
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K> __global__ void k_fmac(double* t, const double* rk) {
  double a = t[threadIdx.x];
  const double b = rk[threadIdx.x];
  asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "i"(K));
  t[threadIdx.x] = a;
}
template <int K> int run() {
  std::vector<double> T(64), R(64), E(64);
  for (int l = 0; l < 64; ++l) { T[l] = 1.0 + 0.37 * l + 0.01 * l * l; R[l] = 0.5 - 0.013 * l; }
  for (int l = 0; l < 64; ++l) E[l] = __builtin_fma(-T[(l / 16) * 16 + K], R[l], T[l]);
  double *dt, *dr;
  hipMalloc(&dt, 512); hipMalloc(&dr, 512);
  hipMemcpy(dt, T.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dr, R.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_fmac<K>, dim3(1), dim3(64), 0, 0, dt, dr);
  std::vector<double> G(64);
  hipMemcpy(G.data(), dt, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) bad += G[l] != E[l];
  printf("row_newbcast:%d  v_fmac_f64_dpp d, -d, s : %s (%d lanes differ)\n", K, bad ? "MISMATCH" : "bit-exact vs fma(-T[bcast], rk, T)", bad);
  return bad;
}
int main() { return run<0>() + run<3>() + run<7>() + run<15>(); }
