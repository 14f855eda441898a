// phase_timer_f32.hip -- DIAGNOSTIC build of the fp32 fused step kernel with s_memtime stamps at the phase boundaries (never shipped: the
// product library is built without MO_F32_STAMPS).  Prints the share of wave time per phase at BASELINE configs[3] (n = 128 / 16 / 64, m_r = 256).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -mllvm -amdgpu-function-calls=false -DMO_F32_STAMPS -DMO_F32_LOOKAHEAD=0 \
//        tools/phase_timer_f32.hip -o tools/phase_timer_f32        usage: tools/phase_timer_f32 [batch] [waves per SIMD: 1 | 2]
#include "../mini_opt_amd/csrc/kkt_fused_f32.hip"

#include <cstdio>
#include <cstring>
#include <vector>

__global__ void fill_uniform32(float* p, size_t n, unsigned seed, float lo, float hi) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
    p[i] = lo + (hi - lo) * (float)((x >> 11) * (1.0 / 9007199254740992.0));
  }
}
__global__ void fill_cons32(int* var, float* a, float* b, float* vars, int n, int k, int m, size_t batch) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= batch * m) return;
  const size_t p = i / m; const int c = i % m;
  var[i] = (c * 7 + (int)(p % 5)) % n; a[i] = (c & 1) ? -1.0f : 1.0f; b[i] = 1.5f;
  const int V = n + 2 * m + k;
  vars[p * V + n + c] = 0.8f + 0.01f * c;           // s
  vars[p * V + n + m + k + c] = 0.5f + 0.02f * c;   // z
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n = 128, k = 16, m = 64, m_r = 256;
  const size_t batch = argc > 1 ? atoll(argv[1]) : 65536;
  if (argc > 2) setenv("MO_FUSED_F32_WPS", argv[2], 1);
  const int V = n + 2 * m + k;
  float *J, *r, *A, *b, *ca, *cb, *vars, *mu, *delta, *alpha; int *cv, *status; unsigned long long* dbg;
  CK(hipMalloc(&J, batch * m_r * n * 4)); CK(hipMalloc(&r, batch * m_r * 4)); CK(hipMalloc(&A, batch * k * n * 4));
  CK(hipMalloc(&b, batch * k * 4)); CK(hipMalloc(&ca, batch * m * 4)); CK(hipMalloc(&cb, batch * m * 4));
  CK(hipMalloc(&cv, batch * m * 4)); CK(hipMalloc(&vars, batch * V * 4)); CK(hipMalloc(&mu, batch * 4));
  CK(hipMalloc(&delta, batch * V * 4)); CK(hipMalloc(&alpha, batch * 8)); CK(hipMalloc(&status, batch * 4));
  CK(hipMalloc(&dbg, 128));
  unsigned long long* ticket; CK(hipMalloc(&ticket, 256));
  fill_uniform32<<<4096, 256>>>(J, batch * m_r * n, 1, -1, 1); fill_uniform32<<<1024, 256>>>(r, batch * m_r, 2, -1, 1);
  fill_uniform32<<<1024, 256>>>(A, batch * k * n, 3, -1, 1); fill_uniform32<<<256, 256>>>(b, batch * k, 4, -1, 1);
  fill_uniform32<<<1024, 256>>>(vars, batch * V, 5, -0.4f, 0.4f); fill_uniform32<<<64, 256>>>(mu, batch, 6, 0.05f, 0.1f);
  fill_cons32<<<(unsigned)((batch * m + 255) / 256), 256>>>(cv, ca, cb, vars, n, k, m, batch);
  CK(hipDeviceSynchronize());
  mo::KernelArgs a; memset(&a, 0, sizeof(a));
  a.n = n; a.k = k; a.m = m; a.m_r = m_r; a.mode = mo::MODE_STEP; a.batch = (long long)batch;
  a.J = J; a.J_stride = (long long)m_r * n; a.J_ld = n; a.J_row_major = 1; a.r = r; a.r_stride = m_r; a.lambda = 1e-3;
  a.A = A; a.A_stride = (long long)k * n; a.A_ld = k; a.b = b; a.b_stride = k;
  a.cons_var = cv; a.cons_a = ca; a.cons_b = cb; a.cons_stride = m;
  a.vars = vars; a.vars_stride = V; a.mu = mu; a.mu_stride = 1; a.tau = 0.995;
  a.delta = delta; a.delta_stride = V; a.alpha = alpha; a.status = status; a.debug = dbg; a.ticket = ticket;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  if (!mo::fused_f32_supported(a, MO_F32)) { printf("fused fp32 kernel does not support this shape\n"); return 1; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(dbg, 0, 128)); CK(hipMemset(ticket, 0, 256));
    CK(hipEventRecord(e0)); CK(mo::launch_fused_f32(a, prop.multiProcessorCount, 0)); CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16]; CK(hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost));
    std::vector<int> st(batch); CK(hipMemcpy(st.data(), status, batch * 4, hipMemcpyDeviceToHost));
    size_t okc = 0; for (size_t i = 0; i < batch; ++i) okc += st[i] == 0;
    if (rep < 2) continue;
    const char* names[7] = {"P0 loop top, ring fill, vector DMAs", "P1 J stream + J^T J MFMA", "P3/P2/P4 constraints, rhs, A tiles", "P5 diagonal sweeps",
                            "P5 panel + trailing MFMA", "P6 backward", "P7 epilogue"};
    double tot = 0; for (int i = 0; i < 7; ++i) tot += (double)h[i];
    printf("fp32 n=%d batch=%zu: %.3f ms (%.2f M steps/s, stamped build), status ok %zu/%zu, waves %llu\n", n, batch, ms, batch / ms / 1e3, okc, batch, h[8]);
    for (int i = 0; i < 7; ++i) printf("  %-38s %9.0f ticks/problem/wave  %5.1f %%\n", names[i], (double)h[i] / batch, 100.0 * h[i] / tot);
    printf("  total %.0f ticks per problem per wave (s_memtime: 100 MHz constant clock => x ~23 for core cycles)\n", tot / batch);
  }
  return 0;
}
