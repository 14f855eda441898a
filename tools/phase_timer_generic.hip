// phase_timer_generic.hip [n batch k m m_r] -- DIAGNOSTIC build of the generic kernel (LDS-resident and LARGE) with s_memtime stamps at phase boundaries (never shipped:
// the product library is built without MO_FUSED_STAMPS).  Prints the share of wave time per phase.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DMO_FUSED_STAMPS tools/phase_timer.hip -o tools/phase_timer
#include "../mini_opt_amd/csrc/kkt_fused.hip"
#include "../mini_opt_amd/csrc/kkt_generic.hip"

#include <cstdio>
#include <cstring>
#include <vector>

__global__ void fill_uniform(double* p, size_t n, unsigned seed, double lo, double hi) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
    p[i] = lo + (hi - lo) * ((x >> 11) * (1.0 / 9007199254740992.0));
  }
}
__global__ void fill_cons(int* var, double* a, double* b, double* vars, int n, int k, int m, size_t batch) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= batch * m) return;
  const size_t p = i / m; const int c = i % m;
  const int v = (c * 7 + (int)(p % 5)) % n;
  var[i] = v; a[i] = (c & 1) ? -1.0 : 1.0; b[i] = 1.5;
  const int V = n + 2 * m + k;
  vars[p * V + n + c] = 0.8 + 0.01 * c;           // s
  vars[p * V + n + m + k + c] = 0.5 + 0.02 * c;   // z
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 64;
  const size_t batch = argc > 2 ? atoll(argv[2]) : 65536;
  const int k = argc > 3 ? atoi(argv[3]) : n / 8, m = argc > 4 ? atoi(argv[4]) : n / 2, m_r = argc > 5 ? atoi(argv[5]) : 2 * n;
  const int V = n + 2 * m + k;
  double *J, *r, *A, *b, *ca, *cb, *vars, *mu, *delta, *alpha; int *cv, *status; unsigned long long* dbg;
  CK(hipMalloc(&J, batch * m_r * n * 8)); CK(hipMalloc(&r, batch * m_r * 8)); CK(hipMalloc(&A, batch * k * n * 8));
  CK(hipMalloc(&b, batch * k * 8)); CK(hipMalloc(&ca, batch * m * 8)); CK(hipMalloc(&cb, batch * m * 8));
  CK(hipMalloc(&cv, batch * m * 4)); CK(hipMalloc(&vars, batch * V * 8)); CK(hipMalloc(&mu, batch * 8));
  CK(hipMalloc(&delta, batch * V * 8)); CK(hipMalloc(&alpha, batch * 16)); CK(hipMalloc(&status, batch * 4));
  CK(hipMalloc(&dbg, 128));
  unsigned long long* ticket; CK(hipMalloc(&ticket, 256));
  fill_uniform<<<4096, 256>>>(J, batch * m_r * n, 1, -1, 1); fill_uniform<<<1024, 256>>>(r, batch * m_r, 2, -1, 1);
  fill_uniform<<<1024, 256>>>(A, batch * k * n, 3, -1, 1); fill_uniform<<<256, 256>>>(b, batch * k, 4, -1, 1);
  fill_uniform<<<1024, 256>>>(vars, batch * V, 5, -0.4, 0.4); fill_uniform<<<64, 256>>>(mu, batch, 6, 0.05, 0.1);
  fill_cons<<<(unsigned)((batch * m + 255) / 256), 256>>>(cv, ca, cb, vars, n, k, m, batch);
  CK(hipDeviceSynchronize());
  mo::KernelArgs a; memset(&a, 0, sizeof(a));
  a.n = n; a.k = k; a.m = m; a.m_r = m_r; a.mode = mo::MODE_STEP; a.batch = (long long)batch;
  a.J = J; a.J_stride = (long long)m_r * n; a.J_ld = n; a.J_row_major = 1; a.r = r; a.r_stride = m_r; a.lambda = 1e-3;
  a.A = A; a.A_stride = (long long)k * n; a.A_ld = k; a.b = b; a.b_stride = k;
  a.cons_var = cv; a.cons_a = ca; a.cons_b = cb; a.cons_stride = m;
  a.vars = vars; a.vars_stride = V; a.mu = mu; a.mu_stride = 1; a.tau = 0.995;
  a.delta = delta; a.delta_stride = V; a.alpha = alpha; a.status = status; a.debug = dbg; a.ticket = ticket;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  if (mo::generic_needs_large(a, 8)) {  // H in a global workspace per workgroup of the persistent grid (what mo_api.hip's launch_chosen allocates)
    const size_t per_wg = mo::generic_large_workspace_elems(a);
    const size_t wgs = (size_t)mo::generic_large_grid(a, 8, prop.multiProcessorCount);
    CK(hipMalloc(&a.H_work, wgs * per_wg * 8));
    a.H_work_stride = (long long)per_wg;
    a.H_work_slots = (int)wgs;
    printf("LARGE path: %zu workgroups, %zu B of H each, %zu B of LDS\n", wgs, per_wg * 8, mo::generic_large_lds_bytes(a, 8));
  }

  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(dbg, 0, 128));
    { unsigned long long big = ~0ull; CK(hipMemcpy(dbg + 10, &big, 8, hipMemcpyHostToDevice)); }
    CK(hipEventRecord(e0)); CK(mo::launch_generic(a, MO_F64, prop.multiProcessorCount, 0)); CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16]; CK(hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost));
    std::vector<int> st(batch); CK(hipMemcpy(st.data(), status, batch * 4, hipMemcpyDeviceToHost));
    size_t okc = 0; for (size_t i = 0; i < batch; ++i) okc += st[i] == 0;
    if (rep < 2) continue;
    const char* names[7] = {"load state/cons", "load_qp (zero H, A)", "accumulate_jtj", "eval_kkt",
                            "newton_direction (assemble+LDLT+solve+alpha)", "-", "epilogue writes"};
    double tot = 0; for (int i = 0; i < 7; ++i) tot += (double)h[i];
    printf("n=%d batch=%zu: %.3f ms (%.2f M steps/s, stamped build), status ok %zu/%zu, waves %llu\n", n, batch, ms, batch / ms / 1e3, okc, batch, h[8]);
    if (true) { printf("generic kernel\n"); }
    for (int i = 0; i < 7; ++i) printf("  %-30s %9.0f ticks/problem  %5.1f %%\n", names[i], (double)h[i] / batch, 100.0 * h[i] / tot);
    printf("  total %.0f ticks per problem (s_memtime; on this pool it counts close to core cycles: total x problems per workgroup / launch time)\n", tot / batch);
  }
  unsigned long long nd[16];
  CK(hipMemcpyFromSymbol(nd, HIP_SYMBOL(mo::g_nd_stamps), sizeof(nd)));
  printf("  inside newton_direction (3 launches summed): assemble + factorise %llu, rhs + solve + ds/dz %llu ticks per problem\n",
         nd[0] / (3 * batch), nd[1] / (3 * batch));
  printf("  inside factor_blocked: other (assembly, loop) %llu, accumulate from H %llu, stage %llu, factor half %llu, accumulate from panel %llu ticks per problem; pivot loops alone (wave 0) %llu\n",
         nd[2] / (3 * batch), nd[3] / (3 * batch), nd[4] / (3 * batch), nd[5] / (3 * batch), nd[6] / (3 * batch), nd[7] / (3 * batch));
  printf("  factor_panel_regs (wave 0): LDS loads + barrier %llu, stores issued %llu ticks per problem; fallbacks to the LDS loop: %llu in %llu problems\n",
         nd[9] / (3 * batch), nd[10] / (3 * batch), nd[8], (unsigned long long)(3 * batch));
  return 0;
}
