// kkt_fused_tiny.hip -- the interior-point kernels for the sizes mini_opt's own problems have (qp_test.cc, nonlinear_test.cc: n = 2 ... 8):
// n + k <= 15, m <= 64.  The WHOLE reduced KKT matrix [[G + Sigma, A_eq^T], [A_eq, 0]] with its right-hand side in column 15 is ONE
// 16x16 tile in the v_mfma_f64_16x16x4_f64 C/D layout (lane (g = l >> 4, j = l & 15), register t <-> row g + 4t, col j): 8 VGPRs per
// problem, one wavefront per QP, natural variable order.  The factorisation is the symmetric sweep of kkt_fused.hip over the n + k pivots
// (x rows, then y rows -- the order of the block elimination there): afterwards column 15 holds the solution [dx; -dy] and the swept block
// -K^-1, through which Mehrotra's corrector is one tile x vector product.  No matrix cores: at these sizes J^T J is a handful of FMAs.
// Modes: Solve (qp.cc:100-151, all initial guesses and barrier strategies), Iterate (qp.cc:153-201), the bare step in the reference's
// residual form (qp.cc:275-364, 485-507; also MO_STEP_NO_INEQUALITIES, qp.cc:366-386) and EvaluateKKTConditions + ComputeErrors
// (qp.cc:391-437).  Input: (G, c) or (J, r, lambda) with J in any layout of the C ABI and m_r <= 64; G and c stay in registers for the
// whole Solve.  A translation unit of its own (kkt_fused.hip supplies the sweep and the cross-lane helpers).
#define MO_FUSED_IMPL_ONLY
#include "kkt_fused.hip"

namespace mo {
namespace {

constexpr int kTinyLds = 6 * 16 * 8;  // per wave: x, sum a z, Sigma, rho, a scratch vector, a hop for the row-layout copies

// y(j) = sum_i T(i, j) v(i) over the tile rows i < rows; v is V16 (value at lane j, replicated over g); the row-layout copy goes through
// a 16-double LDS hop.  T symmetric in the block that matters.
__device__ inline double tile_times_vector(const d4& T, double v, int rows, int g, int j, double* hop) {
  if (g == 0) hop[j] = v;
  lds_fence();
  double acc = 0.0;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc = fma(T[t], (g + 4 * t < rows) ? hop[g + 4 * t] : 0.0, acc);
  const double y = cross_row_sum(acc);
  lds_fence();  // hop has been read
  return y;
}

template <int WPS>
__global__ __launch_bounds__(256 * WPS, WPS) void kkt_tiny_kernel(const KernelArgs a) {
  constexpr int WAVES = 4 * WPS;
  __shared__ __attribute__((aligned(16))) char smem_all[WAVES * kTinyLds];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* const xs = reinterpret_cast<double*>(smem_all + wave * kTinyLds);  // x by variable
  double* const azS = xs + 16;     // sum a z per variable
  double* const diagS = azS + 16;  // barrier diagonal per variable
  double* const rhoS = diagS + 16; // inequality part of r_aug per variable
  double* const tmp = rhoS + 16;   // dx by variable / right-hand side by row
  double* const hop = tmp + 16;

  // The argument block is read through the kernarg segment at the use sites (fresh_args, kkt_fused.hip): held in SGPRs for the whole
  // kernel it overflows the scalar file, and the spills cost VGPRs (214 spilled SGPRs, 149 VGPRs in a first version).
  const int n = a.n, k = a.k, m = a.m, P = n + k;
  const double inv_m = m > 0 ? 1.0 / (double)m : 0.0;
  const int mode = a.mode;
  const long long batch = a.batch;
  const bool qpl = a.J == nullptr;

  // Work distribution: tickets of up to 64 problems from the device counter; inside a ticket the `skip` words of the caller's outer loop
  // (mo_nls_solve) are read by all lanes at once and only the active problems are visited.
  const int chunk_shift = 63 - __builtin_clzll((unsigned long long)gridDim.x * WAVES * 4);
  // guided tickets (a quarter of the remaining share per wave, at least one problem); with `skip` words at least eight, so that a launch
  // whose problems are mostly finished does not spend its time on the counter
  const long long per_wave = batch / ((long long)gridDim.x * WAVES);
  const bool sparse = a.skip_active >= 0 && a.skip_active * 8 < batch;  // mostly finished: whole-wave tickets, the skip scan is what costs
  const int floor_chunk = (a.skip && per_wave >= 8) ? (sparse ? 64 : 8) : 1;
  auto chunk_for = [&](long long observed) -> int {
    const long long c = (batch - observed) >> chunk_shift;
    return c < floor_chunk ? floor_chunk : (c > 64 ? 64 : (int)c);
  };
  auto take_ticket = [&](int chunk) -> unsigned long long {
    unsigned long long t = 0;
    if (lane_id() == 0) t = atomicAdd(fresh_args()->ticket, (unsigned long long)chunk);
    return t;
  };
  auto uniform64 = [](unsigned long long v) -> long long {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  // the first ticket of every wave is static (see kkt_fused.hip): tickets from the counter start behind that part
  int chunk = chunk_for(0);
  const long long ticket_base = (long long)gridDim.x * WAVES * chunk;
  long long base = ((long long)blockIdx.x * WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * chunk;

  while (base < batch) {
    const int len = (batch - base < chunk) ? (int)(batch - base) : chunk;
    const int next_chunk = chunk_for(base);
    const unsigned long long next_ticket = take_ticket(next_chunk);   // asked for early: its latency hides under this ticket's problems
    unsigned long long active;
    {
      KArgs kq = fresh_args();
      const int l0 = lane_id();
      int flag = -1;
      if (kq->skip && l0 < len) flag = kq->skip[(base + l0) * kq->skip_stride];  // >= 0: finished in the caller's outer loop
      active = __ballot(l0 < len && flag < 0);
    }
  while (active != 0) {
    const int bit = __builtin_ctzll(active);
    active &= active - 1;
    const long long p = base + bit;
    KArgs ka = fresh_args();
    const int lane = lane_id();
    const int g = lane >> 4, j = lane & 15;

    // ---- the problem: K0 = [[G, A^T], [A, 0]] (both triangles, no Sigma) and q = [c; b_eq], kept in registers for all passes
    d4 K0 = d4{0.0, 0.0, 0.0, 0.0};
    double qv = 0.0;
    if (qpl) {  // only the lower triangle of G is read (qp.cc:289, 404)
      const double* Gp = (const double*)ka->G + p * ka->G_stride;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = g + 4 * t;
        if (i < n && j < n) K0[t] = Gp[(i > j ? i : j) + (size_t)(i > j ? j : i) * ka->G_ld];
      }
      if (j < n) qv = ((const double*)ka->c + p * ka->c_stride)[j];
    } else {    // G = J^T J + lambda I, c = J^T r (residual.hpp:206-224, nonlinear.cc:187-189); J in any layout
      const double* Jp = (const double*)ka->J + p * ka->J_stride;
      const double* rp = (const double*)ka->r + p * ka->r_stride;
      const long long rs = ka->J_row_major ? (long long)ka->J_ld : 1ll, cs = ka->J_row_major ? 1ll : (long long)ka->J_ld;
      const double lam_in = ka->lambda_vec ? ((const double*)ka->lambda_vec)[p * ka->lambda_vec_stride] : ka->lambda;
      const double lam = lam_in > 0.0 ? lam_in : 0.0;
      const int m_r = ka->m_r;
      for (int q0 = 0; q0 < m_r; q0 += 4) {  // four rows per round: twenty independent loads in flight, then the FMAs
        double jj[4], rq[4], ji[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool live = q0 + e < m_r;
          const double* row = Jp + (live ? q0 + e : 0) * rs;
          jj[e] = (live && j < n) ? row[j * cs] : 0.0;
          rq[e] = live ? rp[q0 + e] : 0.0;
#pragma unroll
          for (int t = 0; t < 4; ++t) ji[e][t] = (live && g + 4 * t < n) ? row[(g + 4 * t) * cs] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int t = 0; t < 4; ++t) K0[t] = fma(ji[e][t], jj[e], K0[t]);
          qv = fma(jj[e], rq[e], qv);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) K0[t] += (j == g + 4 * t && j < n) ? lam : 0.0;
    }
    if (k > 0) {
      const double* Ap = (const double*)ka->A + p * ka->A_stride;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = g + 4 * t;
        if (i < n && j >= n && j < P) K0[t] = Ap[(j - n) + (size_t)i * ka->A_ld];   // A^T block
        if (i >= n && i < P && j < n) K0[t] = Ap[(i - n) + (size_t)j * ka->A_ld];   // A block
      }
      if (j >= n && j < P) qv = ((const double*)ka->b + p * ka->b_stride)[j - n];
    }
    int cvar = 0; double ca = 1.0, cb = 0.0;
    if (lane < m) {
      cvar = ka->cons_var[p * ka->cons_stride + lane];
      ca = ((const double*)ka->cons_a)[p * ka->cons_stride + lane];
      cb = ((const double*)ka->cons_b)[p * ka->cons_stride + lane];
    }
    double* vp = (double*)ka->vars + p * ka->vars_stride;

    // ---- state: u = [x; -y] as V16 (value at lane j, replicated over g), s / z per constraint lane
    double u = 0.0, cs_ = 1.0, cz = 1.0;
    const bool residual_mode = mode == MODE_RESIDUAL;
    const bool step_mode = mode == MODE_STEP;
    const bool iterate_mode = mode == MODE_ITERATE || residual_mode || step_mode;  // one pass on the caller's state and mu
    if (iterate_mode || ka->sp.initial_guess_method == MO_GUESS_USER_PROVIDED) {  // qp.cc:440-442
      if (j < n) u = vp[j];
      if (j >= n && j < P) u = -vp[n + m + (j - n)];
      if (lane < m) { cs_ = vp[n + lane]; cz = vp[n + m + k + lane]; }
    }
    const bool bad_index = __any((lane < m) && ((cvar < 0) || (cvar >= n)));
    if (bad_index) cvar = 0;

    int st = bad_index ? MO_STATUS_BAD_INDEX : MO_STATUS_OK;
    int term = MO_MAX_ITERATIONS, it = 0;
    double mu = iterate_mode ? (ka->mu ? ((const double*)ka->mu)[p * ka->mu_stride] : 0.0) : ka->sp.initial_mu;
    bool guess_pass = !iterate_mode && ka->sp.initial_guess_method == MO_GUESS_SOLVE_EQUALITY_CONSTRAINED;
    double* iter_out = (ka->iterations && !iterate_mode) ? (double*)ka->iterations + (size_t)p * ka->sp.max_iterations * MO_ITER_RECORD : nullptr;

    // s = max(1e-9, a x + b), z = 1/s after clamping x into the feasible region in constraint order (qp.cc:464-481)
    auto clamp_and_init_slacks = [&]() {
      if (g == 0) xs[j] = (j < n) ? u : 0.0;
      lds_fence();
      for (int c = 0; c < m; ++c) {  // wave-uniform loop; one constraint at a time keeps the reference's order
        if (lane == c) {
          const double x0 = xs[cvar];
          double x1;
          if (ca < 0.0) { const double lim = cb / -ca; x1 = x0 < lim ? x0 : lim; }  // ClampX, qp.hpp:43-53
          else { const double lim = -cb / ca; x1 = x0 > lim ? x0 : lim; }
          xs[cvar] = x1;
        }
        lds_fence();
      }
      if (j < n) u = xs[j];
      double sz = 0.0;
      if (lane < m) {
        const double sv = ca * xs[cvar] + cb;
        cs_ = sv > 1.0e-9 ? sv : 1.0e-9;
        cz = 1.0 / cs_;
        sz = cs_ * cz;
      }
      if (ka->sp.initialize_mu_with_complementarity) mu = wave_sum_f64(sz) * inv_m;  // qp.cc:115
    };
    if (st == MO_STATUS_OK && !iterate_mode && ka->sp.initial_guess_method == MO_GUESS_NAIVE) clamp_and_init_slacks();
    if (!iterate_mode && ka->sp.initial_guess_method == MO_GUESS_USER_PROVIDED && ka->sp.initialize_mu_with_complementarity)
      mu = wave_sum_f64(lane < m ? cs_ * cz : 0.0) * inv_m;  // qp.cc:115 on the caller's state

    double n_rd2 = 0, n_rpe2 = 0, n_rc2 = 0, n_rc1 = 0, n_rpi2 = 0;
    // ComputeErrors (qp.cc:423-437) as SQUARED norms (the decisions compare squares; square roots only for the records)
    auto kkt_errors_sq = [&](double mu_e, double (&o)[4]) {
      o[0] = n_rd2;
      o[2] = k > 0 ? n_rpe2 : 0.0;
      if (m > 0) {
        const double corrected = n_rc2 - 2 * (n_rc1 * mu_e) + (mu_e * mu_e) * (double)m;
        o[1] = corrected > 0.0 ? corrected : 0.0;
        o[3] = n_rpi2;
      } else { o[1] = 0.0; o[3] = 0.0; }
    };
    double mu_used = mu;
    double ip_alpha_p = 1.0, ip_alpha_d = 1.0;
    const bool use_pc = (step_mode ? MO_COMPLEMENTARITY : (iterate_mode ? ka->barrier_strategy : ka->sp.barrier_strategy)) == MO_PREDICTOR_CORRECTOR && m > 0;
    const double nanv = __builtin_nan("");
    double ip_mu = mu, probe_p = nanv, probe_d = nanv, mu_aff = nanv, mu_pc = 0.0;
    double dsv = 0.0, dzv = 0.0, ap = 1.0, ad = 1.0, sol = 0.0;  // sol = [dx; -dy] of the last solve

    while (st == MO_STATUS_OK) {
      const bool include_ineq = !guess_pass && !((residual_mode || step_mode) && (ka->flags & MO_STEP_NO_INEQUALITIES));
      const int lane = lane_id(), g = lane >> 4, j = lane & 15;  // re-made opaque every pass
      ka = fresh_args();
      // ---------------------------------------------------------------- part A: residual and norms
      if (g == 0) xs[j] = (j < n) ? u : 0.0;
      if (lane < 16) { azS[lane] = 0.0; diagS[lane] = 0.0; rhoS[lane] = 0.0; }
      lds_fence();
      double r_pi = 0.0, r_comp = 0.0;
      if (include_ineq && lane < m) {
        atomicAdd(&azS[cvar], ca * cz);               // qp.cc:415
        r_pi = ca * xs[cvar] + cb - cs_;              // qp.cc:416
        r_comp = cs_ * cz;                            // qp.cc:417
      }
      // r = K0 [x; -y] + [c; b] - [sum a z; 0]  (qp.cc:404-408, 415): r_d in lanes j < n, r_pe in lanes n <= j < P
      const double w = tile_times_vector(K0, u, P, g, j, hop);
      const double r16 = (j < P) ? w + qv - azS[j] : 0.0;
      {
        const double sq = r16 * r16;
        n_rd2 = readlane_f64(row_sum(j < n ? sq : 0.0), 0);
        n_rpe2 = readlane_f64(row_sum(j >= n ? sq : 0.0), 0);
        if (m > 0) {  // wave-uniform
          n_rc2 = wave_sum_f64(r_comp * r_comp);
          n_rc1 = wave_sum_f64(r_comp);
          n_rpi2 = wave_sum_f64(r_pi * r_pi);
        }
      }
      if (residual_mode) {  // r_ = [r_d | r_comp | r_pe | r_pi] (qp.cc:391-420) and the four norms of ComputeErrors (qp.cc:423-437)
        double* ro = (double*)ka->r_out + p * ka->r_out_stride;
        if (g == 0) {
          if (j < n) ro[j] = r16;
          if (j >= n && j < P) ro[n + m + (j - n)] = r16;
        }
        if (lane < m) { ro[n + lane] = r_comp; ro[n + m + k + lane] = r_pi; }
        if (ka->kkt_out) {
          double kq[4];
          kkt_errors_sq(mu, kq);
          if (!include_ineq) { kq[1] = 0.0; kq[3] = 0.0; }
          const double e0 = sqrt(kq[0]), e1 = sqrt(kq[1]), e2 = sqrt(kq[2]), e3 = sqrt(kq[3]);
          if (lane == 0) { double* ko = (double*)ka->kkt_out + 4 * p; ko[0] = e0; ko[1] = e1; ko[2] = e2; ko[3] = e3; }
        }
        break;
      }
      if (!guess_pass && !iterate_mode) {
        // ---- the decision point of Solve (qp.cc:116-147)
        if (it > 0) {
          double kf[4];
          kkt_errors_sq(mu_used, kf);                               // kkt_after of the previous iteration (squared), qp.cc:127
          const double cur_mu = n_rc1 * inv_m;                      // ComputeMu, qp.cc:509-516
          if (iter_out) {
            const double r4 = sqrt(kf[0]), r5 = sqrt(kf[1]), r6 = sqrt(kf[2]), r7 = sqrt(kf[3]);
            if (lane == 0) {
              double* rec = iter_out + (size_t)(it - 1) * MO_ITER_RECORD;
              rec[4] = r4; rec[5] = r5; rec[6] = r6; rec[7] = r7;
              rec[8] = ip_mu; rec[9] = ip_alpha_p; rec[10] = ip_alpha_d;
              rec[11] = probe_p; rec[12] = probe_d; rec[13] = mu_aff;
            }
          }
          double kmax2 = kf[0];                                     // KKTError::Max() squared
          kmax2 = kf[1] > kmax2 ? kf[1] : kmax2; kmax2 = kf[2] > kmax2 ? kf[2] : kmax2; kmax2 = kf[3] > kmax2 ? kf[3] : kmax2;
          if (kmax2 < ka->sp.termination_kkt_tol * ka->sp.termination_kkt_tol && cur_mu < ka->sp.termination_complementarity_tol) {  // qp.cc:132-137
            term = MO_SATISFIED_KKT_TOL;
            break;
          }
          if (kmax2 <= mu * mu || !ka->sp.decrease_mu_only_on_small_error) {                       // qp.cc:140-146 (mu > 0)
            if (ka->sp.barrier_strategy == MO_FIXED_DECREASE) mu *= ka->sp.sigma;
            else mu = ka->sp.sigma * cur_mu;
          }
        }
        if (it >= ka->sp.max_iterations) break;                          // MAX_ITERATIONS, qp.cc:149
        if (iter_out) {                                              // kkt_prev is only ever recorded, qp.cc:118
          double ki[4];
          kkt_errors_sq(mu, ki);
          const double r0 = sqrt(ki[0]), r1 = sqrt(ki[1]), r2 = sqrt(ki[2]), r3 = sqrt(ki[3]);
          if (lane == 0) {
            double* rec = iter_out + (size_t)it * MO_ITER_RECORD;
            rec[0] = r0; rec[1] = r1; rec[2] = r2; rec[3] = r3;
          }
        }
      }
      // ---------------------------------------------------------------- part B: right-hand side, factorisation, direction
      const bool predictor_pass = use_pc && !guess_pass;  // Mehrotra: solve with mu = 0, probe, then the corrector through the same factors
      const double mu_step = (m > 0 && include_ineq) ? (predictor_pass ? 0.0 : mu) : 0.0;  // qp.cc:165-187
      double aff = 0.0, cs_inv = 1.0;  // aff = ds_aff dz_aff (qp.cc:341), set by the predictor
      if (include_ineq) {
        if (__any((lane < m) && !(cs_ > 0.0))) { st = MO_STATUS_NONPOSITIVE_SLACK; break; }  // qp.cc:285
        cs_inv = rcp_f64(cs_);
        if (lane < m) {
          const double zs = cz * cs_inv;
          atomicAdd(&diagS[cvar], ca * zs * ca);                                            // qp.cc:296
          atomicAdd(&rhoS[cvar], ca * zs * r_pi + ca * (r_comp + aff - mu_step) * cs_inv);  // qp.cc:340-341
        }
      }
      lds_fence();
      if (g == 0) tmp[j] = (j < P) ? -(r16 + ((j < n) ? rhoS[j] : 0.0)) : 0.0;  // -[r_aug; r_pe] by row (qp.cc:337-342)
      lds_fence();
      d4 T = K0;
      {
        const double dd = (j < n) ? diagS[j] : 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          T[t] += (j == g + 4 * t) ? dd : 0.0;
          if (j == kRC) T[t] = tmp[g + 4 * t];   // the right-hand side rides in column 15 (P <= 15)
        }
      }
      lds_fence();
      __builtin_amdgcn_sched_barrier(0);
      if (!sweep_tile<3>(T, P, g, j)) { st = MO_STATUS_FACTORIZATION_FAILED; break; }
      {  // the solution sits in column 15 of the swept tile: element (q, 15) at lane (q & 3, 15), register q >> 2
        double v = 0.0;
        const int src = (16 * (j & 3) + kRC) * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double wv = bpermute_f64(src, T[t]);
          if ((j >> 2) == t) v = wv;
        }
        sol = (j < P) ? v : 0.0;
      }
      // From a solution to the direction: dx (by variable in LDS), ds, dz, the step lengths (qp.cc:359-363, 485-507)
      auto finish_direction = [&](double mu_s, double tau) -> bool {
        bool finite = fabs(sol) < INFINITY;
        if (g == 0) tmp[j] = sol;
        lds_fence();
        ap = 1.0; ad = 1.0; dsv = 0.0; dzv = 0.0;
        if (include_ineq && lane < m) {
          dsv = ca * tmp[cvar] + r_pi;                                              // qp.cc:361
          dzv = -(cz * cs_inv) * dsv - cs_inv * (r_comp + aff - mu_s);              // qp.cc:362
          if (cs_ + dsv <= 0.0 && fabs(dsv) > 0.0) ap = -tau * cs_ * rcp_f64(dsv);  // qp.cc:498-503
          if (cz + dzv <= 0.0 && fabs(dzv) > 0.0) ad = -tau * cz * rcp_f64(dzv);
          finite = finite && (fabs(dsv) < INFINITY) && (fabs(dzv) < INFINITY);
        }
        if (!__all(finite)) return false;
        if (m > 0) {  // wave-uniform
          ap = cross_row_min(row_min(ap));
          ad = cross_row_min(row_min(ad));
        }
        lds_fence();
        return true;
      };
      if (guess_pass) {                      // qp.cc:455-460: x, y <- the equality-constrained solution
        if (!__all(fabs(sol) < INFINITY)) { st = MO_STATUS_NONFINITE; break; }
        u = sol;
        guess_pass = false;
        clamp_and_init_slacks();
        continue;
      }
      const double tau_final = step_mode ? ka->tau : 0.995;                                                    // qp.cc:192
      if (!finish_direction(mu_step, predictor_pass ? 1.0 : tau_final)) { st = MO_STATUS_NONFINITE; break; }  // tau = 1: qp.cc:174
      ip_mu = mu;
      if (predictor_pass) {
        probe_p = ap; probe_d = ad;                                                    // alpha_probe, qp.cc:174
        const double sdz = wave_sum_f64(lane < m ? cs_ * dzv : 0.0), zds = wave_sum_f64(lane < m ? cz * dsv : 0.0),
                     dsdz = wave_sum_f64(lane < m ? dsv * dzv : 0.0);
        aff = dsv * dzv;                                                               // delta_affine_ (qp.cc:177)
        double ma = mu;                                                                // qp.cc:519-537
        ma += ad * sdz * inv_m;
        ma += ap * zds * inv_m;
        ma += (ad * ap) * dsdz * inv_m;
        mu_aff = ma > 0.0 ? ma : 0.0;
        const double ratio = mu_aff * rcp_f64(mu);
        mu_pc = (ratio * ratio * ratio) * mu;                                          // qp.cc:182-183
        // The corrector solve (qp.cc:187): same matrix, new right-hand side -- the swept block is -K^-1
        if (lane < 16) rhoS[lane] = 0.0;
        lds_fence();
        if (lane < m) {
          const double zs = cz * cs_inv;
          atomicAdd(&rhoS[cvar], ca * zs * r_pi + ca * (r_comp + aff - mu_pc) * cs_inv);  // qp.cc:340-341
        }
        lds_fence();
        const double rhs2 = (j < P) ? -(r16 + ((j < n) ? rhoS[j] : 0.0)) : 0.0;
        const double s2 = -tile_times_vector(T, rhs2, P, g, j, hop);
        sol = (j < P) ? s2 : 0.0;
        if (!finish_direction(mu_pc, 0.995)) { st = MO_STATUS_NONFINITE; break; }
        ip_mu = mu_pc;
      }
      if (step_mode) break;  // the bare step: direction and step lengths only
      // x,s += alpha_p (dx,ds) ; y,z += alpha_d (dy,dz), qp.cc:196-199   (u holds -y, sol holds -dy)
      u = fma(sol, (j < n) ? ap : ad, u);
      cs_ = fma(dsv, ap, cs_); cz = fma(dzv, ad, cz);
      mu_used = mu; ip_alpha_p = ap; ip_alpha_d = ad;
      ++it;
      if (iterate_mode) break;
    }

    // ---- outputs
    ka = fresh_args();
    if (step_mode || mode == MODE_ITERATE) {  // delta_ = [dx | ds | dy | dz] (NaN on a failed problem of the step), step lengths
      const bool okp = st == MO_STATUS_OK;
      if (ka->delta && (okp || step_mode)) {
        double* dp = (double*)ka->delta + p * ka->delta_stride;
        if (g == 0) {
          if (j < n) dp[j] = okp ? sol : nanv;
          if (j >= n && j < P) dp[n + m + (j - n)] = okp ? -sol : nanv;
        }
        if (lane < m) { dp[n + lane] = okp ? dsv : nanv; dp[n + m + k + lane] = okp ? dzv : nanv; }
      }
      if (lane == 0) {
        if (step_mode && ka->alpha) {
          ((double*)ka->alpha)[2 * p] = okp ? ap : nanv;
          ((double*)ka->alpha)[2 * p + 1] = okp ? ad : nanv;
        }
        if (mode == MODE_ITERATE && ka->ip_out && okp) {
          double* ip = (double*)ka->ip_out + p * MO_IP_RECORD;
          ip[0] = ip_mu; ip[1] = ap; ip[2] = ad;  // outputs.mu = mu_input (sigma mu_input after a corrector), qp.cc:160, 183
          ip[3] = probe_p; ip[4] = probe_d; ip[5] = mu_aff;
        }
      }
    }
    if (!residual_mode && !step_mode) {  // the state is an input only there
      if (g == 0) {
        if (j < n) vp[j] = u;
        if (j >= n && j < P) vp[n + m + (j - n)] = -u;
      }
      if (lane < m) { vp[n + lane] = cs_; vp[n + m + k + lane] = cz; }
    }
    const double yv = (j >= n && j < P) ? -u : 0.0;
    const double ymin = row_min((j >= n && j < P) ? yv : INFINITY), yabs = -row_min((j >= n && j < P) ? -fabs(yv) : INFINITY);
    if (lane == 0) {
      if (ka->termination) ka->termination[p] = term;
      if (ka->num_iterations) ka->num_iterations[p] = it;
      if (ka->status) ka->status[p] = st;
      if (ka->lagrange) {  // qp.cc:539-546
        ((double*)ka->lagrange)[2 * p] = k > 0 ? ymin : nanv;
        ((double*)ka->lagrange)[2 * p + 1] = k > 0 ? yabs : nanv;
      }
    }
    lds_fence();
  }  // problems of this ticket
    base = uniform64(next_ticket) + ticket_base;
    chunk = next_chunk;
  }
}

}  // namespace

bool fused_tiny_supported(const KernelArgs& a) {  // the caller has run fused_supported() on the same arguments
  if (a.mode == MODE_LINEARIZE || a.no_tiny) return false;
  if (a.n + a.k > 15 || a.m > 64) return false;
  if (a.J && a.m_r > 64) return false;  // larger stacks: the 32-variable grid streams J through the matrix cores
  return true;
}

hipError_t launch_fused_tiny(const KernelArgs& a, int num_cus, hipStream_t stream) {  // the work counter has been zeroed by launch_fused
  constexpr int WPS = 3;
  long long grid = num_cus;        // one workgroup of 12 waves per CU (145 VGPRs: three waves per SIMD; four would spill)
  const long long need = (a.batch + 4 * WPS - 1) / (4 * WPS);
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((kkt_tiny_kernel<WPS>), dim3((unsigned)grid), dim3(256 * WPS), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mo
