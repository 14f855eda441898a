// kkt_fused_ny2.hip -- the fused fp64 kernels (kkt_fused.hip) instantiated with TWO y tiles (NY = 2): 16 <= k <= 31 equality constraints.
// The first y tile is a full 16-pivot tile of the Schur complement, the second holds the remaining k - 16 rows and, as in the one-tile
// kernels, carries the right-hand side in index 15.  A translation unit of its own so that the instantiations compile beside the others.
// Waves per SIMD follow the register budget of the (NT + 2)(NT + 3) / 2 live tiles: 3 on the 32 grid, 2 on the 64 grid, 1 beyond.
#define MO_FUSED_IMPL_ONLY
#include "kkt_fused.hip"

namespace mo {

hipError_t launch_fused_ny2(const KernelArgs& a, int num_cus, hipStream_t stream) {  // the work counter has been zeroed by launch_fused
  const bool solve = a.mode == MODE_SOLVE || a.mode == MODE_ITERATE || a.mode == MODE_RESIDUAL;
  const int grid_tile = a.n > 96 ? 8 : a.n > 64 ? 6 : a.n > 32 ? 4 : 2;
  // Solve on the 64 grid: 21 live tiles + the loop state do not fit the 256 registers of two waves per SIMD (408 B of spill per lane, which
  // showed up as 30 GB of HBM traffic per launch in profiles/r03_solve_k24_*).  A/B knob MO_NY2_SOLVE_WPS=1: one wave per SIMD, no spill.
#ifdef MO_TUNING
  static const int env_solve_wps = [] { const char* e = getenv("MO_NY2_SOLVE_WPS"); return e ? atoi(e) : 0; }();
#else
  constexpr int env_solve_wps = 0;
#endif
  const bool solve64_one_wave = solve && grid_tile == 4 && env_solve_wps == 1;
  const int wps = grid_tile == 2 ? 3 : grid_tile == 4 ? (solve64_one_wave ? 1 : 2) : 1;
  const bool one_slot = a.m <= 64;   // one constraint slot per lane is enough: fewer live registers
  // J-level input: 16-byte pieces of a packed row-major J, or the per-lane gather stream for every other layout (odd n included)
  const bool gather = a.J && (fused_needs_gather(a) || (a.n & 1));
  long long grid = num_cus;
  const long long need = (a.batch + 3) / 4;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  const dim3 gd((unsigned)grid), bd(256 * wps);
#define MO_NY2(KERNEL, NT_, WPS_, MC_)                                                                                       \
  do {                                                                                                                       \
    if (!a.J) hipLaunchKernelGGL((KERNEL<NT_, WPS_, 3, true, MC_, JMODE_VECTOR, 2>), gd, bd, 0, stream, a);                  \
    else if (gather) hipLaunchKernelGGL((KERNEL<NT_, WPS_, 3, false, 1, JMODE_GATHER, 2>), gd, bd, 0, stream, a);            \
    else hipLaunchKernelGGL((KERNEL<NT_, WPS_, 3, false, MC_, JMODE_VECTOR, 2>), gd, bd, 0, stream, a);                      \
  } while (0)
  // The corrector's second solve keeps all 21 factor tiles of the 64 grid alive behind the back-substitution: with it the kernel needs more
  // than the 256 registers of two waves per SIMD (84 B of scratch per lane), without it none.  Hence two instantiations on the 32 / 64 grids,
  // picked by barrier strategy: PCK = false for COMPLEMENTARITY / FIXED_DECREASE (and the KKT residual), PCK = true for PREDICTOR_CORRECTOR.
  const bool pc = (a.mode == MODE_SOLVE ? a.sp.barrier_strategy : a.barrier_strategy) == MO_PREDICTOR_CORRECTOR && a.mode != MODE_RESIDUAL;
#define MO_NY2S(NT_, WPS_, MC_, PCK_)                                                                                                    \
  do {                                                                                                                                   \
    if (!a.J) hipLaunchKernelGGL((kkt_fused_solve_kernel<NT_, WPS_, 3, true, MC_, JMODE_VECTOR, 2, PCK_>), gd, bd, 0, stream, a);        \
    else if (gather) hipLaunchKernelGGL((kkt_fused_solve_kernel<NT_, WPS_, 3, false, 1, JMODE_GATHER, 2, PCK_>), gd, bd, 0, stream, a);  \
    else hipLaunchKernelGGL((kkt_fused_solve_kernel<NT_, WPS_, 3, false, MC_, JMODE_VECTOR, 2, PCK_>), gd, bd, 0, stream, a);            \
  } while (0)
  const bool four_slots = a.m > 128;   // (packed J or (G, c) only: fused_supported keeps the gather stream at m <= 64)
  if (solve && four_slots) {
    switch (grid_tile) {
      case 2: MO_NY2(kkt_fused_solve_kernel, 2, 3, 4); break;
      case 4: MO_NY2(kkt_fused_solve_kernel, 4, 2, 4); break;
      case 6: MO_NY2(kkt_fused_solve_kernel, 6, 1, 4); break;
      default: MO_NY2(kkt_fused_solve_kernel, 8, 1, 4); break;
    }
  } else if (!solve && four_slots) {
    switch (grid_tile) {
      case 2: MO_NY2(kkt_fused_f64_kernel, 2, 3, 4); break;
      case 4: MO_NY2(kkt_fused_f64_kernel, 4, 2, 4); break;
      case 6: MO_NY2(kkt_fused_f64_kernel, 6, 1, 4); break;
      default: MO_NY2(kkt_fused_f64_kernel, 8, 1, 4); break;
    }
  } else if (solve) {
    switch (grid_tile) {
      case 2: if (pc) MO_NY2S(2, 3, 2, true); else MO_NY2S(2, 3, 2, false); break;
      case 4:
#ifdef MO_TUNING
        if (solve64_one_wave) { if (one_slot) MO_NY2(kkt_fused_solve_kernel, 4, 1, 1); else MO_NY2(kkt_fused_solve_kernel, 4, 1, 2); }
        else
#endif
        if (one_slot) { if (pc) MO_NY2S(4, 2, 1, true); else MO_NY2S(4, 2, 1, false); }
        else { if (pc) MO_NY2S(4, 2, 2, true); else MO_NY2S(4, 2, 2, false); }
        break;
      case 6: if (one_slot) MO_NY2(kkt_fused_solve_kernel, 6, 1, 1); else MO_NY2(kkt_fused_solve_kernel, 6, 1, 2); break;
      default: if (one_slot) MO_NY2(kkt_fused_solve_kernel, 8, 1, 1); else MO_NY2(kkt_fused_solve_kernel, 8, 1, 2); break;
    }
  } else {
    switch (grid_tile) {
      case 2: MO_NY2(kkt_fused_f64_kernel, 2, 3, 2); break;
      case 4: MO_NY2(kkt_fused_f64_kernel, 4, 2, 2); break;
      case 6: MO_NY2(kkt_fused_f64_kernel, 6, 1, 2); break;
      default: MO_NY2(kkt_fused_f64_kernel, 8, 1, 2); break;
    }
  }
#undef MO_NY2S
#undef MO_NY2
  return hipGetLastError();
}

}  // namespace mo
