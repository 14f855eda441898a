// kkt_fused_mc4.hip -- the fused fp64 kernels (kkt_fused.hip) with FOUR constraint slots per lane: 128 < m <= 256 inequality entries, e.g. a
// two-sided box on every one of 128 variables -- and Solve / Iterate / the KKT residual on the 96 / 128 tile grids with m > 64 (two or four
// slots), which the one-slot instantiations of kkt_fused.hip do not take.  k <= 15, packed (J, r, lambda) or (G, c).  The constraint state
// (s, z, a, b, variable, residual parts) lives in registers, so the larger grids spill: correctness-first instantiations, one or two waves per
// SIMD.  A translation unit of its own so that the instantiations compile beside the others.
#define MO_FUSED_IMPL_ONLY
#include "kkt_fused.hip"

namespace mo {

hipError_t launch_fused_mc4(const KernelArgs& a, int num_cus, hipStream_t stream) {  // the work counter has been zeroed by launch_fused
  const bool solve = a.mode == MODE_SOLVE || a.mode == MODE_ITERATE || a.mode == MODE_RESIDUAL;
  const int grid_tile = a.n > 96 ? 8 : a.n > 64 ? 6 : a.n > 32 ? 4 : 2;
  const int wps = grid_tile == 2 ? 3 : grid_tile == 4 ? 2 : 1;
  long long grid = num_cus;
  const long long need = (a.batch + 3) / 4;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  const dim3 gd((unsigned)grid), bd(256 * wps);
#define MO_MC(KERNEL, NT_, WPS_, MC_)                                                                  \
  do {                                                                                                 \
    if (a.J) hipLaunchKernelGGL((KERNEL<NT_, WPS_, 3, false, MC_>), gd, bd, 0, stream, a);             \
    else hipLaunchKernelGGL((KERNEL<NT_, WPS_, 3, true, MC_>), gd, bd, 0, stream, a);                  \
  } while (0)
  if (solve) {
    switch (grid_tile) {
      case 2: MO_MC(kkt_fused_solve_kernel, 2, 3, 4); break;
      case 4: MO_MC(kkt_fused_solve_kernel, 4, 2, 4); break;
      case 6: if (a.m > 128) MO_MC(kkt_fused_solve_kernel, 6, 1, 4); else MO_MC(kkt_fused_solve_kernel, 6, 1, 2); break;
      default: if (a.m > 128) MO_MC(kkt_fused_solve_kernel, 8, 1, 4); else MO_MC(kkt_fused_solve_kernel, 8, 1, 2); break;
    }
  } else {
    switch (grid_tile) {
      case 2: MO_MC(kkt_fused_f64_kernel, 2, 3, 4); break;
      case 4: MO_MC(kkt_fused_f64_kernel, 4, 2, 4); break;
      case 6: MO_MC(kkt_fused_f64_kernel, 6, 1, 4); break;
      default: MO_MC(kkt_fused_f64_kernel, 8, 1, 4); break;
    }
  }
#undef MO_MC
  return hipGetLastError();
}

}  // namespace mo
