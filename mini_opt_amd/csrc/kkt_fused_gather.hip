// kkt_fused_gather.hip -- the fused fp64 kernels (kkt_fused.hip) instantiated with the per-lane gather stream (JMODE_GATHER): J in any
// layout the C ABI accepts -- column-major, a leading dimension beyond n, rows that are only 8-byte aligned, odd n up to 128.  A
// translation unit of its own so that the instantiations compile beside the fast-path ones.
#define MO_FUSED_IMPL_ONLY
#include "kkt_fused.hip"

namespace mo {

hipError_t launch_fused_gather(const KernelArgs& a, int num_cus, hipStream_t stream) {  // the work counter has been zeroed by launch_fused
  const bool solve = a.mode == MODE_SOLVE || a.mode == MODE_ITERATE || a.mode == MODE_RESIDUAL;
  const int grid_tile = a.n > 96 ? 8 : a.n > 64 ? 6 : a.n > 32 ? 4 : 2;
  // waves per SIMD: the register budgets of the fast-path instantiations
  // (the 32 grid's step kernel runs four waves per SIMD since round 4, as its packed sibling does: 105 VGPRs)
  const bool step32 = !solve && a.mode != MODE_LINEARIZE && grid_tile == 2;
  const int wps = solve ? (grid_tile == 2 ? 3 : grid_tile == 4 ? 2 : 1) : (step32 ? 4 : grid_tile == 2 || grid_tile == 4 ? 3 : grid_tile == 6 ? 2 : 1);
  long long grid = num_cus;
  const long long need = (a.batch + 3) / 4;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  const dim3 gd((unsigned)grid), bd(256 * wps);
  if (a.mode == MODE_LINEARIZE) {  // mo_linearize / mo_fill_qp with column-major, strided, unaligned J or odd n
    switch (grid_tile) {
      case 2: hipLaunchKernelGGL((kkt_fused_linearize_kernel<2, 3, JMODE_GATHER>), gd, bd, 0, stream, a); break;
      case 4: hipLaunchKernelGGL((kkt_fused_linearize_kernel<4, 3, JMODE_GATHER>), gd, bd, 0, stream, a); break;
      case 6: hipLaunchKernelGGL((kkt_fused_linearize_kernel<6, 2, JMODE_GATHER>), gd, bd, 0, stream, a); break;
      default: hipLaunchKernelGGL((kkt_fused_linearize_kernel<8, 1, JMODE_GATHER>), gd, bd, 0, stream, a); break;
    }
    return hipGetLastError();
  }
  if (solve) {
    switch (grid_tile) {
      case 2: hipLaunchKernelGGL((kkt_fused_solve_kernel<2, 3, 3, false, 1, JMODE_GATHER>), gd, bd, 0, stream, a); break;
      case 4: hipLaunchKernelGGL((kkt_fused_solve_kernel<4, 2, 3, false, 1, JMODE_GATHER>), gd, bd, 0, stream, a); break;
      case 6: hipLaunchKernelGGL((kkt_fused_solve_kernel<6, 1, 3, false, 1, JMODE_GATHER>), gd, bd, 0, stream, a); break;
      default: hipLaunchKernelGGL((kkt_fused_solve_kernel<8, 1, 3, false, 1, JMODE_GATHER>), gd, bd, 0, stream, a); break;
    }
  } else {
    switch (grid_tile) {
      case 2: hipLaunchKernelGGL((kkt_fused_f64_kernel<2, 4, 3, false, 1, JMODE_GATHER>), gd, bd, 0, stream, a); break;
      case 4: hipLaunchKernelGGL((kkt_fused_f64_kernel<4, 3, 3, false, 1, JMODE_GATHER>), gd, bd, 0, stream, a); break;
      case 6: hipLaunchKernelGGL((kkt_fused_f64_kernel<6, 2, 3, false, 1, JMODE_GATHER>), gd, bd, 0, stream, a); break;
      default: hipLaunchKernelGGL((kkt_fused_f64_kernel<8, 1, 3, false, 1, JMODE_GATHER>), gd, bd, 0, stream, a); break;
    }
  }
  return hipGetLastError();
}

}  // namespace mo
