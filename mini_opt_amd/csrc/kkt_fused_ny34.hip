// kkt_fused_ny34.hip -- the fused fp64 kernels (kkt_fused.hip) instantiated with THREE and FOUR y tiles: 32 <= k <= 47 and 48 <= k <= 63 equality
// constraints on the 32- and 64-variable tile grids (the reference puts no limit on the number of equality rows, qp.cc:36-48; beyond these
// shapes the generic kernel takes over).  Every y tile but the last is a full 16-pivot tile of the Schur complement; the last holds the
// remaining rows and the right-hand side in index 15.  (NT + NY)(NT + NY + 1) / 2 live tiles: 15 / 21 on the 32 grid (three / two waves per
// SIMD), 28 / 36 on the 64 grid (one wave per SIMD), 45 / 55 on the 96 grid, 66 on the 128 grid with three (round 4: k <= 63 up to n = 96, k <= 47 up to n = 128).  Packed even-n J or (G, c) input with m <= 256 (two constraint slots per lane, four beyond 128: round 4); any other layout of J through the gather
// stream with m <= 64 (round 4).
#define MO_FUSED_IMPL_ONLY
#include "kkt_fused.hip"

namespace mo {

hipError_t launch_fused_ny34(const KernelArgs& a, int num_cus, hipStream_t stream) {  // the work counter has been zeroed by launch_fused
  const bool solve = a.mode == MODE_SOLVE || a.mode == MODE_ITERATE || a.mode == MODE_RESIDUAL;
  const bool big = a.n > 32, huge = a.n > 64, four = a.k > 47;   // (huge: the 96 grid, three y tiles only -- fused_supported)
  const int wps = big ? 1 : (four ? 2 : (solve ? 2 : 3));
  long long grid = num_cus;
  const long long need = (a.batch + 3) / 4;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  const dim3 gd((unsigned)grid), bd(256 * wps);
  // J-level input: 16-byte pieces of a packed row-major J, or the per-lane gather stream for every other layout (odd n included; m <= 64)
  const bool gather = a.J && (fused_needs_gather(a) || (a.n & 1));
  const bool four_slots = a.m > 128;   // (packed J or (G, c): fused_supported keeps the gather stream at m <= 64)
#define MO_NY34(KERNEL, NT_, WPS_, NY_)                                                                          \
  do {                                                                                                           \
    if (four_slots) {                                                                                            \
      if (!a.J) hipLaunchKernelGGL((KERNEL<NT_, WPS_, 3, true, 4, JMODE_VECTOR, NY_>), gd, bd, 0, stream, a);    \
      else hipLaunchKernelGGL((KERNEL<NT_, WPS_, 3, false, 4, JMODE_VECTOR, NY_>), gd, bd, 0, stream, a);        \
    } else if (!a.J) hipLaunchKernelGGL((KERNEL<NT_, WPS_, 3, true, 2, JMODE_VECTOR, NY_>), gd, bd, 0, stream, a); \
    else if (gather) hipLaunchKernelGGL((KERNEL<NT_, WPS_, 3, false, 1, JMODE_GATHER, NY_>), gd, bd, 0, stream, a); \
    else hipLaunchKernelGGL((KERNEL<NT_, WPS_, 3, false, 2, JMODE_VECTOR, NY_>), gd, bd, 0, stream, a);          \
  } while (0)
  if (a.n > 96) {          // the 128 grid: three y tiles only (fused_supported)
    if (solve) MO_NY34(kkt_fused_solve_kernel, 8, 1, 3); else MO_NY34(kkt_fused_f64_kernel, 8, 1, 3);
  } else if (huge) {       // the 96 grid
    if (solve) { if (four) MO_NY34(kkt_fused_solve_kernel, 6, 1, 4); else MO_NY34(kkt_fused_solve_kernel, 6, 1, 3); }
    else { if (four) MO_NY34(kkt_fused_f64_kernel, 6, 1, 4); else MO_NY34(kkt_fused_f64_kernel, 6, 1, 3); }
  } else if (solve) {
    if (big) { if (four) MO_NY34(kkt_fused_solve_kernel, 4, 1, 4); else MO_NY34(kkt_fused_solve_kernel, 4, 1, 3); }
    else { if (four) MO_NY34(kkt_fused_solve_kernel, 2, 2, 4); else MO_NY34(kkt_fused_solve_kernel, 2, 2, 3); }
  } else {
    if (big) { if (four) MO_NY34(kkt_fused_f64_kernel, 4, 1, 4); else MO_NY34(kkt_fused_f64_kernel, 4, 1, 3); }
    else { if (four) MO_NY34(kkt_fused_f64_kernel, 2, 2, 4); else MO_NY34(kkt_fused_f64_kernel, 2, 3, 3); }
  }
#undef MO_NY34
  return hipGetLastError();
}

}  // namespace mo
