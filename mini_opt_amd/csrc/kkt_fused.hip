// kkt_fused.hip -- fixed-shape fused Newton-step kernels (placeholder: no shapes registered yet).
#include "mo_kernels.h"

namespace mo {
bool fused_supported(const KernelArgs&, int) { return false; }
const char* fused_name(const KernelArgs&, int) { return "none"; }
hipError_t launch_fused(const KernelArgs&, int, int, hipStream_t) { return hipErrorNotSupported; }
}  // namespace mo
