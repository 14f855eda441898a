#!/bin/bash
# Print VGPR / scratch use of every gfx950 kernel in a HIP source (default: the fused kernels).
# usage: tools/kernel_regs.sh [file.hip]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${1:-$ROOT/mini_opt_amd/csrc/kkt_fused.hip}
OUT=$(mktemp /tmp/kregs.XXXXXX.s)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-function-calls=false \
  -S --cuda-device-only -o "$OUT" "$SRC" 2>/dev/null
awk '/\.amdhsa_kernel /{name=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_private_segment_fixed_size/{s=$2} /\.end_amdhsa_kernel/{print v, s, name}' "$OUT" \
  | while read v s name; do printf "vgpr %4d  scratch %5d B  %s\n" "$v" "$s" "$(echo "$name" | c++filt | sed 's/mo::(anonymous namespace):://; s/(mo::KernelArgs)//; s/^void //')"; done
echo "asm: $OUT"
