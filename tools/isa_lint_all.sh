#!/bin/bash
# Compiles every kernel translation unit of the product to gfx950 assembly with the Makefile's flags and runs tools/isa_lint.py over it
# (the code-generation defect of DESIGN.md section 4.3).  Output: one summary line per file; exit code 1 if anything is flagged.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/mini_opt_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-function-calls=false"
OUT=${ISA_LINT_DIR:-/tmp/isa_lint}
mkdir -p $OUT
rc=0
for f in kkt_generic kkt_fused kkt_fused_gather kkt_fused_ny2 kkt_fused_ny34 kkt_fused_mc4 kkt_fused_tiny kkt_fused_f32 nls_kernels; do
  ( /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o $OUT/$f.s $SRC/$f.hip 2>/dev/null ) &
done
wait
for f in kkt_generic kkt_fused kkt_fused_gather kkt_fused_ny2 kkt_fused_ny34 kkt_fused_mc4 kkt_fused_tiny kkt_fused_f32 nls_kernels; do
  python3 $ROOT/tools/isa_lint.py $OUT/$f.s | tail -1 | sed "s#^#$f.hip: #"
  python3 $ROOT/tools/isa_lint.py $OUT/$f.s > /dev/null || rc=1
  grep -c "s_swappc_b64" $OUT/$f.s | sed "s#^#$f.hip: device function calls (s_swappc_b64): #"
done
exit $rc
