#!/bin/bash
# Diagnostic builds of the library for the "alpha_dual = 1 from generic Solve" finding:
#   inline   every stage inlined (the product's code generation) + probes
#   call     assemble_and_factor forced to be a real function + probes
#   auto     the inliner's own choice, no -amdgpu-function-calls=false (the build that showed the wrong result) + probes
#   byval    call + the workspace descriptor passed by value
#   fence    call + fence / waitcnt(0) / barrier after the call returns
# Only kkt_generic.hip is recompiled; the other objects come from the product build (make -C mini_opt_amd/csrc).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/mini_opt_amd/csrc
OUT=$ROOT/tools/alpha_dual_probe/lib
mkdir -p "$OUT"
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -mllvm -amdgpu-atomic-optimizer-strategy=None"
OTHERS=$(ls $SRC/*.o | grep -v kkt_generic.o)
build() {  # name, extra flags
  local name=$1; shift
  /opt/rocm/bin/hipcc $BASE "$@" -DMO_GENERIC_PROBE -c $SRC/kkt_generic.hip -o $OUT/kkt_generic_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libminiopt_$name.so $OTHERS $OUT/kkt_generic_$name.o
  echo built $OUT/libminiopt_$name.so
}
build inline -mllvm -amdgpu-function-calls=false &
build call -DMO_GENERIC_PROBE_CALL &
wait
build auto -DMO_GENERIC_PROBE_AUTO &
build byval -DMO_GENERIC_PROBE_CALL -DMO_GENERIC_PROBE_BYVAL &
wait
build fence -DMO_GENERIC_PROBE_CALL -DMO_GENERIC_PROBE_FENCE
