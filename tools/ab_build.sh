#!/bin/bash
# A/B library: tools/ab_build.sh NAME "-DFLAG ..." [file.hip ...]  ->  tools/ab_libs/libminiopt_NAME.so  (objects of the named files rebuilt
# with the extra flags in a scratch directory, every other object taken from the product build).  Use: MO_LIB_PATH=tools/ab_libs/... python ...
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/mini_opt_amd/csrc
make -s -C "$src" -j8
tmp=$(mktemp -d)
objs=""
for f in mo_api eig_kernels kkt_generic kkt_fused kkt_fused_gather kkt_fused_ny2 kkt_fused_ny34 kkt_fused_mc4 kkt_fused_tiny kkt_fused_f32 nls_kernels; do
  rebuilt=0
  for g in "$@"; do [ "$g" = "$f.hip" ] && rebuilt=1; done
  if [ $rebuilt = 1 ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast \
      -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-function-calls=false -DMO_TUNING $flags -I"$src" -c "$src/$f.hip" -o "$tmp/$f.o" &
    objs="$objs $tmp/$f.o"
  else
    objs="$objs $src/build/$f.o"
  fi
done
wait
mkdir -p "$root/tools/ab_libs"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$root/tools/ab_libs/libminiopt_$name.so" $objs
rm -rf "$tmp"
echo "built tools/ab_libs/libminiopt_$name.so"
