#!/bin/bash
# Build a VARIANT of libdiffnet_hip.so locally (cross-compile): only the named translation unit is recompiled with the extra
# flags, the other objects come from diffnet_amd/build/.  usage: tools/variant_build.sh <tag> <tu.hip> "<flags>"
# -> variants/libdn_<tag>.so   (select it with DN_LIB_PATH=... in the tools/ timing scripts)
set -e
cd "$(dirname "$0")/.."
tag=$1; tu=$2; flags=$3
mkdir -p variants/obj
obj=variants/obj/${tag}_${tu%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -mllvm -amdgpu-sdwa-peephole=0 --offload-arch=gfx950 -w $flags -c diffnet_amd/csrc/$tu -o $obj
objs=""
for o in diffnet_amd/build/*.o; do
  if [ "$(basename $o)" == "${tu%.hip}.o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o variants/libdn_${tag}.so
echo "built variants/libdn_${tag}.so [$flags]"
