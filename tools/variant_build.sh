#!/bin/bash
# Build a VARIANT of libdiffnet_hip.so for A/B measurements: the listed sources (default: the 2-D closed-form kernel) are recompiled with
# extra flags, everything else is taken from the objects of the default build (diffnet_amd/build/*.o).  The result goes to
# variants/libdn_<name>.so (git-ignored, travels with gpurun); select it with DN_LIB_PATH.  Fails loudly.
# usage: tools/variant_build.sh <name> "<flags>" [source.hip ...]
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
srcs=${@:-poisson2d_q1_cf.hip}
base="-O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize --offload-arch=gfx950 -mllvm -amdgpu-sdwa-peephole=0 -Wall -Wno-unused-variable -Wno-unused-but-set-variable"
mkdir -p variants/objs
objs=""
skip=""
for s in $srcs; do
    o=variants/objs/${name}_${s%.hip}.o
    /opt/rocm/bin/hipcc $base $flags -c diffnet_amd/csrc/$s -o $o
    objs="$objs $o"
    skip="$skip ${s%.hip}.o"
done
for o in diffnet_amd/build/*.o; do
    b=$(basename $o)
    case " $skip " in *" $b "*) ;; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o variants/libdn_${name}.so
echo "built variants/libdn_${name}.so [$flags]"
