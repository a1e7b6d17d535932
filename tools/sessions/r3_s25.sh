#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s25
mkdir -p $O
run() { echo "== lib=${DN_LIB_PATH:-default} plan=$1 form=$2"; timeout -k 10 300 python tools/rotate_batches.py $1 $2 2>&1 | grep -v amdgpu.ids; }
(for lib in pf2 pf1; do for plan in 128,4,16 128,4,32 128,4,24 64,4,32; do DN_LIB_PATH=variants/libdn_$lib.so run $plan bits; done; done; run 128,4,16 bits) 2>&1 | tee $O/rotate_pf.txt
