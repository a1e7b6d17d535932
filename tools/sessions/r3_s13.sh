#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s13
mkdir -p $O
run() { echo "== lib=${DN_LIB_PATH:-default} plan=$1 form=$2"; timeout -k 10 300 python tools/rotate_batches.py $1 $2 2>&1 | grep -v amdgpu.ids; }
(
DN_LIB_PATH=variants/libdn_pk0.so run 128,4,22 bits && DN_LIB_PATH=variants/libdn_pk0.so run 128,4,32 bits && run 128,4,22 box && run 128,4,24 bits && run 128,4,20 bits
) 2>&1 | tee $O/rotate_pk2.txt
