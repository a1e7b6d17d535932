#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s17
mkdir -p $O
(
export DN_LIB_PATH=variants/libdn_pk0.so
NSETS=16 NBS=4,8,16,4,2,1 timeout -k 10 300 python tools/rotate_batches.py default bits &&
NSETS=16 NBS=4,8,16 timeout -k 10 300 python tools/rotate_batches.py default box
) 2>&1 | grep -v amdgpu.ids | tee $O/rotate_nb.txt
