#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s4
mkdir -p $O
for lib in w1 w1_ntc w1_ntall w4_ntc w4_ntall; do
  for form in box bits; do
    export DN_LIB_PATH=$PWD/variants/libdn_$lib.so
    timeout -k 10 300 python tools/rotate_batches.py default $form 2>&1 | grep -v amdgpu.ids | tee -a $O/rotate.txt || exit 1
  done
done
