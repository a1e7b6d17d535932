#!/bin/bash
# round 3, GPU session 2: strip height sweep on rotating batches (is the 16-row strip = 32 KB spacing a DRAM-channel pathology?)
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s2
mkdir -p $O
export DN_LIB_PATH=$PWD/variants/libdn_w1.so
for R in 11 13 14 15 16 17 18 19 21 23; do
  timeout -k 10 300 python tools/rotate_batches.py 128,4,$R box 2>&1 | grep -v amdgpu.ids | tee -a $O/rotate_R.txt || exit 1
done
unset DN_LIB_PATH
for R in 13 15 17 19; do
  timeout -k 10 300 python tools/rotate_batches.py 128,4,$R,4 box 2>&1 | grep -v amdgpu.ids | tee -a $O/rotate_R.txt || exit 1
done
