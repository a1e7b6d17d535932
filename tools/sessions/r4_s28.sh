#!/bin/bash
# round 4, session 2: the access pattern alone -- every request and the store of the kernel, no LDS, no hand-over, no barrier, no arithmetic
set -o pipefail
mkdir -p gpurun_out
{
for v in pat pat_pf2 pat_bar; do
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 256 1 u8 tag=$v
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 256 1 u8 sums=0 tag=$v-nosums
done
DN_LIB_PATH=variants/libdn_pat.so python tools/r4_time.py 3 128 1 u8 tag=128-pat
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s28_times.txt
