#!/bin/bash
# round 4, session 2: what the store costs in the access pattern alone, on the cube (row pieces straddle lines) and on a mesh one chunk wide (row pieces = whole aligned lines)
set -o pipefail
mkdir -p gpurun_out
{
for v in pat pat_nost pat_nohalo pat_nohalo_nost; do
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 256 1 u8 sums=0 tag=cube-$v
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 0 1 u8 sums=0 sizes=32,2048,256 tag=rows32-$v
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s35_times.txt
