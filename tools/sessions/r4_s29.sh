#!/bin/bash
# round 4, session 2: the access pattern alone, piece by piece: halo requests, mask requests, store
set -o pipefail
mkdir -p gpurun_out
{
for v in pat pat_nohalo pat_nomask pat_nohalo_nomask pat_nohalo_nomask_nost full_nohalo; do
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 256 1 u8 sums=0 tag=$v-nosums
done
DN_LIB_PATH=variants/libdn_full_nohalo.so python tools/r4_time.py 3 256 1 u8 tag=full_nohalo
python tools/r4_time.py 3 256 1 u8 tag=full
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s29_times.txt
