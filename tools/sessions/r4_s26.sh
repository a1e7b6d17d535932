#!/bin/bash
# round 4, session 2: the fixed per-layer chain (no arithmetic, only u requested) taken apart: publish, hand-over, barrier
set -o pipefail
mkdir -p gpurun_out
{
for v in nomath_noreq a_pub a_xch a_pub_xch a_pub_xch_bar; do
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 256 1 u8 tag=$v
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 256 1 u8 sums=0 tag=$v-nosums
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s26_times.txt
