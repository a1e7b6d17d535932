#!/bin/bash
# round 4, session 2: the memory side alone with one / two planes in flight
set -o pipefail
mkdir -p gpurun_out
{
for v in nomath nomath_pf2 nomath_noreq nomath_noreq_pf2; do
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 256 1 u8 tag=$v
done
DN_LIB_PATH=variants/libdn_nomath_pf2.so python tools/r4_time.py 3 128 1 u8 tag=128-nomath_pf2
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 128 1 u8 tag=128-nomath
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s25_times.txt
