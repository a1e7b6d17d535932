#!/bin/bash
# round 4, session 2: the memory side alone, one piece removed at a time
set -o pipefail
mkdir -p gpurun_out
{
for v in nomath nogather nomath_nobar nomath_noreq; do
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 256 1 u8 tag=$v
done
DN_LIB_PATH=variants/libdn_nogather.so python tools/r4_time.py 3 256 1 u8 sums=0 tag=nogather-nosums
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 256 1 u8 sums=0 tag=nomath-nosums
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s24_times.txt
