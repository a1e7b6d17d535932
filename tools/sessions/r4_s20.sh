#!/bin/bash
# round 4, session 2: the memory side of the closed-form 3-D kernel alone (no element arithmetic)
set -o pipefail
mkdir -p gpurun_out
{
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 256 1 u8 tag=cfg4-nomath
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 128 1 u8 tag=cfg3-nomath
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 256 1 none tag=cfg4-nomath-nomask
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 256 1 u8 f=0 nu=0 tag=cfg4-nomath-bare
python tools/r4_time.py 3 256 1 u8 tag=cfg4
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s20_times.txt
