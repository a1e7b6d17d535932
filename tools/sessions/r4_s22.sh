#!/bin/bash
# round 4, session 2: do the arrays' relative start addresses matter?  (64 MB fields allocated back to back share their low address bits)
set -o pipefail
mkdir -p gpurun_out
{
python tools/r4_time.py 3 256 1 u8 tag=cfg4-pad0
python tools/r4_time.py 3 256 1 u8 pad=4096 tag=cfg4-pad4k
python tools/r4_time.py 3 256 1 u8 pad=1280 tag=cfg4-pad1280
python tools/r4_time.py 3 256 1 u8 pad=69632 tag=cfg4-pad68k
python tools/r4_time.py 3 256 1 u8 pad=1114112 tag=cfg4-pad1088k
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 256 1 u8 tag=cfg4-nomath-pad0
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 256 1 u8 pad=69632 tag=cfg4-nomath-pad68k
python tools/r4_time.py 3 128 1 u8 tag=cfg3-pad0
python tools/r4_time.py 3 128 1 u8 pad=69632 tag=cfg3-pad68k
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s22_times.txt
