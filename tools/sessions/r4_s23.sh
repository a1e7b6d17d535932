#!/bin/bash
# round 4, session 2: is it the row pieces' alignment?  Meshes with the node count of 256^3 whose rows are one chunk wide (32 nodes = one 128-byte line,
# line-aligned, no halo column) or two (62 nodes), against the cube
set -o pipefail
mkdir -p gpurun_out
{
python tools/r4_time.py 3 256 1 u8 tag=cube256
python tools/r4_time.py 3 0 1 u8 sizes=32,2048,256 tag=rows-of-32
python tools/r4_time.py 3 0 1 u8 sizes=64,1024,256 tag=rows-of-64
python tools/r4_time.py 3 0 1 u8 sizes=32,512,1024 tag=rows-of-32-small-planes
python tools/r4_time.py 3 0 1 u8 sizes=1024,64,256 tag=rows-of-1024
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 0 1 u8 sizes=32,2048,256 tag=nomath-rows-of-32
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 0 1 u8 sizes=1024,64,256 tag=nomath-rows-of-1024
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s23_times.txt
