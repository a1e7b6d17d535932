#!/bin/bash
# round 4, session 2: does a fourth wave per SIMD pay?  The instantiation with nu only (no forcing, no conditions) fits 126 VGPRs
set -o pipefail
mkdir -p gpurun_out
{
python tools/r4_time.py 3 256 1 none f=0 tag=nu-only-3waves-R51
python tools/r4_time.py 3 256 1 none f=0 plan=16,16,2,37 tag=nu-only-3waves-R37
DN_LIB_PATH=variants/libdn_cf4w.so python tools/r4_time.py 3 256 1 none f=0 tag=nu-only-4waves-R51
DN_LIB_PATH=variants/libdn_cf4w.so python tools/r4_time.py 3 256 1 none f=0 plan=16,16,2,37 tag=nu-only-4waves-R37
DN_LIB_PATH=variants/libdn_cf4w.so python tools/r4_time.py 3 256 1 none f=0 plan=16,16,2,32 tag=nu-only-4waves-R32
DN_LIB_PATH=variants/libdn_cf4w.so python tools/r4_time.py 3 256 1 none f=0 plan=16,16,2,26 tag=nu-only-4waves-R26
python tools/r4_time.py 3 128 1 none f=0 tag=128-nu-only-3waves
DN_LIB_PATH=variants/libdn_cf4w.so python tools/r4_time.py 3 128 1 none f=0 plan=16,16,2,6 tag=128-nu-only-4waves-R6
DN_LIB_PATH=variants/libdn_cf4w.so python tools/r4_time.py 3 128 1 none f=0 tag=128-nu-only-4waves
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s19_times.txt
