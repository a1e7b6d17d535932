#!/bin/bash
# round 4, session 2: one against two planes in flight, finer stamps of phase A
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_q1cf3d.py -x -q > gpurun_out/s17_tests.log 2>&1 || { tail -5 gpurun_out/s17_tests.log; exit 1; }
tail -1 gpurun_out/s17_tests.log
{
python tools/r4_time.py 3 256 1 u8 tag=cfg4-pf2
python tools/r4_time.py 3 128 1 u8 tag=cfg3-pf2
DN_LIB_PATH=variants/libdn_pf1.so python tools/r4_time.py 3 256 1 u8 tag=cfg4-pf1
DN_LIB_PATH=variants/libdn_pf1.so python tools/r4_time.py 3 128 1 u8 tag=cfg3-pf1
python tools/r4_time.py 3 256 1 u8 cfg=Q1_3D_N2:1 tag=cfg4-r3kernel
python tools/r4_time.py 3 256 1 u8 tag=cfg4-pf2-again
DN_LIB_PATH=variants/libdn_pf1.so python tools/r4_time.py 3 256 1 u8 tag=cfg4-pf1-again
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s17_times.txt
DN_LIB_PATH=variants/libdn_stamp.so python tools/stamp3d.py 256 1 2>&1 | grep -v "amdgpu.ids" | head -10 | tee gpurun_out/s17_stamp256.txt
