#!/bin/bash
# round 4, session 2: closed-form 3-D kernel with the tile's nodes requested four per lane by waves that share the fields
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_q1cf3d.py "tests/test_gpu_round3.py::test_3d_two_elements_per_thread_equals_one" -x -q > gpurun_out/s21_tests.log 2>&1 || { tail -30 gpurun_out/s21_tests.log; exit 1; }
tail -1 gpurun_out/s21_tests.log
{
python tools/r4_time.py 3 256 1 u8 tag=cfg4
python tools/r4_time.py 3 128 1 u8 tag=cfg3
python tools/r4_time.py 3 256 1 u8 sums=fold tag=cfg4-fold
python tools/r4_time.py 3 128 1 u8 sums=fold tag=cfg3-fold
python tools/r4_time.py 3 256 1 u8 cfg=Q1_3D_N2:1 tag=cfg4-r3kernel
DN_LIB_PATH=variants/libdn_nomath.so python tools/r4_time.py 3 256 1 u8 tag=cfg4-nomath
python tools/r4_time.py 3 256 1 none tag=cfg4-nomask
python tools/r4_time.py 3 256 1 u8 load=1 tag=cfg4-load
python tools/r4_time.py 3 256 1 u8 f=0 tag=cfg4-nof
python tools/r4_time.py 3 256 1 u8 f=0 nu=0 tag=cfg4-bare
python tools/r4_time.py 3 256 1 box tag=cfg4-box
python tools/r4_time.py 3 256 1 f32 tag=cfg4-f32
python tools/r4_time.py 3 128 8 u8 tag=128x8
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s21_times.txt
DN_LIB_PATH=variants/libdn_stamp.so python tools/stamp3d.py 256 1 2>&1 | grep -v "amdgpu.ids" | head -10 | tee gpurun_out/s21_stamp256.txt
