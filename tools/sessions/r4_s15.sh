#!/bin/bash
# round 4, session 2: closed-form 3-D kernel, even / odd node columns in LDS (conflict-free read2 pairs), two planes in flight
set -o pipefail
mkdir -p gpurun_out
python -c "from diffnet_amd import _lib; print(_lib.lib().dn_build_info().decode())"
timeout -k 10 900 python -m pytest tests/test_gpu_q1cf3d.py "tests/test_gpu_round3.py::test_3d_two_elements_per_thread_equals_one" -x -q > gpurun_out/s15_tests.log 2>&1
rc=$?
tail -3 gpurun_out/s15_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
{
python tools/r4_time.py 3 256 1 u8 tag=cfg4
python tools/r4_time.py 3 128 1 u8 tag=cfg3
python tools/r4_time.py 3 256 1 u8 sums=fold tag=cfg4-fold
python tools/r4_time.py 3 128 1 u8 sums=fold tag=cfg3-fold
python tools/r4_time.py 3 256 1 u8 cfg=Q1_3D_N2:1 tag=cfg4-r3kernel
python tools/r4_time.py 3 256 1 none tag=cfg4-nomask
python tools/r4_time.py 3 256 1 u8 load=1 tag=cfg4-load
python tools/r4_time.py 3 256 1 none load=1 tag=cfg4-load-nomask
python tools/r4_time.py 3 256 1 u8 f=0 tag=cfg4-nof
python tools/r4_time.py 3 256 1 u8 f=0 nu=0 tag=cfg4-bare
python tools/r4_time.py 3 256 1 box tag=cfg4-box
python tools/r4_time.py 3 256 1 f32 tag=cfg4-f32
python tools/r4_time.py 3 128 8 u8 tag=128x8
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s15_times.txt
DN_LIB_PATH=variants/libdn_stamp.so python tools/stamp3d.py 256 1 2>&1 | grep -v "amdgpu.ids" | head -12 | tee gpurun_out/s15_stamp256.txt
DN_LIB_PATH=variants/libdn_stamp.so python tools/stamp3d.py 128 1 2>&1 | grep -v "amdgpu.ids" | head -12 | tee gpurun_out/s15_stamp128.txt
