#!/bin/bash
# round 4, session 2: closed-form 3-D kernel with two planes in flight, one LDS block with literal slots: parity, A/B, phase stamps
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_q1cf3d.py tests/test_gpu_round4.py tests/test_gpu_plans.py "tests/test_gpu_round3.py::test_3d_two_elements_per_thread_equals_one" -x -q > gpurun_out/s12_tests.log 2>&1
rc=$?
tail -3 gpurun_out/s12_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
{
for cfg in "" "Q1_3D_N2:1"; do
  python tools/r4_time.py 3 256 1 u8 cfg=$cfg tag=cfg4
  python tools/r4_time.py 3 128 1 u8 cfg=$cfg tag=cfg3
  python tools/r4_time.py 3 256 1 u8 cfg=$cfg sums=fold tag=cfg4-fold
  python tools/r4_time.py 3 128 1 u8 cfg=$cfg sums=fold tag=cfg3-fold
done
python tools/r4_time.py 3 256 1 u8 load=1 tag=cfg4-load
python tools/r4_time.py 3 256 1 u8 f=0 tag=cfg4-nof
python tools/r4_time.py 3 256 1 u8 f=0 nu=0 tag=cfg4-bare
python tools/r4_time.py 3 256 1 box tag=cfg4-box
python tools/r4_time.py 3 256 1 f32 tag=cfg4-f32
python tools/r4_time.py 3 128 8 u8 tag=128x8
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s12_times.txt
DN_LIB_PATH=variants/libdn_stamp.so python tools/stamp3d.py 256 1 2>&1 | grep -v "amdgpu.ids" | head -12 | tee gpurun_out/s12_stamp256.txt
DN_LIB_PATH=variants/libdn_stamp.so python tools/stamp3d.py 128 1 2>&1 | grep -v "amdgpu.ids" | head -12 | tee gpurun_out/s12_stamp128.txt
