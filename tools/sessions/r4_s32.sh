#!/bin/bash
# round 4, session 2: the store without a branch (exec mask inside one asm block): the compiler's vmcnt waits become exact; one / two planes in flight
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_q1cf3d.py tests/test_gpu_round4.py -x -q > gpurun_out/s32_tests.log 2>&1 || { tail -30 gpurun_out/s32_tests.log; exit 1; }
tail -1 gpurun_out/s32_tests.log
{
python tools/r4_time.py 3 256 1 u8 tag=cfg4-pf1
DN_LIB_PATH=variants/libdn_pf2.so python tools/r4_time.py 3 256 1 u8 tag=cfg4-pf2
python tools/r4_time.py 3 128 1 u8 tag=cfg3-pf1
DN_LIB_PATH=variants/libdn_pf2.so python tools/r4_time.py 3 128 1 u8 tag=cfg3-pf2
python tools/r4_time.py 3 256 1 u8 cfg=Q1_3D_N2:1 tag=cfg4-r3kernel
python tools/r4_time.py 3 256 1 box load=1 sums=fold tag=cfg4-box-load-fold-pf1
DN_LIB_PATH=variants/libdn_pf2.so python tools/r4_time.py 3 256 1 box load=1 sums=fold tag=cfg4-box-load-fold-pf2
python tools/r4_time.py 3 256 1 none tag=cfg4-nomask-pf1
DN_LIB_PATH=variants/libdn_pf2.so python tools/r4_time.py 3 256 1 none tag=cfg4-nomask-pf2
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s32_times.txt
