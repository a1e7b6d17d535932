#!/bin/bash
# round 4, session 2: box-face conditions without a mask image: in-plane result worked out once per thread
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_q1cf3d.py tests/test_gpu_round4.py tests/test_gpu_compact_bc.py -x -q > gpurun_out/s31_tests.log 2>&1 || { tail -30 gpurun_out/s31_tests.log; exit 1; }
tail -1 gpurun_out/s31_tests.log
{
python tools/r4_time.py 3 256 1 u8 tag=cfg4-u8
python tools/r4_time.py 3 256 1 box tag=cfg4-box
python tools/r4_time.py 3 256 1 box load=1 tag=cfg4-box-load
python tools/r4_time.py 3 256 1 box load=1 sums=fold tag=cfg4-box-load-fold
python tools/r4_time.py 3 128 1 box tag=cfg3-box
python tools/r4_time.py 3 128 1 box load=1 sums=fold tag=cfg3-box-load-fold
python tools/r4_time.py 3 256 1 none tag=cfg4-nomask
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s31_times.txt
