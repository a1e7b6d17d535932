#!/bin/bash
# round 4, session 2: first GPU run of the closed-form-in-z 3-D Q1 kernel: parity, then same-box A/B against the round-3 kernel
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_q1cf3d.py -x -q > gpurun_out/s10_tests.log 2>&1
rc=$?
tail -5 gpurun_out/s10_tests.log
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
} 2>&1 | grep -v Warning | tee gpurun_out/s10_times.txt
