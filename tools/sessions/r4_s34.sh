#!/bin/bash
# round 4, session 2: FSDT kernel with one vector-memory instruction per array and node row (4-byte aligned 8- / 12- / 16-byte accesses)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "fsdt or elasticity or plate" > gpurun_out/s34_tests.log 2>&1 || { tail -30 gpurun_out/s34_tests.log; exit 1; }
tail -1 gpurun_out/s34_tests.log
{
echo "== after"; python tools/time_fsdt.py 1025 2
echo "== before (same box)"; DN_LIB_PATH=variants/libdn_fsdt_before.so python tools/time_fsdt.py 1025 2
echo "== after, 513^2 Q2 / 512^2 Q1 / 511^2 ... Q3 (766 = 3*255+1)"; python tools/time_fsdt.py 513 2; python tools/time_fsdt.py 512 1; python tools/time_fsdt.py 766 3
echo "== before"; DN_LIB_PATH=variants/libdn_fsdt_before.so python tools/time_fsdt.py 513 2; DN_LIB_PATH=variants/libdn_fsdt_before.so python tools/time_fsdt.py 512 1; DN_LIB_PATH=variants/libdn_fsdt_before.so python tools/time_fsdt.py 766 3
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s34_times.txt
