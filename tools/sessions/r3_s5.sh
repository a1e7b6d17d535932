#!/bin/bash
# round 3, GPU session 5: full GPU suite on the new defaults, FSDT chain sweep, bench
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s5
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 900 python tools/time_fsdt.py 1025 2 192,2 192,4 64,2,10 64,2,12 64,1,9 64,2,9 64,2,6 64,4,4 64,3,8 64,4,8 64,2,4 > $O/fsdt.txt 2>&1 || { tail -20 $O/fsdt.txt; exit 1; }
grep -v amdgpu.ids $O/fsdt.txt
