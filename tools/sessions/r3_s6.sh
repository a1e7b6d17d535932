#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s6
mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "fsdt" > $O/pytest_fsdt.log 2>&1 || { tail -40 $O/pytest_fsdt.log; exit 1; }
tail -2 $O/pytest_fsdt.log
timeout -k 10 900 python tools/time_fsdt.py 1025 2 192,2 192,4 64,2,10 64,2,12 64,3,4 64,4,4 64,6,4 64,8,4 64,3,12 > $O/fsdt.txt 2>&1 || { tail -20 $O/fsdt.txt; exit 1; }
grep -v amdgpu.ids $O/fsdt.txt
