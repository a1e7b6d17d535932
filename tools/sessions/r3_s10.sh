#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s10
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q -k "two_elements" > $O/pytest_n2.log 2>&1 || { tail -40 $O/pytest_n2.log; exit 1; }
tail -2 $O/pytest_n2.log
timeout -k 10 600 python tools/sweep3d_r2.py 256 1 "" 16,16,2,32 16,16,2,51 16,16,2,64 16,16,2,16 16,16,1,51 2>&1 | grep -v amdgpu.ids | tee $O/sweep256.txt
timeout -k 10 600 python tools/sweep3d_r2.py 128 1 "" 16,16,2,9 16,16,2,16 16,16,2,32 16,16,1,9 2>&1 | grep -v amdgpu.ids | tee $O/sweep128.txt
