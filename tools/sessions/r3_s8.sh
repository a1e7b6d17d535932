#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s8
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python tools/host_profile.py 2>&1 | grep -v amdgpu.ids | head -8 | tee $O/host_profile.txt
