#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s18
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q -k "fsdt" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python tools/host_profile.py 2>&1 | grep -v amdgpu.ids | tee $O/host_profile.txt
