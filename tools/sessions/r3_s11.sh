#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s11
mkdir -p $O
export DN_LIB_PATH=variants/libdn_stamp.so
(timeout -k 10 300 python tools/stamp3d.py 256 1 && timeout -k 10 300 python tools/stamp3d.py 241 2 16,16,2,240 && timeout -k 10 300 python tools/stamp3d.py 241 4 16,16,2,240 && timeout -k 10 300 python tools/stamp3d.py 241 6 16,16,2,240) 2>&1 | grep -v amdgpu.ids | tee $O/stamp_n2.txt
