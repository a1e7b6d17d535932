#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s14
mkdir -p $O
(timeout -k 10 300 python tools/two_streams.py bits && timeout -k 10 300 python tools/two_streams.py box && DN_LIB_PATH=variants/libdn_pk0.so timeout -k 10 300 python tools/two_streams.py bits && timeout -k 10 300 python tools/two_streams.py bits 128,4,22) 2>&1 | grep -v amdgpu.ids | tee $O/two_streams.txt
