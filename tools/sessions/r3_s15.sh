#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s15
mkdir -p $O
export DN_LIB_PATH=variants/libdn_pk0.so
(
STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 128,4,32 &&
STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 128,4,64 &&
STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 128,4,8 &&
STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 128,4,16,4 &&
STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 128,4,8,4 &&
STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 128,4,32,4 &&
STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 64,4,16 &&
NB=8 STREAMS=4,6,8 timeout -k 10 300 python tools/two_streams.py bits &&
NB=8 STREAMS=4,8 timeout -k 10 300 python tools/two_streams.py box
) 2>&1 | grep -v amdgpu.ids | tee $O/streams_plans.txt
