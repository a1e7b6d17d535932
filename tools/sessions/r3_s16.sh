#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s16
mkdir -p $O
(
export DN_LIB_PATH=variants/libdn_pk0.so
NB=8 STREAMS=1,2 timeout -k 10 300 python tools/two_streams.py bits &&
NB=12 STREAMS=3 timeout -k 10 300 python tools/two_streams.py bits &&
NB=16 STREAMS=4,2,1 timeout -k 10 300 python tools/two_streams.py bits &&
NB=16 STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 128,4,32 &&
NB=16 STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 128,4,16,4 &&
NB=16 STREAMS=4,1 timeout -k 10 300 python tools/two_streams.py box &&
unset DN_LIB_PATH &&
NB=16 STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits &&
NB=16 STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 128,4,22 &&
NB=16 STREAMS=4 timeout -k 10 300 python tools/two_streams.py bits 128,4,32
) 2>&1 | grep -v amdgpu.ids | tee $O/streams_fair.txt
