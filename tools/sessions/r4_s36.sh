#!/bin/bash
# round 4, session 2: the per-layer chain where the access pattern does not limit (mesh one chunk wide: pattern alone 73 us, complete kernel 95 us): what shortens it?
set -o pipefail
mkdir -p gpurun_out
{
S="sizes=32,2048,256"
python tools/r4_time.py 3 0 1 u8 $S tag=rows32-default
for v in pf2 nobar nomath noxch; do
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 0 1 u8 $S tag=rows32-$v
done
python tools/r4_time.py 3 0 1 none f=0 $S tag=rows32-nu-only-3waves
DN_LIB_PATH=variants/libdn_cf4w.so python tools/r4_time.py 3 0 1 none f=0 $S plan=16,16,2,43 tag=rows32-nu-only-4waves-R43
DN_LIB_PATH=variants/libdn_cf4w.so python tools/r4_time.py 3 0 1 none f=0 $S tag=rows32-nu-only-4waves-R51
python tools/r4_time.py 3 0 1 none f=0 $S plan=16,16,2,43 tag=rows32-nu-only-3waves-R43
python tools/r4_time.py 3 0 1 u8 $S load=1 tag=rows32-load
python tools/r4_time.py 3 0 1 u8 $S f=0 nu=0 tag=rows32-bare
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s36_times.txt
