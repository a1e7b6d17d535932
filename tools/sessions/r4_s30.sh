#!/bin/bash
# round 4, session 2: non-temporal stores / loads in the closed-form 3-D kernel
set -o pipefail
mkdir -p gpurun_out
{
python tools/r4_time.py 3 256 1 u8 tag=default
for v in nts ntl ntls; do
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 256 1 u8 tag=$v
done
python tools/r4_time.py 3 128 1 u8 tag=128-default
DN_LIB_PATH=variants/libdn_nts.so python tools/r4_time.py 3 128 1 u8 tag=128-nts
DN_LIB_PATH=variants/libdn_nts.so python tools/r4_time.py 3 128 8 u8 tag=128x8-nts
python tools/r4_time.py 3 128 8 u8 tag=128x8-default
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s30_times.txt
