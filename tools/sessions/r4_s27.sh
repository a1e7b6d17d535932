#!/bin/bash
# round 4, session 2: do the stores hold back the retirement of the loads behind them (vmcnt retires in order)?
set -o pipefail
mkdir -p gpurun_out
{
for v in nomath_noreq a_st nomath nomath_st full_st; do
  DN_LIB_PATH=variants/libdn_$v.so python tools/r4_time.py 3 256 1 u8 tag=$v
done
python tools/r4_time.py 3 256 1 u8 tag=full
DN_LIB_PATH=variants/libdn_full_st.so python tools/r4_time.py 3 128 1 u8 tag=128-full_st
python tools/r4_time.py 3 128 1 u8 tag=128-full
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/s27_times.txt
