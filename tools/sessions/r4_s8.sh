#!/bin/bash
# round 4, session 8: bit masks through scalar loads (CF_PK_SMEM), masks packed on first use, branch-free 3-D output block
set -e
python -m pytest tests/test_gpu_compact_bc.py tests/test_gpu_round4.py tests/test_gpu_fuzz.py tests/test_gpu_plans.py -x -q > gpurun_out/r4_s8_pytest.log 2>&1 || { tail -40 gpurun_out/r4_s8_pytest.log; exit 1; }
tail -2 gpurun_out/r4_s8_pytest.log
python -m pytest tests/test_networks.py -x -q -k "gpu or matches or hip" > gpurun_out/r4_s8_pytest2.log 2>&1 || { tail -40 gpurun_out/r4_s8_pytest2.log; exit 1; }
tail -2 gpurun_out/r4_s8_pytest2.log
out=gpurun_out/r4_s8.txt
: > $out
run() { python tools/r4_time.py "$@" 2>&1 | grep -v "amdgpu.ids\|^folded\|^load vector" >> $out; }
run 2 512 64 bits sums=fold tag=smem
DN_CF_BITS_VMEM=1 run 2 512 64 bits sums=fold tag=vmem
run 2 512 64 box sums=fold tag=box
run 2 512 64 bits sums=fold tag=smem
DN_CF_BITS_VMEM=1 run 2 512 64 bits sums=fold tag=vmem
run 2 512 64 bits tag=smem-kernel-sums
run 2 256 64 bits sums=fold tag=smem-256
DN_CF_BITS_VMEM=1 run 2 256 64 bits sums=fold tag=vmem-256
python tools/step_gen3d.py 2>&1 | grep -v amdgpu.ids >> $out
cat $out
