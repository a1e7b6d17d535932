#!/bin/bash
# round 4, session 5: v2 convolution kernels: parity tests of the conv family, per-layer rates (v2 / round-2 forms / MIOpen), U-Net step
set -e
python -m pytest tests/test_networks.py tests/test_gpu_round4.py -x -q -k "conv2d or unet512 or network_matches or hip_blocks" > gpurun_out/r4_s5_pytest.log 2>&1 || { tail -40 gpurun_out/r4_s5_pytest.log; exit 1; }
tail -3 gpurun_out/r4_s5_pytest.log
out=gpurun_out/r4_s5.txt
: > $out
echo "== v2" >> $out
python tools/bench_conv2d.py 16 --miopen >> $out 2>&1
echo "== round-2 forms (DN_CONV2D_V1=1)" >> $out
DN_CONV2D_V1=1 python tools/bench_conv2d.py 16 >> $out 2>&1
echo "== U-Net step v2 / v1" >> $out
python tools/step_unet.py >> $out 2>&1
DN_CONV2D_V1=1 python tools/step_unet.py >> $out 2>&1
grep -v amdgpu.ids $out
