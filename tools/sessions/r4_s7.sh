#!/bin/bash
set -e
python -m pytest tests/test_networks.py tests/test_gpu_round4.py -x -q -k "conv2d or unet512 or network_matches or hip_blocks" > gpurun_out/r4_s7_pytest.log 2>&1 || { tail -40 gpurun_out/r4_s7_pytest.log; exit 1; }
tail -2 gpurun_out/r4_s7_pytest.log
out=gpurun_out/r4_s7.txt
: > $out
for wgs in 2048 1280 1024 768; do
  echo "== CONV_WRW_WGS=$wgs" >> $out
  DN_CONV_WRW_WGS=$wgs python tools/bench_conv2d.py 16 2>&1 | grep -v amdgpu >> $out
  DN_CONV_WRW_WGS=$wgs python tools/step_unet.py 2>&1 | grep -v amdgpu >> $out
done
cat $out | awk '{ if ($0 ~ /^==/ || $0 ~ /sum over/ || $0 ~ /UNet/ || $0 ~ /down1/ || $0 ~ /up4/) print }'
