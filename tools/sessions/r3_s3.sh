#!/bin/bash
# round 3, GPU session 3: shared node from the neighbouring lane + non-temporal row loads, with and without chained strips
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s3
mkdir -p $O
for lib in w4_lane_nt w1_lane_nt; do
  DN_LIB_PATH=$PWD/variants/libdn_$lib.so python -m pytest tests/test_gpu_plans.py -x -q -k "chained or paired" > $O/pytest_$lib.log 2>&1 || { tail -30 $O/pytest_$lib.log; exit 1; }
  tail -1 $O/pytest_$lib.log
done
for lib in default w1 w4_lane w4_lane_ntc w4_lane_nt w2_lane_nt w1_lane w1_lane_nt; do
  for form in box bits; do
    if [ $lib = default ]; then unset DN_LIB_PATH; else export DN_LIB_PATH=$PWD/variants/libdn_$lib.so; fi
    timeout -k 10 300 python tools/rotate_batches.py default $form 2>&1 | grep -v amdgpu.ids | tee -a $O/rotate.txt || exit 1
  done
done
