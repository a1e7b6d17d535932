#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s19
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_plans.py -x -q -k "graded" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { echo "== plan=$1 form=$2"; timeout -k 10 300 python tools/rotate_batches.py $1 $2 2>&1 | grep -v amdgpu.ids; }
(
run 128,4,16 bits && run 128,4,16,1,8 bits && run 128,4,16,1,16 bits && run 128,4,16,1,21 bits && run 128,4,16,1,26 bits && run 128,4,16,1,32 bits && run default bits && run default box && run 128,4,16 box
) 2>&1 | tee $O/rotate_skew.txt
