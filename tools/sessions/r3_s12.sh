#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s12
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plans.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { echo "== lib=${DN_LIB_PATH:-default} plan=$1 form=$2"; timeout -k 10 300 python tools/rotate_batches.py $1 $2 2>&1 | grep -v amdgpu.ids; }
(
run default bits && run 128,4,22 bits && run 128,4,32 bits && run 128,4,11 bits && run default box &&
DN_LIB_PATH=variants/libdn_pk0.so run default bits && DN_LIB_PATH=variants/libdn_pk0.so run default box &&
DN_LIB_PATH=variants/libdn_pkw4.so run default bits
) 2>&1 | tee $O/rotate_pk.txt
