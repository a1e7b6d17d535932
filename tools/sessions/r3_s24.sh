#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s24
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact_bc.py tests/test_gpu_plans.py tests/test_gpu_round3.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { echo "== lib=${DN_LIB_PATH:-default} plan=$1 form=$2"; timeout -k 10 300 python tools/rotate_batches.py $1 $2 2>&1 | grep -v amdgpu.ids; }
(run default f32 && run default u8 && run default bits) 2>&1 | tee $O/rotate_f32c.txt
