#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s22
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "fsdt or plate or elasticity" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
(timeout -k 10 200 python tools/time_fsdt.py 1025 2 && DN_LIB_PATH=variants/libdn_fsdtpk0.so timeout -k 10 200 python tools/time_fsdt.py 1025 2 && timeout -k 10 200 python tools/time_fsdt.py 513 1 && DN_LIB_PATH=variants/libdn_fsdtpk0.so timeout -k 10 200 python tools/time_fsdt.py 513 1 && timeout -k 10 200 python tools/time_fsdt.py 769 3 && DN_LIB_PATH=variants/libdn_fsdtpk0.so timeout -k 10 200 python tools/time_fsdt.py 769 3) 2>&1 | grep -v amdgpu.ids | tee $O/fsdt_pk_ad.txt
