#!/bin/bash
# round 4, session 2: full GPU suite with the closed-form 3-D kernel as the default, counters + kernel stats of cfg4 / cfg3
set -o pipefail
mkdir -p gpurun_out
python -c "from diffnet_amd import _lib; print(_lib.lib().dn_build_info().decode())"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/s16_tests.log 2>&1
rc=$?
tail -3 gpurun_out/s16_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
tools/prof_case.sh r4_3d_256_cf 3 256 1 2 "" 12 > /dev/null 2>&1 && tail -5 gpurun_out/pmc_r4_3d_256_cf.txt
tools/prof_case.sh r4_3d_128_cf 3 128 1 2 "" 12 > /dev/null 2>&1 && tail -5 gpurun_out/pmc_r4_3d_128_cf.txt
