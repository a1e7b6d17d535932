#!/bin/bash
# round 4, session 2: load-vector forcing in the 3-D kernel (3 and 4 waves per SIMD), folded final reduction in 2-D / 3-D; new tests
set -e
out=gpurun_out/r4_s2.txt
: > $out
run() { python tools/r4_time.py "$@" >> $out 2>&1; }
python -m pytest tests/test_gpu_round4.py -x -q -k "handover or 3d_128" > gpurun_out/r4_s2_pytest.log 2>&1 || { tail -30 gpurun_out/r4_s2_pytest.log; exit 1; }
tail -3 gpurun_out/r4_s2_pytest.log
run 2 512 64 bits tag=base
run 2 512 64 bits sums=fold tag=fold
run 2 512 64 box sums=fold tag=fold
run 2 512 64 u8 sums=fold tag=fold
run 3 256 1 u8 tag=base iters=200
run 3 256 1 u8 sums=fold tag=fold iters=200
run 3 256 1 u8 load=1 tag=load iters=200
run 3 256 1 u8 load=1 sums=fold tag=load+fold iters=200
run 3 256 1 none load=1 tag=load-nomask iters=200
DN_LIB_PATH=variants/libdn_n2w4.so run 3 256 1 u8 load=1 tag=load-w4 iters=200
DN_LIB_PATH=variants/libdn_n2w4.so run 3 256 1 u8 load=1 plan=16,16,2,43 tag=load-w4-6strips iters=200
DN_LIB_PATH=variants/libdn_n2w4.so run 3 256 1 u8 load=1 plan=16,16,2,37 tag=load-w4-7strips iters=200
DN_LIB_PATH=variants/libdn_n2w4.so run 3 256 1 u8 load=1 nu=0 plan=16,16,2,43 tag=load-w4-6strips-nonu iters=200
run 3 256 1 u8 load=1 nu=0 tag=load-nonu iters=200
run 3 256 1 u8 load=1 nu=0 plan=16,16,2,43 tag=load-nonu-6strips iters=200
run 3 128 1 u8 tag=base iters=400
run 3 128 1 u8 load=1 tag=load iters=400
run 3 128 1 u8 load=1 sums=fold tag=load+fold iters=400
DN_LIB_PATH=variants/libdn_n2w4.so run 3 128 1 u8 load=1 tag=load-w4 iters=400
cat $out
