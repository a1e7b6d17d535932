#!/bin/bash
# round 4, session 3: new tests (load vector, box faces in 3-D, pipelined sums); opposite marches with pairs on one XCD; 3-D load + box
set -e
out=gpurun_out/r4_s3.txt
: > $out
run() { python tools/r4_time.py "$@" >> $out 2>&1; }
python -m pytest tests/test_gpu_round4.py -x -q -k "load_vector or box_faces or pipelined or handover" > gpurun_out/r4_s3_pytest.log 2>&1 || { tail -40 gpurun_out/r4_s3_pytest.log; exit 1; }
tail -3 gpurun_out/r4_s3_pytest.log
run 2 512 64 bits sums=fold tag=fold
DN_LIB_PATH=variants/libdn_revpairs.so run 2 512 64 bits sums=fold tag=revpairs+fold
DN_LIB_PATH=variants/libdn_revpairs2.so run 2 512 64 bits sums=fold tag=revpairs2+fold
DN_LIB_PATH=variants/libdn_revpairs2.so run 2 512 64 box sums=fold tag=revpairs2+fold
run 3 256 1 u8 load=1 sums=fold tag=load+fold iters=200
run 3 256 1 box load=1 sums=fold tag=load+box+fold iters=200
run 3 256 1 box sums=fold tag=box+fold iters=200
run 3 128 1 box load=1 sums=fold tag=load+box+fold iters=400
run 3 128 1 u8 sums=fold tag=fold iters=400
cat $out
