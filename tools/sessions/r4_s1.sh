#!/bin/bash
# round 4, session 1: 2-D reduction cost / opposite marches without the XCD map; 3-D mask-path ablations
set -e
out=gpurun_out/r4_s1.txt
: > $out
run() { python tools/r4_time.py "$@" >> $out 2>&1; }
run 2 512 64 bits tag=base
run 2 512 64 bits sums=0 tag=nosums
run 2 512 64 bits sums=defer tag=defer
run 2 512 64 box tag=base
run 2 512 64 u8 tag=base
DN_LIB_PATH=variants/libdn_rev.so run 2 512 64 bits tag=rev
DN_LIB_PATH=variants/libdn_revnox.so run 2 512 64 bits tag=revnox
DN_LIB_PATH=variants/libdn_revnox.so run 2 512 64 box tag=revnox
run 2 512 64 bits tag=base2
run 3 256 1 u8 tag=base iters=200
run 3 256 1 none tag=nomask iters=200
run 3 256 1 f32 tag=f32 iters=200
run 3 256 1 u8 sums=0 tag=nosums iters=200
run 3 256 1 u8 f=0 tag=nof iters=200
run 3 256 1 u8 nu=0 f=0 tag=neither iters=200
run 3 128 1 u8 tag=base iters=400
run 3 128 1 none tag=nomask iters=400
run 3 128 1 u8 sums=0 tag=nosums iters=400
cat $out
