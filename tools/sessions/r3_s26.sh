#!/bin/bash
# same-box A/B of the 3-D two-element kernel over the commits of the round: libraries built from git worktrees into variants/libdn_c_<commit>.so
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s26
mkdir -p $O
(for rep in 1 2; do for lib in variants/libdn_c_4a6bb17.so variants/libdn_c_03e8642.so variants/libdn_c_fd30696.so ""; do echo "== rep $rep lib=${lib:-HEAD}"; DN_LIB_PATH=$lib timeout -k 10 200 python tools/sweep3d_r2.py 256 1 "" 2>&1 | grep "n=256"; DN_LIB_PATH=$lib timeout -k 10 200 python tools/sweep3d_r2.py 257 1 "" 2>&1 | grep "n=257"; DN_LIB_PATH=$lib timeout -k 10 200 python tools/sweep3d_r2.py 128 1 "" 2>&1 | grep "n=128"; done; done) 2>&1 | tee $O/ab3d_commits.txt
