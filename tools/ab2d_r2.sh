#!/bin/bash
# A/B of 2-D kernel BUILD variants (extra compile flags) on the GPU box: tools/ab2d_r2.sh "<flags A>" "<flags B>" ...
for flags in "$@"; do
  tools/ab_build.sh "$flags" || exit 1
  echo "== [$flags]"
  python tools/ab_kernels.py 2d 2>/dev/null | grep default
done
tools/ab_build.sh "" || exit 1
