#!/bin/bash
# A/B of 3-D kernel BUILD variants (extra compile flags) on the GPU box: tools/ab3d_r2.sh "<plan>" "<flags A>" "<flags B>" ...
plan=$1; shift
for flags in "$@"; do
  tools/ab_build.sh "$flags" || exit 1
  echo "== [$flags]"
  python tools/sweep3d_r2.py 256 1 $plan 2>/dev/null | grep "plan=$plan"
  python tools/sweep3d_r2.py 128 4 $plan 2>/dev/null | grep "plan=$plan"
done
tools/ab_build.sh "" || exit 1
