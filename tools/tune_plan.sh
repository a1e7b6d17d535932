#!/bin/bash
# Sweep launch geometries of the fused kernel on the GPU box (tuning aid; prints kernel_avg_ms per setting).
# usage: tools/tune_plan.sh "<bench args>" plan1 plan2 ...   (plans: "T,E,R" for 2-D, "TX,TY,E,R" for 3-D)
args="$1"; shift
var=DN_PLAN2D
case "$args" in *"--nsd 3"*) var=DN_PLAN3D;; esac
for plan in "$@"; do
  out=$(env $var=$plan python bench.py $args --no-cpu --steps 60 --warmup 10 2>/dev/null | tail -1)
  echo "$var=$plan $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms_step=%.4f kern_ms=%.4f GB/s=%.0f frac=%.3f" % (d["ms_per_step"], r["kernel_avg_ms"], r["achieved"], r["frac"]))')"
done
