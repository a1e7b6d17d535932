#!/bin/bash
# Sweep 3-D launch geometries (TX,TY,E,R) for the sizes of BASELINE configs[2]/[3]; prints kernel time per plan.
run() { # size batch plans...
  local size=$1 batch=$2; shift 2
  for plan in "$@"; do
    out=$(env DN_PLAN3D=$plan python bench.py --nsd 3 --size $size --ngp 2 --batch $batch --no-cpu --steps 40 --warmup 5 2>/dev/null | tail -1)
    echo "n=$size B=$batch plan=$plan $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("kern_us=%.1f min_us=%.1f frac=%.3f" % (r["kernel_avg_ms"]*1e3, r["kernel_min_ms"]*1e3, r["frac"]))')"
  done
}
if [ "$1" = "R" ]; then
run 128 1 64,4,2,8 64,4,2,12 64,4,2,16 64,4,2,24 64,4,2,32 64,4,1,8 64,4,1,16 128,2,1,16 32,8,2,16 16,16,2,16
run 128 4 64,4,2,16 64,4,2,24 64,4,2,32 64,4,2,43 64,4,2,64 64,4,2,127 32,8,2,32 16,16,2,32
run 256 1 128,2,2,16 128,2,2,32 128,2,2,64 128,2,2,128 128,2,2,255 64,4,2,32 64,4,2,64 64,4,2,128 32,8,2,64
exit 0
fi
run 128 1 default 64,4,2,8 64,4,2,4 64,8,2,8 64,8,2,4 64,8,2,16 64,16,2,8 64,16,2,16 64,16,2,4 64,12,2,8 64,6,2,8
run 128 4 default 64,4,2,16 64,8,2,8 64,8,2,16 64,8,2,32 64,16,2,8 64,16,2,16 64,16,2,32 64,12,2,16
run 256 1 default 128,2,2,8 128,4,2,8 128,4,2,16 128,8,2,8 128,8,2,16 128,8,2,32 64,16,2,16 64,8,2,16 128,6,2,16
