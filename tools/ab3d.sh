#!/bin/bash
# A/B of 3-D kernel build variants on the GPU box. usage: tools/ab3d.sh "<flags A>" "<flags B>" ...
b() { python bench.py --nsd 3 --size $1 --ngp 2 --batch $2 --no-cpu --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("kern_us=%.1f min_us=%.1f frac=%.3f" % (r["kernel_avg_ms"]*1e3, r["kernel_min_ms"]*1e3, r["frac"]))'; }
for flags in "$@"; do
  tools/ab_build.sh "$flags" || exit 1
  for cfg in "128 1" "128 4" "256 1"; do echo "[$flags] $cfg: $(b $cfg)"; done
done
tools/ab_build.sh "" || exit 1
