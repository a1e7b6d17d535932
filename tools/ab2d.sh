#!/bin/bash
# A/B of 2-D kernel build variants on the GPU box. usage: tools/ab2d.sh "<flags A>" "<flags B>" ...
b() { python bench.py --no-cpu --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms_step=%.4f kern_us=%.1f min_us=%.1f frac=%.3f" % (d["ms_per_step"], r["kernel_avg_ms"]*1e3, r["kernel_min_ms"]*1e3, r["frac"]))'; }
for flags in "$@"; do
  tools/ab_build.sh "$flags" || exit 1
  echo "[$flags] 512^2 B=64 g3: $(b)"
  echo "[$flags] 512^2 B=64 g3: $(b)"
  echo "[$flags] 1024^2 B=16 g3: $(b --size 1024 --batch 16)"
done
tools/ab_build.sh "" || exit 1
