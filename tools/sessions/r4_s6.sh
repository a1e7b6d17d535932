#!/bin/bash
# kernel trace of the U-Net training step (v2 convolutions)
set -e
export TMPDIR=/tmp
root=$(pwd)
mkdir -p gpurun_out/r4_unet_kt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r4_unet_kt -- python3 $root/tools/step_unet.py --steps 10 > $root/gpurun_out/r4_unet_kt.log 2>&1)
python3 - <<'PY'
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
f = glob.glob(root + "/gpurun_out/r4_unet_kt/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
steps = 13
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = ["U-Net(2->1) 512^2 B=16 training step, kernel time per step (13 steps traced): total %.3f ms" % (tot / steps / 1e6)]
for r in rows[:40]:
    out.append("  %8.3f ms %5.1f calls  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
open(root + "/gpurun_out/r4_unet_step_kernels.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -1 gpurun_out/r4_unet_kt.log
