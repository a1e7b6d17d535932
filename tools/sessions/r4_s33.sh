#!/bin/bash
# round 4, session 2: weight gradient of the U-Net output block with 16-byte staging loads
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
root=$(pwd)
timeout -k 10 900 python -m pytest tests/test_networks.py -x -q -m gpu > gpurun_out/s33_tests.log 2>&1 || { tail -30 gpurun_out/s33_tests.log; exit 1; }
tail -1 gpurun_out/s33_tests.log
python tools/step_unet.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/s33_step.txt
rm -rf gpurun_out/s33_kt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/s33_kt -- python3 $root/tools/step_unet.py --steps 13 > $root/gpurun_out/s33_kt.log 2>&1)
python - <<'PY' | tee gpurun_out/s33_kernels.txt
import csv, glob
rows = []
for f in glob.glob('gpurun_out/s33_kt/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((float(r['TotalDurationNs']), int(r['Calls']), float(r['AverageNs']), r['Name'][:110]))
rows.sort(reverse=True)
steps = 13 + 3
tot = sum(r[0] for r in rows) / steps / 1e6
print(f"U-Net(2->1) 512^2 B=16 training step, kernel time per step ({steps} steps traced): total {tot:.3f} ms")
for t, c, a, n in rows[:16]:
    print(f"  {t / steps / 1e6:8.3f} ms  {c / steps:5.1f} calls  avg {a / 1e3:8.1f} us  {n}")
PY
