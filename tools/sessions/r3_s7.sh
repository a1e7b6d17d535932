#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s7
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_s7/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/step %.4f frac %.3f kern_avg %.2f us one_batch %.2f us stream %s" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"] * 1e3, r["one_batch_kernel_avg_ms"] * 1e3, r["stream_ceiling"]))
print(r["rotation_kernel_median_us_by_mask_format"]); print(r.get("deeper_rotation"))
for c in d.get("configs", []):
    print(c)
print(d.get("slab_3d"))
print(d.get("cpu_baseline"))
PY
