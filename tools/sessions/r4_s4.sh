#!/bin/bash
# round 4, session 4: whole GPU suite, bench line (driver's arguments), world-2 gloo rehearsal of both bench modes on one GPU
set -e
python -m pytest tests -x -q -m gpu > gpurun_out/r4_s4_pytest.log 2>&1 || { tail -40 gpurun_out/r4_s4_pytest.log; exit 1; }
tail -3 gpurun_out/r4_s4_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_s4_bench.json 2> gpurun_out/r4_s4_bench.err || { tail -30 gpurun_out/r4_s4_bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_s4_bench.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "steady", d["steady_state"]["value"], d["steady_state"]["ms_per_step"])
r = d["roofline"]
print("frac", r["frac"], "kernel_avg_ms", r["kernel_avg_ms"], "in-kernel sums", r["kernel_avg_ms_with_in_kernel_sums"], "stream", r["stream_ceiling"]["avg_ms"], "by mask", r["rotation_kernel_median_us_by_mask_format"])
for row in d.get("configs", []):
    print(row)
print("slab", {k: d.get("slab_3d", {}).get(k) for k in ("ms_per_step", "hbm_frac_of_all_gpus", "error")}, {k: d.get("slab_3d_b8", {}).get(k) for k in ("ms_per_step", "hbm_frac_of_all_gpus", "error")})
print("cpu", d.get("cpu_baseline"))
PY
DN_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 12 --warmup 3 --settle 20 --no-configs --no-cpu --slab-steps 5 --slab-warmup 5 > gpurun_out/r4_s4_gloo2.json 2> gpurun_out/r4_s4_gloo2.err || { tail -30 gpurun_out/r4_s4_gloo2.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4_s4_gloo2.json") if l.startswith("{")][-1])
print("gloo world 2:", d["n_gpus"], d["value"], d["ms_per_step"], d["roofline"]["sums_mode"], {k: d.get("slab_3d", {}).get(k) for k in ("ms_per_step", "error")}, {k: d.get("slab_3d_b8", {}).get(k) for k in ("ms_per_step", "error")})
PY
