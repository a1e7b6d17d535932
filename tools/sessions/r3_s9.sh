#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_s9
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_plans.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 900 python bench.py --no-configs --slab-size 0 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_s9/bench.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("value %.4g ms/step %.4f frac %.3f kern %.2f us steady %.2f | in-kernel sums: %.2f us frac %.3f | stream %.2f us one-batch %.2f" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"] * 1e3, r["steady_ms_per_launch_back_to_back"] * 1e3, r["kernel_avg_ms_with_in_kernel_sums"] * 1e3, r["frac_with_in_kernel_sums"], r["stream_ceiling"]["avg_ms"] * 1e3, r["one_batch_kernel_avg_ms"] * 1e3), r["rotation_kernel_median_us_by_mask_format"])
PY
