#!/bin/bash
# round 3, GPU session 1: chained strips -- parity first, then the rotation timings of the chain lengths, then the bench line
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r3_s1
mkdir -p $O
python -m pytest tests/test_gpu_plans.py -x -q > $O/pytest_plans.log 2>&1 || { tail -30 $O/pytest_plans.log; exit 1; }
tail -3 $O/pytest_plans.log
for lib in default w1 w2 w8; do
  for form in box bits; do
    if [ $lib = default ]; then unset DN_LIB_PATH; else export DN_LIB_PATH=$PWD/variants/libdn_$lib.so; fi
    timeout -k 10 300 python tools/rotate_batches.py default $form 2>&1 | grep -v amdgpu.ids | tee -a $O/rotate_w.txt || exit 1
  done
done
unset DN_LIB_PATH
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_s1/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/step %.4f frac %.3f kern_avg %.2f us one_batch %.2f us stream %s" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"] * 1e3, r["one_batch_kernel_avg_ms"] * 1e3, r["stream_ceiling"]))
print(r["rotation_kernel_median_us_by_mask_format"])
for c in d.get("configs", []):
    print(c)
print(d.get("slab_3d"))
PY
