#!/bin/bash
# Round-end evidence in ONE gpurun call: GPU test suite, bench line (default and with the driver's arguments), rocprofv3 kernel stats of the
# bench command, per-config timings, counter passes of the 2-D bench kernel (box condition) and of the 3-D kernel at 256^3.
# Everything lands under gpurun_out/ (copy what is judged to profiles/).  DN_SKIP_PMC=1 skips the counter passes.
export TMPDIR=/tmp
root=$(pwd)
python -m pytest tests -x -q -m gpu > gpurun_out/r2_final_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2_final_pytest.log
python bench.py > gpurun_out/r2_final_bench.json 2> gpurun_out/r2_final_bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_args.json 2>/dev/null; echo "bench (driver args) rc=$?"
rm -rf gpurun_out/r2_final_kt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r2_final_kt -- python3 $root/bench.py --no-cpu --slab-size 0 > $root/gpurun_out/r2_final_kt.log 2>&1); echo "kt rc=$?"
python tools/bench_configs.py gpurun_out/configs.json > gpurun_out/configs.txt 2>&1; echo "configs rc=$?"
if [ -z "$DN_SKIP_PMC" ]; then
  DN_BC_FORM=box tools/prof_case.sh r2_2d_box 2 512 64 3 "" 12 > /dev/null 2>&1; echo "pmc 2d rc=$?"
  tools/prof_case.sh r2_3d256_n 3 256 1 2 "" 12 > /dev/null 2>&1; echo "pmc 3d rc=$?"
fi
tail -3 gpurun_out/r2_final_pytest.log; head -3 gpurun_out/r2_final_kt/*/*_kernel_stats.csv | cut -c1-200; grep frac gpurun_out/configs.txt; cat gpurun_out/r2_final_bench.json
