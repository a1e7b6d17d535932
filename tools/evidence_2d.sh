export TMPDIR=/tmp
root=$(pwd)
DN_BC_FORM=u8 tools/prof_case.sh r2_2d 2 512 64 3 "" 12 > /dev/null 2>&1; echo "pmc u8 rc=$?"
DN_BC_FORM=bits tools/prof_case.sh r2_2d_bits 2 512 64 3 "" 12 > /dev/null 2>&1; echo "pmc bits rc=$?"
python bench.py > gpurun_out/r2_final_bench.json 2> gpurun_out/r2_final_bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_args.json 2>/dev/null; echo "bench2 rc=$?"
rm -rf gpurun_out/r2_final_kt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r2_final_kt -- python3 $root/bench.py --no-cpu --slab-size 0 > $root/gpurun_out/r2_final_kt.log 2>&1); echo "kt rc=$?"
grep -n "FETCH_SIZE\|WRITE_SIZE" gpurun_out/pmc_r2_2d.txt gpurun_out/pmc_r2_2d_bits.txt
head -3 gpurun_out/r2_final_kt/*/*_kernel_stats.csv | cut -c1-200
python - <<PY
import json
for f in ("r2_final_bench","bench_driver_args"):
    d=json.loads(open("gpurun_out/%s.json"%f).readlines()[-1]); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["kernel_median_ms"], r["timed_region_ms_per_launch"], r["frac_over_timed_region_one_batch"], r["one_batch_kernel_median_us_by_mask_format"])
PY
