#!/bin/bash
# round 3 evidence in one call: bench line (default and with the driver's arguments), rocprofv3 kernel stats of the bench command, counter passes
# of the 2-D bench kernel (default plan, mask bits; and the chained plan), the slab timeline, the host-side costs
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
root=$(pwd)
O=gpurun_out/r3_ev
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs > $O/bench_driver_args.json 2>/dev/null; echo "bench (driver args) rc=$?"
rm -rf $O/kt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$O/kt -- python3 $root/bench.py --no-cpu --slab-size 0 --no-configs > $root/$O/kt.log 2>&1); echo "kt rc=$?"
rm -rf $O/kt_slab
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$O/kt_slab -- python3 $root/tools/slab_timeline.py > $root/$O/slab_timeline.txt 2>&1); echo "slab kt rc=$?"
DN_BC_FORM=bits tools/prof_case.sh r3_2d_bits 2 512 64 3 "" 12 > /dev/null 2>&1; echo "pmc bits rc=$?"
DN_BC_FORM=bits tools/prof_case.sh r3_2d_bits_chained 2 512 64 3 "128,4,16,4" 12 > /dev/null 2>&1; echo "pmc chained rc=$?"
python tools/host_profile.py 2>&1 | grep -v amdgpu.ids | head -8 > $O/host_profile.txt
head -4 $O/kt/*/*_kernel_stats.csv | cut -c1-220
head -6 $O/kt_slab/*/*_kernel_stats.csv | cut -c1-220
grep -v amdgpu.ids $O/slab_timeline.txt
grep -n "FETCH_SIZE\|WRITE_SIZE\|^_ZN\|^void\|dn::" gpurun_out/pmc_r3_2d_bits.txt gpurun_out/pmc_r3_2d_bits_chained.txt | head -20
cat $O/host_profile.txt
python - <<'PY'
import json
for f in ("bench", "bench_driver_args"):
    d = json.loads(open("gpurun_out/r3_ev/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, "value %.4g ms/step %.4f frac %.3f kern %.2f us steady %.2f stream %.2f us one-batch %.2f" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"] * 1e3, r["steady_ms_per_launch_back_to_back"] * 1e3, r["stream_ceiling"]["avg_ms"] * 1e3, r["one_batch_kernel_avg_ms"] * 1e3), r["rotation_kernel_median_us_by_mask_format"])
PY
