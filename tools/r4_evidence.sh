#!/bin/bash
# round 4 evidence in one call: GPU test suite, bench line (default and with the driver's arguments), rocprofv3 kernel stats of the bench command,
# counter passes of the 2-D bench kernel (mask bits, folded sums) and of the closed-form 3-D kernel at 256^3 / 128^3, phase stamps of the 3-D kernel
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
root=$(pwd)
O=gpurun_out/r4_ev
mkdir -p $O
python -c "from diffnet_amd import _lib; print(_lib.lib().dn_build_info().decode())"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs > $O/bench_driver_args.json 2>/dev/null; echo "bench (driver args) rc=$?"
rm -rf $O/kt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$O/kt -- python3 $root/bench.py --no-cpu --slab-size 0 --no-configs > $root/$O/kt.log 2>&1); echo "kt rc=$?"
DN_BC_FORM=bits DN_SUMS=fold tools/prof_case.sh r4_2d_bits 2 512 64 3 "" 12 > /dev/null 2>&1; echo "pmc 2d bits rc=$?"
tools/prof_case.sh r4_3d_256_cf 3 256 1 2 "" 12 > /dev/null 2>&1; echo "pmc 3d 256 rc=$?"
tools/prof_case.sh r4_3d_128_cf 3 128 1 2 "" 12 > /dev/null 2>&1; echo "pmc 3d 128 rc=$?"
if [ -f variants/libdn_stamp.so ]; then
  DN_LIB_PATH=variants/libdn_stamp.so python tools/stamp3d.py 256 1 2>&1 | grep -v amdgpu.ids | head -12 > $O/stamp256.txt
  DN_LIB_PATH=variants/libdn_stamp.so python tools/stamp3d.py 128 1 2>&1 | grep -v amdgpu.ids | head -12 > $O/stamp128.txt
fi
# second half of the round: FSDT stencil form against the element form (single launches, rotation-free: B = 8 fits the 256 MB cache, see the pair), the loss + gradient
# pair with deferred sums, the cost of the in-kernel sums, the generic operators through the raw launches, the 3-D generator's training step
(python tools/time_fsdt.py 1025 2; DN_FSDT_FORM=elem python tools/time_fsdt.py 1025 2) 2>&1 | grep -v amdgpu.ids > $O/fsdt_forms.txt
python tools/time_fsdt_sums.py 2>&1 | grep -v amdgpu.ids > $O/fsdt_sums.txt
(python tools/time_fsdt_pair.py; DN_FSDT_FORM=elem python tools/time_fsdt_pair.py) 2>&1 | grep -v amdgpu.ids > $O/fsdt_pair.txt
python tools/bench_ops.py 2>&1 | grep -v amdgpu.ids > $O/ops.txt
(python tools/step_gen3d.py --steps 10; python tools/step_gen3d.py --steps 5 --size 256) 2>&1 | grep -v amdgpu.ids > $O/gen3d.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$O/kt_gen3d -- python3 $root/tools/step_gen3d.py --steps 10 > $root/$O/kt_gen3d.log 2>&1); python tools/trace_step.py $(ls $O/kt_gen3d/*/*kernel_trace.csv | tail -1) poisson3d_q1_cf 30000 > $O/gen3d_step_kernels.txt 2>&1
tail -n 20 $O/fsdt_forms.txt $O/fsdt_pair.txt $O/gen3d.txt
head -4 $O/kt/*/*_kernel_stats.csv | cut -c1-220
grep -n "FETCH_SIZE\|WRITE_SIZE\|^void\|dn::" gpurun_out/pmc_r4_2d_bits.txt gpurun_out/pmc_r4_3d_256_cf.txt | head -20
python - <<'PY'
import json
for f in ("bench", "bench_driver_args"):
    d = json.loads(open("gpurun_out/r4_ev/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, "value %.4g ms/step %.4f frac %.3f kern %.2f us" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"] * 1e3))
    for row in d.get("configs", []):
        print("   %9.2f us  frac %s  %s" % (row["us_per_eval"], row.get("frac"), row["name"][:100]))
PY
