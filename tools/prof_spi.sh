#!/bin/bash
# SPI resource-allocation stall counters for the fused kernel (why waves cannot be placed). usage: tools/prof_spi.sh [bench args]
export TMPDIR=/tmp
root=$(pwd)
rm -rf gpurun_out/spi; mkdir -p gpurun_out/spi
i=0
for ctrs in "SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN" "SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN" "SPI_RA_BAR_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN" "SPI_RA_RES_STALL_CSN SPI_RA_REQ_NO_ALLOC_CSN" "MeanOccupancyPerCU"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $ctrs --output-format csv -d $root/gpurun_out/spi/p$i -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu "$@" > $root/gpurun_out/spi/p$i.log 2>&1)
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/spi/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'poisson' not in k: continue
        agg[k.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print('   %-28s n=%d mean=%.4g' % (c,len(v),sum(v)/len(v)))
PY
