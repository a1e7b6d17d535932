#!/bin/bash
# rocprofv3 PMC passes for the fused kernel (separate passes per counter group, as MI355X_MICROARCH.md prescribes:
# no trace domains combined with --pmc).  usage: tools/prof_pmc.sh <tag> [bench args...]
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p gpurun_out/pmc_$tag
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $ctrs --output-format csv -d $root/gpurun_out/pmc_$tag/p$i -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu "$@" > $root/gpurun_out/pmc_$tag/p$i.log 2>&1)
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_$tag/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'poisson' not in k: continue
        agg[k.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print('   %-24s n=%d mean=%.4g' % (c,len(v),sum(v)/len(v)))
PY
