#!/bin/bash
# SQ / MFMA counters of one conv2d_k4s2 contraction: tools/prof_conv.sh <tag> <run_conv args...>  -> gpurun_out/pmc_<tag>.txt
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
            "GRBM_GUI_ACTIVE" "MeanOccupancyPerCU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $root/tools/run_conv.py "$@" > $out/p$i.log 2>&1)
done
python3 - <<PY > $root/gpurun_out/pmc_$tag.txt
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'dn::' not in k: continue
        agg[k.split('(')[0][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
print("command: tools/prof_conv.sh $tag $*")
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print('   %-28s n=%d mean=%.5g' % (c,len(v),sum(v)/len(v)))
PY
cat $root/gpurun_out/pmc_$tag.txt
