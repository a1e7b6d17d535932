#!/bin/bash
# rocprofv3 counter passes (SQ / TCC / SPI groups in separate runs, no trace domains with --pmc) + a kernel-trace pass for one
# tools/run_case.py invocation.  usage: tools/prof_case.sh <tag> <run_case args...>      output: gpurun_out/pmc_<tag>.txt
#        DN_PROF_SCRIPT=tools/other.py tools/prof_case.sh <tag> <args of that script...>   profiles another driver script
tag=$1; shift
script=${DN_PROF_SCRIPT:-tools/run_case.py}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
            "SQ_IFETCH SQ_IFETCH_LEVEL SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU" \
            "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "MeanOccupancyPerCU"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $root/$script "$@" > $out/p$i.log 2>&1)
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/$script "$@" > $out/kt.log 2>&1)
python3 - <<PY > $root/gpurun_out/pmc_$tag.txt
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'dn::' not in k: continue
        agg[k.split('(')[0][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
print("command: tools/prof_case.sh $tag $*")
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print('   %-28s n=%d mean=%.5g' % (c,len(v),sum(v)/len(v)))
for f in glob.glob('$out/kt/*/*kernel_stats.csv'):
    print("kernel stats:", f.split('/')[-1])
    for r in csv.DictReader(open(f)):
        if 'dn::' in r['Name']: print('   ', r['Name'][:100], 'calls', r['Calls'], 'avg_ns', r['AverageNs'], 'min_ns', r['MinNs'], 'max_ns', r['MaxNs'])
PY
cat $root/gpurun_out/pmc_$tag.txt
