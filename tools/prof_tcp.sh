#!/bin/bash
# rocprofv3 TA / TCP / UTCL1 counter passes (vector-memory pipeline of the CU) for one tools/run_case.py invocation.
# usage: tools/prof_tcp.sh <tag> <run_case args...>      output: gpurun_out/pmc_tcp_<tag>.txt
tag=$1; shift
script=${DN_PROF_SCRIPT:-tools/run_case.py}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/pmc_tcp_$tag
mkdir -p $out
i=0
for ctrs in "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS" \
            "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_TOTAL_ACCESSES" \
            "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_GATE_EN1" \
            "TCP_TOTAL_READ TCP_GATE_EN2" \
            "TCP_UTCL1_THRASHING_STALL TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS" \
            "TCP_TCR_TCP_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES" \
            "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $root/$script "$@" > $out/p$i.log 2>&1)
done
python3 - <<PY > $root/gpurun_out/pmc_tcp_$tag.txt
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'dn::' not in k: continue
        agg[k.split('(')[0][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
print("command: tools/prof_tcp.sh $tag $*")
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print('   %-40s n=%d mean=%.5g' % (c,len(v),sum(v)/len(v)))
PY
cat $root/gpurun_out/pmc_tcp_$tag.txt
