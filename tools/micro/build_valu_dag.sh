#!/bin/bash
# two-pass build: count the VALU instructions the compiler left in the loop, then bake that count into the binary
set -e
cd "$(dirname "$0")"
for cfg in "48 256 0" "80 256 0" "48 256 1" "48 256 2" "48 256 4"; do
  set -- $cfg
  python3 gen_valu_dag.py $1 $2 $3 > valu_dag_body.h
  /opt/rocm/bin/hipcc -w -O3 -ffp-contract=fast -fno-slp-vectorize --offload-arch=gfx950 valu_dag.hip -o /tmp/vd.bin --save-temps 2>/dev/null
  n=$(awk '/^.LBB0_[0-9]+:/{f=1;c=0} f&&/^[ \t]*v_/{c++} /s_cbranch_scc[01] .LBB0_/{if(f){print c; f=0}}' valu_dag-hip-amdgcn-amd-amdhsa-gfx950.s | sort -n | tail -1)
  echo "cfg $cfg loop VALU = $n"
  /opt/rocm/bin/hipcc -w -O3 -ffp-contract=fast -fno-slp-vectorize --offload-arch=gfx950 -DNLOOP=$n valu_dag.hip -o valu_dag_$1_$3.bin
  rm -f valu_dag-h* valu_dag.hip-*
done
