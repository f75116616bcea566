#!/bin/bash
# A/B of two library builds in the same call (box-to-box variation is 2-3 %): tools/gpu_ab.sh <variantA.so|default> <variantB.so|default> [rounds]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A=$1; B=$2; N=${3:-3}
run() {
  if [ "$1" = default ]; then unset PSI_HIP_LIB; else export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/$1; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('$1', d['ms_per_step'], ' '.join('%s=%.1f'%(k.replace('_kernel',''),v.get('us')) for k,v in kb.items()))"
}
for i in $(seq $N); do run $A; run $B; done
