#!/bin/bash
# several library builds in the same call: tools/gpu_abn.sh <rounds> <lib|default>...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
N=$1; shift
run() {
  if [ "$1" = default ]; then unset PSI_HIP_LIB; else export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/$1; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('%-12s'%'$1', d['ms_per_step'], ' '.join('%s=%.1f'%(k.replace('_kernel',''),v.get('us')) for k,v in kb.items()))"
}
for i in $(seq $N); do for l in "$@"; do run $l; done; done
