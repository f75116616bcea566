#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3g
run() {  # label env
  env $2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('%-14s'%'$1', d['ms_per_step'], d.get('fresh_start_protocol',{}).get('ms_per_step'), ' '.join('%s=%.1f'%(k.replace('_kernel',''),v.get('us')) for k,v in kb.items()))"
}
for i in 1 2 3; do run nn_first X=1; run skin_first PSI_SCENE_ORDER=1; done | tee gpurun_out/r3g/order_ab.txt
STOPS="1 2 3 0" bash tools/skin_stops.sh 2>&1 | tee gpurun_out/r3g/stops.txt
