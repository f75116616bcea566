#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_hip_ops_gpu.py tests/test_fitting_gpu.py -x -q 2>&1 | grep -E "passed|failed"
run() {  # label env batch
  env $2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 --batch $3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('%-10s B=%-4s'%('$1','$3'), d['ms_per_step'], (d.get('fresh_start_protocol') or {}).get('ms_per_step'), ' '.join('%s=%.1f'%(k.replace('_kernel',''),v.get('us')) for k,v in kb.items()))"
}
run merged X=1 32; run split PSI_SPLIT_SCENE=1 32; run merged X=1 32; run split PSI_SPLIT_SCENE=1 32; run merged X=1 64; run merged X=1 512
