#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_fitting_gpu.py -x -q -k "mfma_skinning or full_baseline or golden" 2>&1 | grep -v amdgpu.ids | tail -5
run() {  # label env batch
  env $2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 --batch $3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('%-10s B=%-4s'%('$1','$3'), d['ms_per_step'], ' '.join('%s=%.1f'%(k.replace('_kernel',''),v.get('us')) for k,v in kb.items()))"
}
for i in 1 2; do run vector PSI_SKIN_MFMA=0 512; run mfma X=1 512; done
run vector PSI_SKIN_MFMA=0 256; run mfma X=1 256; run vector PSI_SKIN_MFMA=0 4096; run mfma X=1 4096
