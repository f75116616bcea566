#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3h
python -m pytest tests/test_fitting_gpu.py tests/test_parity_gaps_gpu.py tests/test_configs_gpu.py tests/test_dist_gpu.py tests/test_stress_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5
run() {  # label env batch
  env $2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 --batch $3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('%-10s B=%-4s'%('$1','$3'), d['ms_per_step'], (d.get('fresh_start_protocol') or {}).get('ms_per_step'), ' '.join('%s=%.1f'%(k.replace('_kernel',''),v.get('us')) for k,v in kb.items()))"
}
for i in 1 2 3; do run mode1 PSI_SCENE_MODE=1 32; run mode2 X=1 32; done | tee gpurun_out/r3h/scene_mode_ab.txt
run mode1 PSI_SCENE_MODE=1 64; run mode2 X=1 64; run mode1 PSI_SCENE_MODE=1 128; run mode2 X=1 128; run mode1 PSI_SCENE_MODE=1 8; run mode2 X=1 8
