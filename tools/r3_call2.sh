#!/bin/bash
# round 3, GPU call 2: apron bricks + grid ball query — parity tests, then A/B against the previous library in the same call
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
python -m pytest tests/test_hip_ops_gpu.py tests/test_fitting_gpu.py tests/test_configs_gpu.py tests/test_dist_gpu.py -x -q > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest.log
tail -6 gpurun_out/r3c/pytest.log
run() {  # label lib env batch
  if [ "$2" = default ]; then unset PSI_HIP_LIB; else export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/$2; fi
  env $3 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 --batch $4 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('%-14s B=%-4s'%('$1','$4'), d['ms_per_step'], ' '.join('%s=%.1f'%(k.replace('_kernel',''),v.get('us')) for k,v in kb.items()))"
}
for i in 1 2 3; do
  run base base.so X=1 32; run apron default PSI_NN_GRID=0 32; run apron+grid default X=1 32
done | tee gpurun_out/r3c/ab32.txt
for i in 1 2; do
  run base base.so X=1 512; run apron default PSI_NN_GRID=0 512; run apron+grid default X=1 512
done | tee gpurun_out/r3c/ab512.txt
run base base.so X=1 128; run apron+grid default X=1 128
