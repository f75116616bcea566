#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
python -m pytest tests/test_bnorm_gpu.py tests/test_linear_gpu.py tests/test_training_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -6
for bn in 1 0 1; do
PSI_HIP_BN=$bn python bench.py --workload train_s2 --steps 10 --warmup 3 2>gpurun_out/r3f/bench.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('PSI_HIP_BN=$bn', d['ms_per_step'], d['ms_per_step_min'], d.get('roofline',{}).get('frac'))"
done | tee gpurun_out/r3f/train_s2.txt
tail -3 gpurun_out/r3f/bench.err
