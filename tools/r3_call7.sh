#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
for lin in "" 1 "" 1; do
PSI_HIP_LINEAR=$lin python bench.py --workload train_s2 --steps 10 --warmup 3 2>gpurun_out/r3e/bench_lin.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('PSI_HIP_LINEAR=$lin', d['ms_per_step'], d['ms_per_step_min'], d.get('roofline',{}).get('frac'))"
done | tee gpurun_out/r3e/train_s2_lin_ab.txt
PSI_HIP_LINEAR=1 python -m pytest tests/test_training_gpu.py tests/test_parity_gaps_gpu.py -x -q -k "train or s2 or cal_loss or graph" 2>&1 | grep -v amdgpu.ids | tail -5
