#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_conv_gpu.py tests/test_bnorm_gpu.py tests/test_training_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4
for cv in 1 0 1; do
PSI_HIP_CONV_WRW=$cv python bench.py --workload train_s2 --steps 10 --warmup 3 2>gpurun_out/bench_cv.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('PSI_HIP_CONV_WRW=$cv', d['ms_per_step'], d['ms_per_step_min'], d.get('roofline',{}).get('frac'))"
done
tail -3 gpurun_out/bench_cv.err
