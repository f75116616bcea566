#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
python -m pytest tests/test_bnorm_gpu.py -x -q > gpurun_out/r3d/pytest_bn.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d/pytest_bn.log
tail -30 gpurun_out/r3d/pytest_bn.log
python -m pytest tests/test_configs_gpu.py tests/test_training_gpu.py -x -q > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d/pytest.log
tail -8 gpurun_out/r3d/pytest.log
for bn in 0 1 0 1; do
PSI_HIP_BN=$bn python bench.py --workload train_s2 --steps 10 --warmup 3 2>gpurun_out/r3d/bench_bn$bn.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('PSI_HIP_BN=$bn', d['ms_per_step'], d['ms_per_step_min'], d.get('roofline',{}).get('frac'))"
done | tee gpurun_out/r3d/train_s2_ab.txt
tail -3 gpurun_out/r3d/bench_bn1.err; python tools/time_bn.py 2>/dev/null | tee gpurun_out/r3d/time_bn.txt
