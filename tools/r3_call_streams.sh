#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_leaf_stream_gpu.py tests/test_training_gpu.py tests/test_linear_gpu.py tests/test_conv_gpu.py -x -q > gpurun_out/t.log 2>&1; grep -E " passed| failed|Error|assert" gpurun_out/t.log | head -8
for i in 1 2; do for st in 0 1; do
  PSI_LEAF_STREAM=$st timeout 300 python bench.py --workload train_s2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('leaf_stream=$st', d['ms_per_step'])"
done; done
