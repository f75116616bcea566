#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_cvae_glue_gpu.py tests/test_linear_gpu.py -x -q > gpurun_out/t.log 2>&1; grep -E " passed| failed|Error" gpurun_out/t.log | head -5
for i in 1 2; do for st in 0 1; do
  PSI_TRUNK_STREAMS=$st timeout 300 python bench.py --workload train_s2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('trunk_streams=$st', d['ms_per_step'])"
done; done
