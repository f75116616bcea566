#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_training_gpu.py tests/test_section8f_gpu.py tests/test_cvae_glue_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do
  timeout 300 python bench.py --workload train_s2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('now', d['ms_per_step'])"
done
