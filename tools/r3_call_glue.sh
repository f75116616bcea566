#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cvae_glue_gpu.py tests/test_training_gpu.py -x -q 2>&1 | tail -15
for i in 1 2; do
for g in 0 1; do
  PSI_HIP_GLUE=$g timeout 300 python bench.py --workload train_s2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('glue=$g', d['ms_per_step'])"
done; done
