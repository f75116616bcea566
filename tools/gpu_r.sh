#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lbs_gpu.py tests/test_fitting_gpu.py tests/test_parity_gaps_gpu.py tests/test_training_gpu.py -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); kb=d.get('kernel_bandwidth',{}); print({k:v.get('us') for k,v in kb.items()})"
done
