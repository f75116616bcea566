#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fitting_gpu.py tests/test_lbs_gpu.py tests/test_parity_gaps_gpu.py -m gpu -q -x 2>&1 | tail -2
for hc in 1 8; do
  echo "== PSI_HEAD_CLUSTER=$hc"
  PSI_HEAD_CLUSTER=$hc timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); kb=d.get('kernel_bandwidth',{}); print({k:v.get('us') for k,v in kb.items()})"
done
timeout 300 python tools/phase_clock.py 2>&1 | tail -22
