#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2; do for st in 0 1; do
  PSI_MIOPEN_FIND=$st timeout 400 python bench.py --workload train_s2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('miopen_find=$st', d['ms_per_step'])"
done; done
