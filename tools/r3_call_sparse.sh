#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for a in "--batch 512 --weight-nnz 4" "--batch 4096 --weight-nnz 4" "--batch 32 --weight-nnz 4"; do
timeout 300 python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$a', d['ms_per_step'], d['value'], json.dumps(d.get('kernel_bandwidth')), json.dumps(d.get('roofline')))"
done
