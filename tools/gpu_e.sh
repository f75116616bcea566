#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/g; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -15 $O/tests.log
for b in 32 64 128 512; do
 timeout 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --secondary 0 > $O/b$b.json 2> $O/b$b.err
 python - <<PY
import json
d=json.loads([l for l in open('$O/b$b.json') if l.startswith('{')][-1])
print($b, d['value'], d['ms_per_step'], round(d['value']*$b), d['kernels_us'])
PY
done
