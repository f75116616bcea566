#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/h; mkdir -p $O
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --secondary 0 > $O/b32.json 2> $O/b32.err
python - <<PY
import json
d=json.loads([l for l in open('$O/b32.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['kernels_us'])
PY
timeout 900 python tools/sensitivity.py > $O/sens.log 2>&1; cp gpurun_out/sensitivity.json $O/; tail -3 $O/sens.log | cut -c1-400
