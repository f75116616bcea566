#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/d; mkdir -p $O
for nb in 1 2 3 4 6 8; do
 PSI_SKA_NBODY=$nb timeout 120 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --secondary 0 > $O/nb$nb.json 2> $O/nb$nb.err
 python - <<PY
import json
d=json.loads([l for l in open('$O/nb$nb.json') if l.startswith('{')][-1])
print($nb, d['value'], d['ms_per_step'], d['kernels_us'])
PY
done
