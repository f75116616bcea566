#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s; mkdir -p $O
timeout 1200 python -m pytest tests/test_fitting_gpu.py tests/test_parity_gaps_gpu.py tests/test_dist_gpu.py tests/test_section8f_gpu.py -m gpu -q -x 2>&1 | tail -3
for sp in 0 1 0 1; do for b in 32; do
 PSI_SPLIT_SCENE=$sp timeout 300 python bench.py --batch $b --steps 100 --warmup 5 --no-cpu-baseline --secondary 0 > $O/sp${sp}_b$b.json 2> $O/sp${sp}_b$b.err
 python - <<PY
import json
d=json.loads([l for l in open('$O/sp${sp}_b$b.json') if l.startswith('{')][-1])
print('split=$sp B=$b', d['value'], d['ms_per_step'], d['kernels_us'])
PY
done; done
for sp in 0 1; do PSI_SPLIT_SCENE=$sp timeout 300 python bench.py --batch 512 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 > $O/sp${sp}_b512.json 2> $O/sp${sp}_b512.err
python - <<PY
import json
d=json.loads([l for l in open('$O/sp${sp}_b512.json') if l.startswith('{')][-1])
print('split=$sp B=512', d['value'], d['ms_per_step'], d['kernels_us'])
PY
done
