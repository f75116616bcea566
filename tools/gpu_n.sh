#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/t; mkdir -p $O
for b in 16 64 128 256; do for sp in 0 1; do
 PSI_SPLIT_SCENE=$sp timeout 300 python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --secondary 0 > $O/sp${sp}_b$b.json 2> $O/sp${sp}_b$b.err
 python - <<PY
import json
d=json.loads([l for l in open('$O/sp${sp}_b$b.json') if l.startswith('{')][-1])
k=d['kernels_us']
print('split=$sp B=$b', d['value'], d['ms_per_step'], k.get('fwd_scene_kernel'), k.get('kd_query_kernel'), k.get('skin_fwd_sdf_kernel'))
PY
done; done
