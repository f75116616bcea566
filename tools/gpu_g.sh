#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/j; mkdir -p $O
timeout 300 python tools/time_linear.py 2>&1 | tail -6
for h in 1 0; do
  PSI_HIP_LINEAR=$h timeout 600 python bench.py --workload train_s2 --steps 20 --warmup 3 > $O/s2_hip$h.json 2> $O/s2_hip$h.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/s2_hip$h.json') if l.startswith('{')][-1])
print('hip_linear=$h', d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('flops_per_step'))
PY
done
