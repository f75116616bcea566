#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for r in 1 2; do for o in 0 1 2; do
  PSI_SCENE_ORDER=$o timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('order $o', d['ms_per_step'], 'fwd_scene', kb['fwd_scene_kernel']['us'])"
done; done
