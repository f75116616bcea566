#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for nb in 2 3 4 6 8; do
  PSI_SKA_NBODY=$nb timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('nbody $nb', d['ms_per_step'], 'bwd_joint', kb['bwd_joint_kernel']['us'])"
done
