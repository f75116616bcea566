#!/bin/bash
# dev: differential timing with the -DPSI_HEAD_STOPS variant library (fit.hip + lbs.hip): fwd_scene (1 = NN search only, 2 = skinning + SDF
# only, 3 = empty launch), skin_bwd_v (11 = empty, 12 = loads issued, 13 = + statistics, 14 = + blend), bwd_joint (21 = blend_bwd stream only,
# 22 = skin_bwd_A only, 23 = empty launch)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/stops.so
for k in ${STOPS:-1 2 3 11 12 13 14 21 22 23 0}; do
  PSI_SKIN_STOP=$k timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('stop $k', 'fwd_scene', kb['fwd_scene_kernel']['us'], 'skin_bwd_v', kb['skin_bwd_v_grad_kernel']['us'], 'bwd_joint', kb['bwd_joint_kernel']['us'])"
done
