#!/bin/bash
# dev: differential timing of the head / tail kernels — the variant library built with -DPSI_HEAD_STOPS leaves the kernels at point k
# (tools/mkvariant.sh stops psi-release_amd/csrc/fit.hip fit.hip -DPSI_HEAD_STOPS); HIP-event time of the truncated kernels per k
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/stops.so
for k in 1 2 3 4 5 6 7 8 0; do
  PSI_HEAD_STOP=$k PSI_TAIL_STOP=$k timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('stop $k', 'head_fwd', kb['head_fwd_kernel']['us'], 'head_bwd_adam', kb['head_bwd_adam_kernel']['us'])"
done
