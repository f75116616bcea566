#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r; mkdir -p $O
for v in base hb1024 base hb1024; do
 if [ $v = base ]; then unset PSI_HIP_LIB; else export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/$v.so; fi
 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --secondary 0 > $O/$v.json 2> $O/$v.err
 python - <<PY
import json
d=json.loads([l for l in open('$O/$v.json') if l.startswith('{')][-1])
k=d['kernels_us']; print('$v', d['value'], d['ms_per_step'], k.get('head_fwd_kernel'), k.get('head_bwd_adam_kernel'))
PY
done
unset PSI_HIP_LIB
PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/hb1024.so timeout 600 python -m pytest tests/test_fitting_gpu.py -m gpu -q -x 2>&1 | tail -2
