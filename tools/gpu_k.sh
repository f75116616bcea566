#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/q; mkdir -p $O
timeout 900 python -m pytest tests/test_fitting_gpu.py tests/test_parity_gaps_gpu.py -m gpu -q -x 2>&1 | tail -3
for lin in 0 1; do for b in 32 512; do
 PSI_SDF_LINEAR=$lin timeout 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --secondary 0 > $O/l${lin}_b$b.json 2> $O/l${lin}_b$b.err
 python - <<PY
import json
d=json.loads([l for l in open('$O/l${lin}_b$b.json') if l.startswith('{')][-1])
print('linear=$lin B=$b', d['value'], d['ms_per_step'], d['kernels_us'].get('skin_fwd_sdf_kernel'))
PY
done; done
