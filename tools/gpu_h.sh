#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/k; mkdir -p $O
timeout 600 python -m pytest tests/test_linear_gpu.py -m gpu -q -x 2>&1 | tail -3
for h in 1 0; do
  PSI_HIP_LINEAR=$h timeout 600 python bench.py --workload train_s2 --steps 20 --warmup 3 > $O/s2_hip$h.json 2> $O/s2_hip$h.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/s2_hip$h.json') if l.startswith('{')][-1])
print('hip_linear=$h', d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'))
PY
done
cd /tmp; rm -rf /tmp/prof_s2; PSI_HIP_LINEAR=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2 -o r -- python $GRAFT_REPO_ROOT/bench.py --workload train_s2 --steps 40 --warmup 3 > /dev/null 2>&1
f=$(find /tmp/prof_s2 -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/$O/train_s2_hip_kernel_stats.csv; grep -i "linear_" $f | cut -c1-160
