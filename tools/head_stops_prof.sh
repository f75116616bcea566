#!/bin/bash
# dev: rocprofv3 averages of the head / tail kernels left at stop point k (the -DPSI_HEAD_STOPS library): differential timing without the
# launch gaps a HIP-event measurement includes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/stops.so
for k in 1 2 3 4 5 6 7 8 21 22 23 24 0; do
  rm -rf /tmp/hs; ( cd /tmp; PSI_HEAD_STOP=$k PSI_TAIL_STOP=$k rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hs -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 > /dev/null 2>&1 )
  python - "$(find /tmp/hs -name '*kernel_stats.csv' | head -1)" $k <<'PY'
import csv, sys
rows = {('head_fwd' if 'head_fwd' in r['Name'] else 'head_bwd'): float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if 'head_' in r['Name'] and int(r['Calls']) > 1000}
print('stop', sys.argv[2], ' '.join('%s=%.2f' % kv for kv in sorted(rows.items())))
PY
done
