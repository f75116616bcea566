#!/bin/bash
# usage (GPU box): tools/pmc3.sh "<CTR1 CTR2 ...>" <cmd...>  -> one rocprofv3 --pmc pass per counter, per-kernel averages for every kernel
ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in $ctrs; do
  rm -rf /tmp/pmc3_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc3_$c -o p -- "$@" > /tmp/pmc3_$c.log 2>&1
  f=$(find /tmp/pmc3_$c -name "*counter_collection.csv" | head -1)
  python - "$f" "$c" <<'PY'
import csv, sys, collections
f, c = sys.argv[1:3]
acc = collections.defaultdict(lambda: [0.0, 0])
try:
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') != c: continue
        k = r['Kernel_Name'].split('(')[0][:48]
        acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
        print('%-28s %-48s calls %5d avg %14.1f' % (c, k, n, s / max(n, 1)))
except Exception as e:
    print(c, 'FAILED', e)
PY
done
