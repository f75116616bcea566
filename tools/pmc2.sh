#!/bin/bash
# usage (GPU box): tools/pmc2.sh <tag> "<CTR1 CTR2 ...>" <kernel-substring> <cmd...>  -> one rocprofv3 --pmc pass per counter
tag=$1; ctrs=$2; filt=$3; shift 3
cd /tmp && export TMPDIR=/tmp
for c in $ctrs; do
  rm -rf /tmp/pmc2_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc2_$c -o p -- "$@" > /tmp/pmc2_$c.log 2>&1
  f=$(find /tmp/pmc2_$c -name "*counter_collection.csv" | head -1)
  python - "$f" "$c" "$filt" <<'PY'
import csv, sys, collections
f, c, filt = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0.0, 0])
try:
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') != c or filt not in r['Kernel_Name']: continue
        acc['k'][0] += float(r['Counter_Value']); acc['k'][1] += 1
    s, n = acc['k']
    print('%-32s calls %4d avg %14.1f' % (c, n, s / max(n, 1)))
except Exception as e:
    print(c, 'FAILED', e)
PY
done
