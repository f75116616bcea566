#!/bin/bash
# usage (GPU box): tools/pmc.sh <tag>   -> gpurun_out/pmc_<tag>_{fetch,write}.csv  (separate passes, kernel-trace only)
tag=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary 0 > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" "$c" > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$c.txt <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r.get('Counter_Name') != c: continue
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
print('kernel,calls,avg_%s' % c)
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print('%s,%d,%.1f' % (k, n, s / n))
PY
done
head -30 $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_FETCH_SIZE.txt; echo; head -30 $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_WRITE_SIZE.txt; tail -3 /tmp/pmc_FETCH_SIZE.log | cut -c1-300
