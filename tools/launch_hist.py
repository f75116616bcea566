"""dev (GPU box): per-launch durations of one kernel from a rocprofv3 --kernel-trace CSV: histogram, and the durations by position inside the
bench's 100-iteration fresh-start loops (is the slow tail the loops' cold first iterations?)."""
import csv, sys, glob
import numpy as np
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
name = sys.argv[2]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(f)) if name in r['Kernel_Name']]
rows.sort()
d = np.array([x[1] for x in rows]) / 1e3
print(name, 'launches', len(d), 'avg %.2f med %.2f p90 %.2f p99 %.2f max %.2f' % (d.mean(), np.median(d), np.quantile(d, .9), np.quantile(d, .99), d.max()))
print('histogram (us):', ' '.join('%d-%d:%d' % (lo, lo + 4, int(((d >= lo) & (d < lo + 4)).sum())) for lo in range(20, 80, 4)))
slow = np.nonzero(d > 40)[0]
print('launches > 40 us:', len(slow), 'indices', slow[:40].tolist(), 'gaps', np.diff(slow)[:40].tolist())
for i in slow[:6]:
    print('  around', i, np.round(d[max(i - 2, 0):i + 6], 1).tolist())
